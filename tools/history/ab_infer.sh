#!/bin/bash
# A/B of environment switches on the inference bench line, interleaved: usage  bash tools/ab_infer.sh "VAR=1" ...   (first of each round: baseline)
# BENCH_ARGS adds bench.py flags (e.g. "--infer-parts 1")
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for e in "" "$@"; do
    r=$(env $e python bench.py --mode infer $BENCH_ARGS --no-cpu-baseline --no-kernel-timing --no-issue-probe --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "[$e] run $i: $r ms"
  done
done
