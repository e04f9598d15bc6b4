#!/bin/bash
# as ab_env.sh, interleaved rounds (A B C A B C ...) so that drift of the box hits every arm equally: usage  bash tools/ab_env3.sh ROUNDS "VAR=1" ...
cd $GRAFT_REPO_ROOT; R=$1; shift
for i in $(seq 1 $R); do
  for e in "" "$@"; do
    r=$(env $e python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-probe --no-issue-probe --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $i [$e]: $r ms"
  done
done
