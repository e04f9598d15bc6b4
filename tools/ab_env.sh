#!/bin/bash
# A/B of environment switches on the default bench line: usage  bash tools/ab_env.sh "VAR=1" ["VAR2=x" ...]   (first run: baseline)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import torch; print('stream priority range', torch.cuda.Stream.priority_range())" 2>/dev/null
for e in "" "$@"; do
  for i in 1 2; do
    r=$(env $e python bench.py --no-cpu-baseline --no-kernel-timing --no-forward-probe --no-issue-probe --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "[$e] run $i: $r ms"
  done
done
