#!/bin/bash
# A/B of the cross-block DropPath pre-scaling (LMV_PRESCALE, LMV_PRESCALE_MAX_MB): interleaved bench runs.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { echo "$1 $(env $1 timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"; }
for i in 1 2 3; do
  run LMV_PRESCALE=0
  run LMV_PRESCALE_MAX_MB=30
  run LMV_PRESCALE_MAX_MB=50
  run LMV_PRESCALE=1
done
