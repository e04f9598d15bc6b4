#!/bin/bash
# train-step A/B over environment settings, interleaved: ab_env_train.sh rounds "ENV1" "ENV2" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=$1; shift
for i in $(seq $R); do
  for E in "$@"; do
    echo "[$E] $(env $E timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --no-issue-probe ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('forward_ms'))")"
  done
done
