#!/bin/bash
# forward A/B over environment settings, interleaved: ab_env_infer.sh rounds "ENV1" "ENV2" ...   (each ENV a quoted list of VAR=VALUE, "" = defaults)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=$1; shift
for i in $(seq $R); do
  for E in "$@"; do
    echo "[$E] $(env $E timeout 600 python bench.py --mode infer --no-cpu-baseline --no-kernel-timing ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('forward_schedule_probe_ms'))")"
  done
done
