#!/bin/bash
# Interleaved bench A/B over environment settings: ab_env.sh ROUNDS "VAR=a" "VAR=b" ... ("-" = no setting)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=$1; shift
run() { s="$1"; [ "$s" = "-" ] && s="LMV_NOP=1"; echo "$1 $(env $s timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-forward-probe --no-issue-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"; }
for i in $(seq $R); do for s in "$@"; do run "$s"; done; done
