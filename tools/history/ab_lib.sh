#!/bin/bash
# Interleaved train-step A/B of two library builds: ab_lib.sh ROUNDS other.so [pytest -k expression]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=$1; OLD=$PWD/$2; NEW=$PWD/lemevit_amd/csrc/liblemevit_hip.so
if [ -n "$3" ]; then timeout 1500 python -m pytest tests -x -q -m gpu -k "$3" > gpurun_out/ab_lib_tests.log 2>&1; grep -E "passed|failed|Error" gpurun_out/ab_lib_tests.log | tail -3; fi
run() { echo "$(basename $1) $(env LMV_LIB_PATH=$1 timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-forward-probe --no-issue-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline'].get('other_launch_kinds',{}); print(d['value'], d['ms_per_step'], 'attn_bwd', k.get('attention_bwd'))")"; }
for i in $(seq $R); do run $NEW; run $OLD; done
