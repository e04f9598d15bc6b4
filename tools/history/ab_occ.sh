#!/bin/bash
# A/B of a dstage build against tools/native/ab/pre_occ.so: parity, stage times, forward bench (args: model batch)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
NEW=$PWD/lemevit_amd/csrc/liblemevit_hip.so; OLD=$PWD/tools/native/ab/pre_occ.so; M=${1:-lemevit_base}; B=${2:-128}
timeout 900 python -m pytest tests/test_dstage_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do for L in $NEW $OLD; do echo "$(basename $L) $(LMV_LIB_PATH=$L python tools/stage_times.py $M $B 2>/dev/null | grep "whole\|stage 1\|stage 2" | tr '\n' ' ')"; done; done
run() { echo "$(basename $1) $2 $(env LMV_LIB_PATH=$1 timeout 600 python bench.py $2 --mode infer --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"; }
for i in 1 2 3; do run $NEW "--model $M --batch $B"; run $OLD "--model $M --batch $B"; done
