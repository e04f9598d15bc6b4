#!/bin/bash
# Interleaved forward A/B of two library builds (+ stage-kernel parity): ab_infer_lib.sh ROUNDS other.so
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=$1; OLD=$PWD/$2; NEW=$PWD/lemevit_amd/csrc/liblemevit_hip.so
timeout 1200 python -m pytest tests/test_sstage_gpu.py tests/test_dstage_gpu.py -x -q -m gpu > gpurun_out/ab_infer_tests.log 2>&1; grep -E "passed|failed|Error" gpurun_out/ab_infer_tests.log | tail -3
run() { echo "$(basename $1) $2 $(env LMV_LIB_PATH=$1 timeout 600 python bench.py $2 --mode infer --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"; }
for i in $(seq $R); do run $NEW ""; run $OLD ""; done
for i in $(seq $R); do run $NEW "--model lemevit_tiny --batch 256"; run $OLD "--model lemevit_tiny --batch 256"; done
python tools/sstage_timeline.py 5 2>&1 | sed -n 2,20p
