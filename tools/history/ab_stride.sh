#!/bin/bash
# staging-image stride of the in-kernel depth-wise convolution (SS_STG_ROW 104 / DS_STG_ENTRY 40 vs the packed 96 / 32 of tools/native/ab/old_stride.so): parity, timelines, forward A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/stride; mkdir -p $O
timeout 1200 python -m pytest tests/test_sstage_gpu.py tests/test_dstage_gpu.py tests/test_parity_budget_gpu.py -x -q -m gpu > $O/tests.log 2>&1
grep -E "passed|failed|Error" $O/tests.log | tail -3
python tools/sstage_timeline.py 5 > $O/sstage_timeline_new.txt 2>&1; head -16 $O/sstage_timeline_new.txt | tail -15
LMV_LIB_PATH=$PWD/tools/native/ab/old_stride.so python tools/sstage_timeline.py 5 > $O/sstage_timeline_old.txt 2>&1; sed -n 2,3p $O/sstage_timeline_old.txt; grep "inside dwconv\|dwconv taps" $O/sstage_timeline_old.txt
run() { echo "$1 $2 $(env LMV_LIB_PATH=$1 timeout 600 python bench.py $2 --mode infer --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"; }
NEW=$PWD/lemevit_amd/csrc/liblemevit_hip.so; OLD=$PWD/tools/native/ab/old_stride.so
for i in 1 2; do run $NEW ""; run $OLD ""; done
for i in 1 2; do run $NEW "--model lemevit_tiny --batch 256"; run $OLD "--model lemevit_tiny --batch 256"; done
run $NEW "--img 384 --batch 64"; run $OLD "--img 384 --batch 64"
(python tools/stage_times.py lemevit_base 128; python tools/stage_times.py lemevit_tiny 256) > $O/stage_times_new.txt 2>&1; grep "stage\|whole" $O/stage_times_new.txt
