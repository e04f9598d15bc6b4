#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_stem_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
for L in lemevit_amd/csrc/liblemevit_hip.so tools/native/ab/pre_stem.so; do echo "$L $(LMV_LIB_PATH=$PWD/$L python tools/stage_times.py lemevit_base 128 2>/dev/null | grep "downsample 0\|whole")"; echo "$L $(LMV_LIB_PATH=$PWD/$L python tools/stage_times.py lemevit_tiny 256 2>/dev/null | grep "downsample 0\|whole")"; done
done
