#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/prescale; mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "prescale" > $O/tests2.log 2>&1
grep -E "passed|failed|Error" $O/tests2.log | tail -3
for v in 1 0; do LMV_PRESCALE=$v python tools/host_profile.py 45 > $O/host$v.txt 2>&1; done
