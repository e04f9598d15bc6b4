#!/bin/bash
# forward A/B of the in-tree library against tools/native/ab/$1 (parity of the stage kernels first): ab_quick_infer.sh other.so [rounds]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
NEW=$PWD/lemevit_amd/csrc/liblemevit_hip.so; OLD=$PWD/tools/native/ab/$1; R=${2:-3}
timeout 900 python -m pytest tests/test_sstage_gpu.py tests/test_dstage_gpu.py -x -q -m gpu -k "vs_oracle or vs_per_launch" 2>&1 | tail -1
run() { echo "$(basename $1) $2 $(env LMV_LIB_PATH=$1 timeout 600 python bench.py $2 --mode infer --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"; }
for i in $(seq $R); do run $NEW ""; run $OLD ""; done
for i in $(seq $R); do run $NEW "--model lemevit_tiny --batch 256"; run $OLD "--model lemevit_tiny --batch 256"; done
