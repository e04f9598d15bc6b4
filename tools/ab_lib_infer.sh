#!/bin/bash
# forward A/B of the in-tree library against another build (tools/native/ab/$1), interleaved: ab_lib_infer.sh other.so [rounds] [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
NEW=$PWD/lemevit_amd/csrc/liblemevit_hip.so; OLD=$PWD/tools/native/ab/$1; R=${2:-3}; shift; shift
run() { echo "$(basename $1) $(env LMV_LIB_PATH=$1 timeout 600 python bench.py --mode infer --no-cpu-baseline --no-kernel-timing "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('forward_schedule_probe_ms'))")"; }
for i in $(seq $R); do run $NEW "$@"; run $OLD "$@"; done
