#!/bin/bash
# usage (GPU box): [SHAPES="M N K;M N K"] bash tools/native/run_rs_probe.sh  -> gpurun_out/rs_probe.txt
cd "$(dirname "$0")"
LIBDIR=../../lemevit_amd/csrc
[ -x ./rs_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 rs_probe.hip -o rs_probe -DWITH_LMV -L$LIBDIR -llemevit_hip -Wl,-rpath,'$ORIGIN/../../lemevit_amd/csrc'
mkdir -p ../../gpurun_out
out=../../gpurun_out/rs_probe.txt
: > $out
IFS=';' read -ra SH <<< "${SHAPES:-27136 1536 384}"
for shape in "${SH[@]}"; do
  timeout 120 ./rs_probe $shape >> $out 2>&1 || echo "FAILED $shape" >> $out
done
cat $out
