#!/bin/bash
# usage (GPU box): bash tools/native/run_panel_probe.sh  -> gpurun_out/panel_probe.txt
set -e
cd "$(dirname "$0")"
[ -x ./panel_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 panel_probe.hip -o panel_probe
mkdir -p ../../gpurun_out
out=../../gpurun_out/panel_probe.txt
: > $out
for shape in "27136 1536 384" "27136 1152 384" "27136 384 384" "27136 384 1536" "102400 768 192" "102400 192 768" "8320 2048 512" "8320 512 2048" "403456 96 384"; do
  timeout 120 ./panel_probe $shape >> $out 2>&1 || echo "FAILED $shape" >> $out
done
cat $out
