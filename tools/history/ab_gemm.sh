#!/bin/bash
# A/B of a library switch on the GEMM micro-benchmark + its parity tests: usage  bash tools/ab_gemm.sh "VAR=x" ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for e in "" "$@"; do
  echo "=== [$e]"
  env $e python tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids
done
