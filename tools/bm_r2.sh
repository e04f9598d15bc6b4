for cfg in "LMV_GEMM_BIGK=0" "LMV_GEMM_BIGK=1" "LMV_GEMM_BIGK=0 LMV_DW_TARGET_BLOCKS=384" "LMV_GEMM_BIGK=0 LMV_DW_TARGET_BLOCKS=512" "LMV_GEMM_BIGK=1 LMV_DW_TARGET_BLOCKS=128" "LMV_GEMM_BIGK=1 LMV_DW_TARGET_BLOCKS=192"; do
  echo "== $cfg"; env $cfg python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
