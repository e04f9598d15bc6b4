for cfg in "LMV_ATTN_FUSED_BWD=1" "LMV_ATTN_FUSED_BWD=0" "LMV_ATTN_FUSED_BWD=1 LMV_CONV_NATIVE=0" "LMV_ATTN_FUSED_BWD=1 LMV_META_SIDE_STREAM=0"; do
  echo "== $cfg"; env $cfg python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
