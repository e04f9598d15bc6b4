for cfg in "LMV_ATTN_PAIR=1" "LMV_ATTN_PAIR=0" "LMV_ATTN_PAIR=1" "LMV_ATTN_PAIR=0"; do
  echo "== $cfg"; env $cfg python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
