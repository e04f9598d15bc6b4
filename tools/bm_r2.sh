for cfg in "X=0 --graph 0" "X=0 --graph 1" "LMV_SIDE_STREAM=0 --graph 0" "LMV_META_SIDE_STREAM=0 --graph 1" ; do
  set -- $cfg
  echo "== $cfg"; env $1 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 --warmup 8 $2 $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['launch'])"
done
python tools/cpu_launch_time.py 2>&1 | tail -5
