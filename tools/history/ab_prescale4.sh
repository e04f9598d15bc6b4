#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/prescale; mkdir -p $O
for i in 1 2; do for v in 0 1; do LMV_PRESCALE=$v python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_v${v}_$i.json; done; done
python - <<'P'
import json
for i in (1, 2):
    a = json.load(open(f"gpurun_out/prescale/bench_v0_{i}.json")); b = json.load(open(f"gpurun_out/prescale/bench_v1_{i}.json"))
    print(i, a["ms_per_step"], b["ms_per_step"])
    for k in a:
        if isinstance(a[k], dict) and k not in ("config",):
            print(k, json.dumps(a[k])[:1500]); print(k, json.dumps(b.get(k))[:1500])
P
