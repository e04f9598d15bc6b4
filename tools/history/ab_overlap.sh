#!/bin/bash
# A/B of FlatAdamW(overlap=k) (bench.py --adamw-overlap): parity test, then interleaved bench runs.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/overlap; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "adamw or flat_grad or two_rank or model_ema or train_step or eval_fold" > $O/tests.log 2>&1
grep -E "passed|failed|Error" $O/tests.log | tail -3
run() { echo "overlap=$1 $(timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-forward-probe --adamw-overlap $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('host_issue_ms_per_step'))")"; }
for i in 1 2 3; do run 0; run 4; run 2; done
