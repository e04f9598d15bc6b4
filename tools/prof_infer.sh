#!/bin/bash
# Per-kernel breakdown of the inference forward (bench.py --mode infer, eager launches so every kernel is traced once per step).
# usage (on the GPU box): bash tools/prof_infer.sh <out.csv> [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; OUT=${1:-gpurun_out/infer_kernel_stats.csv}; shift
D=gpurun_out/pi; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace -d $D -o t -- python bench.py --mode infer --graph 0 --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-issue-probe "$@" > $D/log.txt 2>&1
DB=$(find $D -name "*.db" | head -1)
python - <<PY > $OUT
import sqlite3, re, collections
db = sqlite3.connect("$DB")
rows = db.execute("select name, start, end from kernels order by start").fetchall()
tmax = max(r[2] for r in rows)
# the last 10 steps: take kernels whose start lies in the final 60 % of the trace window after the warm-up; simpler: count per name / 15 steps
agg = collections.OrderedDict()
for n, s, e in rows:
    n = re.sub(r'\(anonymous namespace\)::|void ', '', n)[:110]
    agg.setdefault(n, []).append((e - s) / 1e3)
steps = 15          # bench.py --steps 10 --warmup 5 --no-issue-probe: 5 warm-up + 10 timed passes
tot = 0.0
out = []
for n, v in agg.items():
    if len(v) < steps: continue
    per = len(v) / steps
    tail = v[-int(per * 10):]               # the timed steps
    ms = sum(tail) / 10 / 1e3
    tot += ms
    out.append((ms, per, sum(tail) / len(tail), n))
print("ms_per_step,calls_per_step,avg_us,kernel")
for ms, per, avg, n in sorted(out, reverse=True):
    print(f"{ms:.3f},{per:.1f},{avg:.1f},\"{n}\"")
print(f"{tot:.3f},,,\"TOTAL kernel time per forward pass\"")
PY
tail -1 $D/log.txt | cut -c1-200
rm -rf $D
head -30 $OUT
