#!/usr/bin/env python3
"""Per-step kernel breakdown of a rocprofv3 rocpd trace of bench.py: last `nsteps` steps (steady state).
usage: step_breakdown.py results.db bench.log [nsteps]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
step_ms = float(re.search(r'ms_per_step": ([0-9.]+)', open(sys.argv[2]).read()).group(1))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tmax = db.execute("select max(end) from kernels").fetchone()[0]
win = n * step_ms * 1e6
rows = db.execute(f"select name, start, end from kernels where start >= {int(tmax - win)} order by start").fetchall()
busy = sum(e - s for _, s, e in rows)
print(f"# step {step_ms} ms (under rocprofv3); window = last {n} steps: {len(rows) / n:.0f} kernels/step, busy {busy / n / 1e6:.2f} ms/step, idle {(win - busy) / n / 1e6:.2f} ms/step")
agg = {}
for name, s, e in rows:
    name = re.sub(r'\(anonymous namespace\)::|void ', '', name)
    name = re.sub(r'\(.*$', '', name)[:90]
    a = agg.setdefault(name, [0, 0]); a[0] += 1; a[1] += e - s
print("ms_per_step,calls_per_step,avg_us,kernel")
for name, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{t / n / 1e6:.3f},{c / n:.1f},{t / c / 1e3:.1f},\"{name}\"")
