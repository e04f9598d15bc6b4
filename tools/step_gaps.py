#!/usr/bin/env python3
"""Idle time of the device inside the last `nsteps` steps of a rocprofv3 rocpd trace of bench.py (side stream on): union of the kernel
intervals vs wall time, and the kernels that most often precede a gap.  usage: step_gaps.py results.db bench.log [nsteps]"""
import re, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
step_ms = float(re.search(r'ms_per_step": ([0-9.]+)', open(sys.argv[2]).read()).group(1))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tmax = db.execute("select max(end) from kernels").fetchone()[0]
rows = db.execute(f"select name, start, end from kernels where start >= {int(tmax - n * step_ms * 1e6)} order by start").fetchall()
short = lambda k: re.sub(r'\(.*$', '', re.sub(r'\(anonymous namespace\)::|void ', '', k))[:70]
t0, t1 = rows[0][1], max(r[2] for r in rows)
busy, cur_s, cur_e, last_name = 0, rows[0][1], rows[0][2], rows[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
for name, s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        g = gaps[(short(last_name), short(name))]; g[0] += 1; g[1] += s - cur_e
        cur_s, cur_e, last_name = s, e, name
    else:
        if e > cur_e: cur_e, last_name = e, name
busy += cur_e - cur_s
tot = sum(e - s for _, s, e in rows)
print(f"window {(t1 - t0) / n / 1e6:.2f} ms/step: device busy {busy / n / 1e6:.2f} ms, idle {(t1 - t0 - busy) / n / 1e6:.2f} ms, sum of kernel durations {tot / n / 1e6:.2f} ms (overlap {(tot - busy) / n / 1e6:.2f} ms), {len(rows) / n:.0f} kernels")
print("idle_ms_per_step,count_per_step,avg_gap_us,before -> after")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda x: -x[1][1])[:25]:
    print(f"{t / n / 1e6:.3f},{c / n:.1f},{t / c / 1e3:.1f},{a} -> {b}")
