#!/usr/bin/env python3
"""Per-(kernel, grid) breakdown of the last `nsteps` steps of a rocprofv3 rocpd trace of bench.py: tells the per-SHAPE cost of the
shared GEMM / attention kernels inside the step.  usage: step_by_grid.py results.db bench.log [nsteps]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
step_ms = float(re.search(r'ms_per_step": ([0-9.]+)', open(sys.argv[2]).read()).group(1))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("# columns:", cols, file=sys.stderr)
gcols = [c for c in cols if re.match(r"grid(_size)?_?[xyz]$", c)] or [c for c in cols if "grid" in c]
wcols = [c for c in cols if re.match(r"workgroup(_size)?_?[xyz]$", c)]
sel = ", ".join(["name", "start", "end"] + gcols + wcols)
tmax = db.execute("select max(end) from kernels").fetchone()[0]
rows = db.execute(f"select {sel} from kernels where start >= {int(tmax - n * step_ms * 1e6)} order by start").fetchall()
agg = {}
for r in rows:
    name = re.sub(r'\(anonymous namespace\)::|void ', '', r[0])
    name = re.sub(r'\(.*$', '', name)[:80]
    key = (name,) + tuple(r[3:])
    a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += r[2] - r[1]
print("ms_per_step,calls_per_step,avg_us,grid+wg,kernel")
for key, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    if t / n / 1e6 < 0.02:
        continue
    print(f"{t / n / 1e6:.3f},{c / n:.1f},{t / c / 1e3:.1f},{'x'.join(str(v) for v in key[1:])},\"{key[0]}\"")
