#!/usr/bin/env python3
"""Per (kernel, grid size) duration summary of the steady part of a rocprofv3 kernel trace: separates the layer shapes one kernel template
serves.  Usage: rocpd_by_grid.py results.db [last_ms] [name filter]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else None
flt = sys.argv[3] if len(sys.argv) > 3 else ""
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
gcol = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "gridx", "grid_size")]
wcol = [c for c in cols if c.lower() in ("workgroup_x", "workgroup_size_x", "wg_x", "workgroup_size")]
if not gcol: print("columns:", cols); sys.exit(1)
where = ""
if last_ms is not None:
    tmax = db.execute("select max(end) from kernels").fetchone()[0]
    where = f"where start >= {int(tmax - last_ms * 1e6)}"
rows = db.execute(f"select {namecol}, {gcol[0]}, {wcol[0] if wcol else 1}, count(*), sum(end - start), min(end - start), max(end - start) from kernels {where} group by {namecol}, {gcol[0]} order by 5 desc").fetchall()
print("kernel,workgroups,calls,total_ms,avg_us,min_us,max_us")
for n, g, w, c, t, mn, mx in rows:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
    if flt and flt not in n: continue
    print(f"\"{n[:110]}\",{g // max(w, 1)},{c},{t / 1e6:.3f},{t / c / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f}")
