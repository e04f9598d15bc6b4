#!/usr/bin/env python3
"""Ordered kernel list of ONE train step from a rocprofv3 rocpd trace of bench.py run in line (LMV_SIDE_STREAM=0 LMV_TRAIN_PARTS=1): the last step, cut at the
optimizer's adamw_kernel.  usage: step_sequence.py results.db  ->  index, start offset us, duration us, gap before us, grid, kernel"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wcol = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
sel = "name, start, end" + (f", {gcol}" if gcol else ", 0") + (f", {wcol}" if wcol else ", 1")
rows = db.execute(f"select {sel} from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
if len(ends) < 2:
    sys.exit("fewer than two optimizer launches in the trace")
lo, hi = ends[-2] + 1, ends[-1] + 1
t0 = rows[lo][1]
prev_end = t0
print("idx,start_us,dur_us,gap_us,workgroups,kernel")
for i, (name, s, e, gx, wx) in enumerate(rows[lo:hi]):
    name = re.sub(r'\(anonymous namespace\)::|void ', '', name)
    name = re.sub(r'\(.*$', '', name)[:100]
    print(f"{i},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{(s - prev_end) / 1e3:.1f},{(gx // max(wx, 1)) if gx else 0},\"{name}\"")
    prev_end = max(prev_end, e)
