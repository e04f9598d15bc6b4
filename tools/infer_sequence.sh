#!/bin/bash
# Ordered kernel list of ONE forward pass (whole batch on one stream): index, start offset, duration, gap before, workgroups, kernel.
# usage (GPU box): bash tools/infer_sequence.sh out.csv [graph 0|1] [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; OUT=${1:-gpurun_out/infer_sequence.csv}; G=${2:-0}; shift; shift
D=gpurun_out/is; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace -d $D -o t -- python bench.py --mode infer --graph $G --infer-parts 1 --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-issue-probe "$@" > $D/log.txt 2>&1
DB=$(find $D -name "*.db" | head -1)
python - <<PY > $OUT
import sqlite3, re
db = sqlite3.connect("$DB")
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wcol = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
sel = "name, start, end" + (f", {gcol}" if gcol else ", 0") + (f", {wcol}" if wcol else ", 1")
rows = db.execute(f"select {sel} from kernels order by start").fetchall()
st = [i for i, r in enumerate(rows) if "stem_kernel" in r[0]]
lo, hi = st[-2], st[-1]
t0 = rows[lo][1]; prev = t0
print("idx,start_us,dur_us,gap_us,workgroups,kernel")
for i, (n, s, e, gx, wx) in enumerate(rows[lo:hi]):
    n = re.sub(r'\(anonymous namespace\)::|void ', '', n); n = re.sub(r'\(.*$', '', n)[:90]
    print(f"{i},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{(s - prev) / 1e3:.1f},{(gx // max(wx, 1)) if gx else 0},\"{n}\"")
    prev = max(prev, e)
print(f"# pass = {(rows[hi][1] - t0) / 1e3:.1f} us start to start")
PY
tail -1 $D/log.txt | cut -c1-160
rm -rf $D
