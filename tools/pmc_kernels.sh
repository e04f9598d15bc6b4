#!/bin/bash
# Per-kernel PMC table of bench.py's train step (SURVEY 8(d) evidence): HBM bytes per launch (FETCH_SIZE x 2 per the gfx950
# correction + WRITE_SIZE), achieved GB/s, and matrix-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES over the launch's SIMD-cycles:
# GRBM_GUI_ACTIVE is summed over the 8 XCDs, a launch offers GRBM_GUI_ACTIVE / 8 x 1024 SIMD-cycles).  Separate passes per set.
# usage (GPU box): bash tools/pmc_kernels.sh out.csv [bench args, e.g. --mode infer]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_per_kernel.csv}; shift; W=gpurun_out/pmc_k; rm -rf $W; mkdir -p $W
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $W/s$i -o p -- python bench.py --graph 0 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe "$@" > $W/s$i.log 2>&1
done
python - <<PY
import sqlite3, glob, re, collections
vals = collections.defaultdict(dict)
def short(n):
    n = re.sub(r"^void ", "", n); n = n.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)[:110]
for f in sorted(glob.glob("$W/s*/*.db")):
    db = sqlite3.connect(f)
    q = "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by 1, 2"
    for name, ctr, n, v, d in db.execute(q):
        k = short(name)
        vals[k][ctr] = v; vals[k].setdefault("n", n); vals[k].setdefault("us", d / 1e3)
rows = []
for k, v in vals.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    fetch = 2.0 * v["FETCH_SIZE"] * 1024; write = v["WRITE_SIZE"] * 1024
    us = v["us"]; calls = v["n"] / 5.0
    mfma = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); grbm = v.get("GRBM_GUI_ACTIVE", 0.0)
    util = mfma / (grbm * 128.0) if grbm else 0.0
    rows.append((calls * us, k, calls, us, fetch / 1e6, write / 1e6, (fetch + write) / (us * 1e-6) / 1e9, util))
rows.sort(reverse=True)
with open("$OUT", "w") as f:
    f.write("kernel,launches_per_step,avg_us_under_pmc,fetch_MB_per_launch(x2_corrected),write_MB_per_launch,achieved_GBps,mfma_busy_frac\n")
    for tot, k, calls, us, fm, wm, gbs, util in rows[:40]:
        f.write(f"\\"{k}\\",{calls:.1f},{us:.1f},{fm:.2f},{wm:.2f},{gbs:.0f},{util:.3f}\n")
print(open("$OUT").read()[:4000])
PY
rm -rf $W
python bench.py --print-csrc-hash > $OUT.csrc_hash
