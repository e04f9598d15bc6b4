#!/bin/bash
# LDS bank-conflict survey per kernel: SQ_LDS_BANK_CONFLICT (extra LDS cycles) over SQ_LDS_IDX_ACTIVE (all LDS-array cycles), and the LDS-array cycles per launch against the launch's
# duration (GRBM_GUI_ACTIVE).  usage (GPU box): bash tools/pmc_lds.sh out.csv [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_lds.csv}; shift; W=gpurun_out/pmc_lds; rm -rf $W; mkdir -p $W
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
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
    act = v.get("SQ_LDS_IDX_ACTIVE", 0.0)
    if not act:
        continue
    conf = v.get("SQ_LDS_BANK_CONFLICT", 0.0); grbm = v.get("GRBM_GUI_ACTIVE", 0.0)
    rows.append((v["n"] * v["us"], k, v["n"] / 5.0, v["us"], conf / act, act / (grbm * 32.0) if grbm else 0.0, v.get("SQ_LDS_ADDR_CONFLICT", 0.0)))
rows.sort(reverse=True)
with open("$OUT", "w") as f:
    f.write("kernel,launches_per_step,avg_us_under_pmc,bank_conflict_cycles_over_lds_active,lds_active_cycles_per_CU_over_launch_cycles,addr_conflict\n")
    for tot, k, calls, us, ratio, share, ac in rows[:40]:
        f.write(f"\\"{k}\\",{calls:.1f},{us:.1f},{ratio:.3f},{share:.3f},{ac:.0f}\n")
print(open("$OUT").read()[:6000])
PY
