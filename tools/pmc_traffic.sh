#!/bin/bash
# HBM traffic of the forward Linear GEMM launches of bench.py's train step, from two separate PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Writes $1 (default gpurun_out/pmc_traffic.json).
# FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM: gfx950 tallies the 128-B requests of wide coalesced reads at 64 B);
# WRITE_SIZE is used as reported (it matches the output bytes of a GEMM exactly: tools/gemm_pmc_probe.py).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/pmc_traffic.json}; WHICH=${2:-fwd}; W=gpurun_out/pmc_work; rm -rf $W; mkdir -p $W          # $2 = dw: the split-K weight-gradient kernel instead (round 5: bench.py's roofline.kernel)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $W/$c -o p -- python bench.py --graph 0 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $W/$c.log 2>&1
done
python - <<PY
import sqlite3, json, glob
which = "$WHICH"
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(glob.glob("$W/%s/*.db" % c)[0])
    # forward-form Linear launches (lmv_linear_fwd and its fused forms): the 128 x 128 tile kernel gemm_kernel<unsigned short, false, false, false, ...>,
    # the register-stationary kernel and the whole-width kernel (rsgemm.hip / wngemm.hip); 2 warm-up + 3 timed steps: last 3/5 of the launches
    if which == "dw":
        rows = db.execute("select value, duration from counters_collection where counter_name=? and kernel_name like '%gemm_kernel<unsigned short, true, true, true%' order by start", (c,)).fetchall()
    else:
        rows = db.execute("select value, duration from counters_collection where counter_name=? and (kernel_name like '%gemm_kernel<unsigned short, false, false, false%' "
                          "or kernel_name like '%rs_gemm_kernel%' or kernel_name like '%wn_gemm_kernel%') order by start", (c,)).fetchall()
    n = len(rows); rows = rows[n * 2 // 5:]
    res[c] = {"launches": len(rows), "avg_kb": sum(r[0] for r in rows) / len(rows), "avg_us": sum(r[1] for r in rows) / len(rows) / 1e3}
fetch = 2.0 * res["FETCH_SIZE"]["avg_kb"] * 1024; write = res["WRITE_SIZE"]["avg_kb"] * 1024
out = {"kernel": "gemm_kernel<bf16, TR, TR, split-K> (weight-gradient GEMM) launches of bench.py's train step" if which == "dw" else "forward-form Linear launches (gemm_kernel<bf16,NT>, rs_gemm_kernel, wn_gemm_kernel) of bench.py's train step (native block schedule: forward layers, and the dX launches that run as forward-form GEMMs on transposed weights)", "launches_averaged": res["FETCH_SIZE"]["launches"],
       "fetch_bytes_per_launch_corrected_x2": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
       "avg_launch_us_under_pmc": res["FETCH_SIZE"]["avg_us"], "raw": res}
out["csrc_hash"] = "$(python bench.py --print-csrc-hash)"
json.dump(out, open("$OUT", "w"), indent=1); print(json.dumps(out))
PY
rm -rf $W
