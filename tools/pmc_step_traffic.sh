#!/bin/bash
# HBM bytes of ONE pass of bench.py (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE, separate --pmc passes, summed over every
# kernel of the timed steps).  usage (GPU box): bash tools/pmc_step_traffic.sh <out.json> [bench args, e.g. --mode infer]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; OUT=$1; shift
D=gpurun_out/pst; rm -rf $D; mkdir -p $D
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $D/$c -o p -- python bench.py --graph 0 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe "$@" > $D/$c.log 2>&1
done
python - "$OUT" "$@" <<PY
import sqlite3, glob, json, sys
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"$D/{c}/**/*.db", recursive=True)[0]
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    cc = [t for t in tabs if t.startswith('counters_collection')][0]
    rows = db.execute(f"select kernel_name, value, start from {cc} where counter_name = '{c}' order by start").fetchall()
    # steps: 2 warm-up + 4 timed (--no-issue-probe); a step starts at the stem's first launch (once per pass), the first passes carry
    # one-time kernels (weight folds, casts), so count from the 3rd marker to the end = the 4 timed passes
    marks = [i for i, r in enumerate(rows) if 'im2col_c3_kernel' in r[0] or 'stem_kernel' in r[0]]          # (training / the per-launch stem; the one-launch stem of the inference forward)
    assert len(marks) % 6 == 0 and marks, len(marks)      # (inference as k concurrent sub-batches: k stem launches per pass)
    sel = rows[marks[2 * (len(marks) // 6)]:]
    per = len(sel) // 4
    kb = sum(r[1] for r in sel) / 4
    by = {}
    for k, v, _ in sel:
        k = k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
        by[k] = by.get(k, 0) + v / 4
    res[c] = dict(kb_per_step=kb, launches_per_step=per, top=sorted(((round(v / 1024, 1), k) for k, v in by.items()), reverse=True)[:12])
batch = 128
args = sys.argv[2:]
if "--batch" in args: batch = int(args[args.index("--batch") + 1])
fetch, write = res["FETCH_SIZE"]["kb_per_step"] * 1024 * 2, res["WRITE_SIZE"]["kb_per_step"] * 1024
out = dict(command="bench.py --graph 0 " + " ".join(args), fetch_bytes_per_step_corrected_x2=fetch, write_bytes_per_step=write, hbm_mb_per_step=(fetch + write) / 1e6,
           hbm_mb_per_image=(fetch + write) / 1e6 / batch, launches_per_step=res["FETCH_SIZE"]["launches_per_step"],
           top_fetch_mb_uncorrected=res["FETCH_SIZE"]["top"], top_write_mb=res["WRITE_SIZE"]["top"])
out["csrc_hash"] = "$(python bench.py --print-csrc-hash)"
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("command", "hbm_mb_per_step", "hbm_mb_per_image", "launches_per_step")}))
PY
rm -rf $D
