# FETCH_SIZE / WRITE_SIZE of ONE Linear GEMM shape and mode (tools/gemm_pmc_probe.py): is the operand stream re-fetched from HBM?
# usage (GPU box): bash tools/pmc_shape.sh rows_x rows_c N K mode      (FETCH_SIZE doubled per the gfx950 correction)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/pmcs; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/$c -o p -- python tools/gemm_pmc_probe.py $@ > $O/$c.log 2>&1
done
python - <<PY
import sqlite3, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(glob.glob("$O/%s/*.db" % c)[0])
    for r in db.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name=? and kernel_name like '%gemm_kernel%' group by 1", (c,)):
        mb = r[2] * 1024 / 1e6 * (2 if c == "FETCH_SIZE" else 1)
        print(f"{c:11s} {r[0][:70]:70s} n={r[1]:3d} {mb:9.1f} MB  {r[3]/1e3:7.1f} us")
PY
tail -1 $O/FETCH_SIZE.log; rm -rf $O
