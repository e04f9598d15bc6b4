# Vector-memory-path counters (TA / TCP / TD / TCC) of one forward Linear GEMM shape: is the L2 -> LDS operand stream the limiter?
# usage (GPU box): bash tools/pmc_mem.sh [rows_x rows_c N K]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/pmcm; rm -rf $O; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TA|TCP|TD|TCC)_[A-Za-z0-9_]+" | sort -u > $O/avail.txt
wc -l $O/avail.txt
i=0
for set in "TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TD_BUSY_avr TCC_BUSY_avr" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "TCC_WRITE_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/s$i -o p -- python tools/gemm_pmc_probe.py $@ > $O/s$i.log 2>&1
done
python - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$O/s*/*.db")):
    db = sqlite3.connect(f)
    try:
        for r in db.execute("select counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%gemm_kernel%' group by 1"):
            print(f"{r[0]:44s} n={r[1]:3d} avg={r[2]:18.1f} dur_us={r[3]/1e3:8.1f}")
    except Exception as e:
        print(f, e)
PY
for f in $O/s*.log; do grep -i -m2 "error\|invalid\|not found" $f; done
cp $O/avail.txt gpurun_out/pmc_avail.txt; rm -rf $O
