cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/pmcg; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/s$i -o p -- python tools/gemm_pmc_probe.py $@ > $O/s$i.log 2>&1
done
python - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$O/s*/*.db")):
    db = sqlite3.connect(f)
    for r in db.execute("select counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%gemm_kernel%' group by 1"):
        print(f"{r[0]:34s} n={r[1]:3d} avg={r[2]:16.1f} dur_us={r[3]/1e3:8.1f}")
PY
rm -rf $O
