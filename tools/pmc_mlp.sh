# SQ / LDS / HBM counters of the fused MLP kernel (separate --pmc passes; usage on the GPU box: bash tools/pmc_mlp.sh B N C)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/pmcm; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/s$i -o p -- python tools/mlp_pmc_probe.py $@ > $O/s$i.log 2>&1
done
python - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$O/s*/*.db")):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    cc = [t for t in tabs if t.startswith('counters_collection')][0]
    for r in db.execute(f"select counter_name, count(*), avg(value), avg(duration) from {cc} where kernel_name like '%mlp_fused%' group by 1"):
        print(f"{r[0]:34s} n={r[1]:3d} avg={r[2]:18.1f} dur_us={r[3]/1e3:8.1f}")
PY
rm -rf $O
