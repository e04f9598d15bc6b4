cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r01f; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats -d $O/kt -o trace -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $O/kernel_stats_full.csv
python tools/step_breakdown.py $DB $O/bench_under_rocprof.log 3 > $O/step_breakdown.csv
find $O/kt -name "*stats*.csv" | head; cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null
rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p -- python tools/gemm_pmc_probe.py > $O/pmc_$c.log 2>&1
  DBP=$(find $O/pmc_$c -name "*.db" | head -1)
  python - <<PY > $O/pmc_${c}_summary.txt
import sqlite3
db = sqlite3.connect("$DBP")
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print("tables:", [t for t in tabs if 'pmc' in t.lower() or 'counter' in t.lower()])
for t in tabs:
    if t.lower() in ('counters_collection', 'pmc_events') or 'counters_collection' == t.lower():
        cols = [r[1] for r in db.execute(f"pragma table_info({t})")]
        print(t, cols)
        for row in db.execute(f"select * from {t} limit 3"): print(row)
PY
done
tail -1 $O/bench_default.json | cut -c1-400
head -12 $O/step_breakdown.csv
cat $O/pmc_FETCH_SIZE_summary.txt | head -20
