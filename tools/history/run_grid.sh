cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r02d; rm -rf $O; mkdir -p $O
LMV_SIDE_STREAM=0 rocprofv3 --kernel-trace -d $O/kt2 -o trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_inline.log 2>&1
DB2=$(find $O/kt2 -name "*.db" | head -1)
python tools/step_by_grid.py $DB2 $O/bench_inline.log 3 > $O/step_by_grid.csv 2> $O/step_by_grid.err
rm -rf $O/kt2
head -70 $O/step_by_grid.csv
