cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/gaps; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o trace -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline > $O/bench.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/step_gaps.py $DB $O/bench.log 3 > $O/gaps.txt 2>&1
rm -rf $O/kt
cat $O/gaps.txt
