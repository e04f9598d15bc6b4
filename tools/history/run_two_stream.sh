cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/ts; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o trace -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/two_stream.py $DB $O/bench.log 3 2>&1 | tee $O/two_stream.txt
python tools/step_gaps.py $DB $O/bench.log 3 2>&1 | head -12 | tee $O/gaps.txt
rm -rf $O/kt
