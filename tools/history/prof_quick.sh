#!/bin/bash
# Short profile bundle (a few minutes): bench line, rocprofv3 --kernel-trace --stats summary of the same command, per-step breakdown.
# usage (on the GPU box): bash tools/prof_quick.sh <tag>   -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; T=${1:-r03}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_train.json 2> $O/bench_train.err
rocprofv3 --kernel-trace --stats -d $O/kt -o trace -- python bench.py --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench_under_rocprof.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB 400 > $O/train_kernel_stats_steady.csv
rm -rf $O/kt
LMV_SIDE_STREAM=0 rocprofv3 --kernel-trace -d $O/kt2 -o trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench_inline_under_rocprof.log 2>&1
DB2=$(find $O/kt2 -name "*.db" | head -1)
python tools/step_breakdown.py $DB2 $O/bench_inline_under_rocprof.log 3 > $O/train_step_breakdown.csv
rm -rf $O/kt2
tail -1 $O/bench_train.json | cut -c1-300; head -30 $O/train_step_breakdown.csv
