#!/bin/bash
# rocprofv3 kernel trace of a short train run, summarised per (kernel, grid): usage  [ENV=..] bash tools/prof_grid.sh <tag> [name filter]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; T=${1:-grid}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o trace -- python bench.py $PROF_ARGS --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_by_grid.py $DB 120 "$2" > $O/by_grid.csv
rm -rf $O/kt
tail -1 $O/bench.log | cut -c1-160; head -40 $O/by_grid.csv
