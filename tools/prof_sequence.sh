#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/seq; mkdir -p $O
LMV_SIDE_STREAM=0 LMV_TRAIN_PARTS=1 rocprofv3 --kernel-trace -d $O/kt -o trace -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/step_sequence.py $DB > $O/train_step_sequence.csv 2> $O/seq.err
rm -rf $O/kt; wc -l $O/train_step_sequence.csv; cat $O/seq.err | tail -3
