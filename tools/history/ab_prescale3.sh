#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/prescale; mkdir -p $O
for v in 1 0; do
  LMV_PRESCALE=$v rocprofv3 --hip-trace -d $O/ht$v -o trace --output-format csv -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/hip$v.log 2>&1
  python tools/hip_api_tail.py $O/ht$v/trace_hip_api_trace.csv 150 > $O/hip_tail$v.txt 2>&1; rm -rf $O/ht$v
done
cat $O/hip_tail1.txt; cat $O/hip_tail0.txt
