#!/bin/bash
# Round profile bundle: default bench line, rocprofv3 --kernel-trace --stats of the same command (summarised),
# per-step breakdown, inference numbers, kernel micro-benchmarks, PMC traffic of the dominant kernel.
# usage (on the GPU box): bash tools/prof_round.sh <tag>   -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; T=${1:-r01}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_train.json 2> $O/bench_train.err
python bench.py --mode infer > $O/bench_infer.json 2>> $O/bench_train.err
# (--no-kernel-timing --no-issue-probe: the 3 event-bracketed steps bench.py appends run the per-launch Python schedule and would skew a "last steps" window)
rocprofv3 --kernel-trace --stats -d $O/kt -o trace -- python bench.py --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench_under_rocprof.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $O/train_kernel_stats_full.csv
python tools/rocpd_stats.py $DB 400 > $O/train_kernel_stats_steady.csv
rm -rf $O/kt
# per-step breakdown with the weight-gradient GEMMs in line (side-stream overlap inflates per-kernel durations)
LMV_SIDE_STREAM=0 rocprofv3 --kernel-trace -d $O/kt2 -o trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench_inline_under_rocprof.log 2>&1
DB2=$(find $O/kt2 -name "*.db" | head -1)
python tools/step_breakdown.py $DB2 $O/bench_inline_under_rocprof.log 3 > $O/train_step_breakdown.csv
rm -rf $O/kt2
python tools/bench_kernels.py gemm attn fused > $O/kernel_microbench.txt 2>&1
bash tools/prof_infer.sh $O/infer_kernel_stats.csv > $O/prof_infer.log 2>&1
bash tools/pmc_step_traffic.sh $O/infer_hbm_traffic_fused.json --mode infer > $O/pmc_step_traffic.log 2>&1
LMV_FUSED=0 bash tools/pmc_step_traffic.sh $O/infer_hbm_traffic_unfused.json --mode infer >> $O/pmc_step_traffic.log 2>&1
LMV_FUSED=0 python bench.py --mode infer --no-cpu-baseline > $O/bench_infer_unfused.json 2>> $O/bench_train.err
bash tools/pmc_mlp.sh 128 3136 96 > $O/mlp_fused_pmc_s1.txt 2>&1
bash tools/pmc_mlp.sh 128 196 384 > $O/mlp_fused_pmc_s3.txt 2>&1
bash tools/probe_prof.sh tools/ln_probe.py > $O/ln_kernel_durations.txt 2>&1
bash tools/probe_prof.sh tools/conv_probe.py > $O/conv_kernel_durations.txt 2>&1
bash tools/pmc_traffic.sh $O/gemm_fwd_pmc_traffic.json > $O/pmc_traffic.log 2>&1
tail -1 $O/bench_train.json | cut -c1-300; tail -1 $O/bench_infer.json | cut -c1-200; head -8 $O/train_step_breakdown.csv; cat $O/gemm_fwd_pmc_traffic.json | head -12
