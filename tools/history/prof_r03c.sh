#!/bin/bash
# Round-3 closing profile bundle (GPU box), everything at HEAD -> gpurun_out/r03/ : bench lines (train, inference with 4 / 1 sub-batch streams, the other
# configs), rocprofv3 kernel stats of the train command, per-step breakdown, per-kernel PMC tables, HBM traffic per step, the two-stream picture.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_train.json 2> $O/bench_train.err
python bench.py --mode infer --no-cpu-baseline > $O/bench_infer.json 2>> $O/bench_train.err
python bench.py --mode infer --no-cpu-baseline --infer-parts 1 > $O/bench_infer_one_stream.json 2>> $O/bench_train.err
LMV_TRAIN_PARTS=1 python bench.py --no-cpu-baseline --no-forward-probe > $O/bench_train_one_range.json 2>> $O/bench_train.err
rocprofv3 --kernel-trace --stats -d $O/kt -o trace -- python bench.py --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench_under_rocprof.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $O/train_kernel_stats_full.csv
python tools/rocpd_stats.py $DB 400 > $O/train_kernel_stats_steady.csv
python tools/rocpd_by_grid.py $DB 400 > $O/train_kernel_stats_by_grid.csv
python tools/two_stream.py $DB $O/bench_under_rocprof.log 3 > $O/two_stream.txt 2>&1
rm -rf $O/kt
LMV_SIDE_STREAM=0 LMV_TRAIN_PARTS=1 rocprofv3 --kernel-trace -d $O/kt2 -o trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench_inline_under_rocprof.log 2>&1
DB2=$(find $O/kt2 -name "*.db" | head -1)
python tools/step_breakdown.py $DB2 $O/bench_inline_under_rocprof.log 3 > $O/train_step_breakdown.csv
rm -rf $O/kt2
bash tools/prof_infer.sh $O/infer_kernel_stats.csv --infer-parts 1 > $O/prof_infer.log 2>&1
bash tools/pmc_kernels.sh $O/pmc_per_kernel_train.csv > gpurun_out/pmc_k_train.log 2>&1
bash tools/pmc_step_traffic.sh $O/train_hbm_traffic.json > gpurun_out/pmc_traffic_train.log 2>&1
bash tools/pmc_step_traffic.sh $O/infer_hbm_traffic_fused.json --mode infer >> gpurun_out/pmc_traffic_train.log 2>&1
python bench.py --torch-adamw --no-cpu-baseline --no-forward-probe > $O/bench_base224_train_torch_adamw.json 2>/dev/null
python bench.py --model lemevit_tiny --batch 256 --mode infer --no-cpu-baseline > $O/bench_tiny224_b256_infer.json 2>/dev/null
python bench.py --model lemevit_tiny --batch 256 --no-cpu-baseline --no-forward-probe > $O/bench_tiny224_b256_train.json 2>/dev/null
python bench.py --img 384 --batch 64 --mode infer --no-cpu-baseline > $O/bench_base384_b64_infer.json 2>/dev/null
python bench.py --img 384 --batch 64 --no-cpu-baseline --no-forward-probe > $O/bench_base384_b64_train.json 2>/dev/null
for f in bench_train bench_train_one_range bench_infer bench_infer_one_stream bench_base224_train_torch_adamw bench_tiny224_b256_infer bench_tiny224_b256_train bench_base384_b64_infer bench_base384_b64_train; do echo "$f: $(tail -1 $O/$f.json | cut -c1-170)"; done
head -12 $O/train_step_breakdown.csv; head -8 $O/train_hbm_traffic.json; head -3 $O/two_stream.txt
