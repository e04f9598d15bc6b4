#!/bin/bash
# Round-5 profile bundle (GPU box), everything at HEAD -> gpurun_out/r05/ : bench lines (train / inference), rocprofv3 kernel stats of the train command, per-step breakdown
# (weight gradients in line), the per-kernel table of a forward pass, HBM traffic per step, the PMC traffic of the kernel bench.py's roofline names (the weight-gradient GEMM) and of the
# forward Linear launches, per-kernel PMC tables, stage times / timelines, the CPU-baseline protocol and the stock-eager column.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
python bench.py > $O/bench_train.json 2> $O/bench_train.err
python bench.py --mode infer --no-cpu-baseline > $O/bench_infer.json 2>> $O/bench_train.err
python bench.py --mode infer --no-cpu-baseline --infer-parts 1 > $O/bench_infer_one_stream.json 2>> $O/bench_train.err
rocprofv3 --kernel-trace --stats -d $O/kt -o trace -- python bench.py --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench_under_rocprof.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $O/train_kernel_stats_full.csv
python tools/rocpd_stats.py $DB 400 > $O/train_kernel_stats_steady.csv
rm -rf $O/kt
LMV_SIDE_STREAM=0 LMV_TRAIN_PARTS=1 rocprofv3 --kernel-trace -d $O/kt2 -o trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench_inline_under_rocprof.log 2>&1
DB2=$(find $O/kt2 -name "*.db" | head -1)
python tools/step_breakdown.py $DB2 $O/bench_inline_under_rocprof.log 3 > $O/train_step_breakdown.csv
rm -rf $O/kt2
bash tools/prof_infer.sh $O/infer_kernel_stats.csv --infer-parts 1 > $O/prof_infer.log 2>&1
bash tools/pmc_traffic.sh $O/gemm_dw_pmc_traffic.json dw > gpurun_out/pmc_traffic_gemm_dw.log 2>&1
bash tools/pmc_traffic.sh $O/gemm_fwd_pmc_traffic.json > gpurun_out/pmc_traffic_gemm.log 2>&1
bash tools/pmc_step_traffic.sh $O/train_hbm_traffic.json > gpurun_out/pmc_traffic_train.log 2>&1
bash tools/pmc_step_traffic.sh $O/infer_hbm_traffic.json --mode infer --infer-parts 4 >> gpurun_out/pmc_traffic_train.log 2>&1
bash tools/pmc_kernels.sh $O/pmc_per_kernel_train.csv > gpurun_out/pmc_k_train.log 2>&1
bash tools/pmc_kernels.sh $O/pmc_per_kernel_infer.csv --mode infer --infer-parts 1 > gpurun_out/pmc_k_infer.log 2>&1
python tools/sstage_timeline.py 5 > $O/sstage_timeline.txt 2>&1
(python tools/dstage_timeline.py 1 128 4 192; python tools/dstage_timeline.py 1 128 4 96) > $O/dstage_timeline.txt 2>&1
(python tools/stage_times.py lemevit_base 128; python tools/stage_times.py lemevit_tiny 256) > $O/stage_times.txt 2>&1
python bench.py --model lemevit_tiny --batch 256 --mode infer --no-cpu-baseline > $O/bench_tiny224_b256_infer.json 2>/dev/null
python bench.py --img 384 --batch 64 --mode infer --no-cpu-baseline > $O/bench_base384_b64_infer.json 2>/dev/null
python tools/stock_eager.py > $O/stock_eager.txt 2>&1
python bench.py --cpu-baseline-protocol $O/cpu_baseline_protocol.json > /dev/null 2> $O/cpu_baseline_protocol.err
for f in bench_train bench_infer bench_infer_one_stream bench_tiny224_b256_infer bench_base384_b64_infer; do echo "$f: $(tail -1 $O/$f.json | cut -c1-170)"; done
head -12 $O/train_step_breakdown.csv; head -8 $O/train_hbm_traffic.json; cat $O/gemm_dw_pmc_traffic.json | head -8; tail -5 $O/stock_eager.txt
