#!/bin/bash
# Round-4 profile bundle (GPU box), everything at HEAD -> gpurun_out/r04/ : bench lines (train, inference with 4 / 1 sub-batch streams, the per-block
# inference schedule without the persistent S stage, the other configs), rocprofv3 kernel stats of the train command, per-step breakdown, the per-kernel
# table of a forward pass, HBM traffic per step (train / forward), the PMC traffic of the forward Linear launches that bench.py quotes, per-kernel PMC tables.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
python bench.py > $O/bench_train.json 2> $O/bench_train.err
python bench.py --mode infer --no-cpu-baseline > $O/bench_infer.json 2>> $O/bench_train.err
python bench.py --mode infer --no-cpu-baseline --infer-parts 1 > $O/bench_infer_one_stream.json 2>> $O/bench_train.err
LMV_SSTAGE=0 python bench.py --mode infer --no-cpu-baseline > $O/bench_infer_per_block_schedule.json 2>> $O/bench_train.err
LMV_DSTAGE=0 python bench.py --mode infer --no-cpu-baseline > $O/bench_infer_no_dstage.json 2>> $O/bench_train.err
LMV_STEM=0 python bench.py --mode infer --no-cpu-baseline > $O/bench_infer_no_stem.json 2>> $O/bench_train.err
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
bash tools/pmc_traffic.sh $O/gemm_fwd_pmc_traffic.json > gpurun_out/pmc_traffic_gemm.log 2>&1
bash tools/pmc_step_traffic.sh $O/train_hbm_traffic.json > gpurun_out/pmc_traffic_train.log 2>&1
bash tools/pmc_step_traffic.sh $O/infer_hbm_traffic.json --mode infer --infer-parts 4 >> gpurun_out/pmc_traffic_train.log 2>&1
bash tools/pmc_kernels.sh $O/pmc_per_kernel_train.csv > gpurun_out/pmc_k_train.log 2>&1
bash tools/pmc_kernels.sh $O/pmc_per_kernel_infer.csv --mode infer --infer-parts 1 > gpurun_out/pmc_k_infer.log 2>&1
python bench.py --model lemevit_tiny --batch 256 --mode infer --no-cpu-baseline > $O/bench_tiny224_b256_infer.json 2>/dev/null
python bench.py --model lemevit_tiny --batch 256 --no-cpu-baseline --no-forward-probe > $O/bench_tiny224_b256_train.json 2>/dev/null
python bench.py --img 384 --batch 64 --mode infer --no-cpu-baseline > $O/bench_base384_b64_infer.json 2>/dev/null
python bench.py --img 384 --batch 64 --no-cpu-baseline --no-forward-probe > $O/bench_base384_b64_train.json 2>/dev/null
python tools/sstage_timeline.py 5 > $O/sstage_timeline.txt 2>&1
(python tools/stem_timeline.py 48 96 128; python tools/stem_timeline.py 32 64 256) > $O/stem_timeline.txt 2>&1
(python tools/dstage_timeline.py 1 128 4 192; python tools/dstage_timeline.py 1 128 4 96; python tools/dstage_timeline.py 1 256 2 128; python tools/dstage_timeline.py 1 256 2 64; python tools/dstage_timeline.py 1 64 18 384) > $O/dstage_timeline.txt 2>&1
(python tools/stage_times.py lemevit_base 128; python tools/stage_times.py lemevit_tiny 256; python tools/stage_times.py lemevit_small 128; python tools/stage_times.py lemevit_base 64 384) > $O/stage_times.txt 2>&1
python bench.py --model lemevit_tiny --batch 256 --mode infer --no-cpu-baseline --infer-parts 1 > $O/bench_tiny224_b256_infer_one_stream.json 2>/dev/null
python bench.py --model lemevit_small --mode infer --no-cpu-baseline > $O/bench_small224_b128_infer.json 2>/dev/null
bash tools/prof_infer.sh $O/tiny_infer_kernel_stats.csv --infer-parts 1 --model lemevit_tiny --batch 256 > $O/prof_infer_tiny.log 2>&1
for f in bench_train bench_infer bench_infer_one_stream bench_infer_per_block_schedule bench_infer_no_dstage bench_infer_no_stem bench_tiny224_b256_infer bench_tiny224_b256_infer_one_stream bench_small224_b128_infer bench_tiny224_b256_train bench_base384_b64_infer bench_base384_b64_train; do echo "$f: $(tail -1 $O/$f.json | cut -c1-170)"; done
head -12 $O/train_step_breakdown.csv; head -8 $O/train_hbm_traffic.json; head -6 $O/infer_hbm_traffic.json; cat $O/gemm_fwd_pmc_traffic.json | head -5
