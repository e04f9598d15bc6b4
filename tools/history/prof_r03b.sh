#!/bin/bash
# Round-3 profile bundle, part 2 (GPU box): the train-step pieces with the forward probe excluded, and the other configs -> gpurun_out/r03/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
bash tools/pmc_traffic.sh $O/gemm_fwd_pmc_traffic.json > $O/pmc_traffic.log 2>&1
cp $O/gemm_fwd_pmc_traffic.json profiles/r03_gemm_fwd_pmc_traffic.json
python bench.py > $O/bench_train.json 2> $O/bench_train.err
rocprofv3 --kernel-trace --stats -d $O/kt -o trace -- python bench.py --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench_under_rocprof.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > $O/train_kernel_stats_full.csv
python tools/rocpd_stats.py $DB 400 > $O/train_kernel_stats_steady.csv
python tools/rocpd_by_grid.py $DB 400 > $O/train_kernel_stats_by_grid.csv
rm -rf $O/kt
LMV_SIDE_STREAM=0 rocprofv3 --kernel-trace -d $O/kt2 -o trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-issue-probe --no-forward-probe > $O/bench_inline_under_rocprof.log 2>&1
DB2=$(find $O/kt2 -name "*.db" | head -1)
python tools/step_breakdown.py $DB2 $O/bench_inline_under_rocprof.log 3 > $O/train_step_breakdown.csv
rm -rf $O/kt2
bash tools/pmc_kernels.sh $O/pmc_per_kernel_train.csv > gpurun_out/pmc_k_train.log 2>&1
bash tools/pmc_step_traffic.sh $O/train_hbm_traffic.json > gpurun_out/pmc_traffic_train.log 2>&1
LMV_MLP_SPLIT384=0 bash tools/pmc_step_traffic.sh $O/infer_hbm_traffic_onekernel_mlp.json --mode infer >> gpurun_out/pmc_traffic_train.log 2>&1
LMV_MLP_SPLIT384=0 python bench.py --mode infer --no-cpu-baseline > $O/bench_infer_onekernel_mlp.json 2>/dev/null
python bench.py --torch-adamw --no-cpu-baseline --no-forward-probe > $O/bench_base224_train_torch_adamw.json 2>/dev/null
python bench.py --model lemevit_tiny --batch 256 --mode infer --no-cpu-baseline > $O/bench_tiny224_b256_infer.json 2>/dev/null
python bench.py --model lemevit_tiny --batch 256 --no-cpu-baseline --no-forward-probe > $O/bench_tiny224_b256_train.json 2>/dev/null
python bench.py --img 384 --batch 64 --mode infer --no-cpu-baseline > $O/bench_base384_b64_infer.json 2>/dev/null
python bench.py --img 384 --batch 64 --no-cpu-baseline --no-forward-probe > $O/bench_base384_b64_train.json 2>/dev/null
for f in bench_train bench_base224_train_torch_adamw bench_tiny224_b256_infer bench_tiny224_b256_train bench_base384_b64_infer bench_base384_b64_train bench_infer_onekernel_mlp; do echo "$f: $(tail -1 $O/$f.json | cut -c1-150)"; done
head -12 $O/train_step_breakdown.csv; cat $O/train_hbm_traffic.json | head -8
