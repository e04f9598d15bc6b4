#!/bin/bash
# Round-3 profile bundle (GPU box): the round bundle + hipBLASLt table + per-kernel PMC tables + train-step HBM traffic -> gpurun_out/r03/
cd $GRAFT_REPO_ROOT
bash tools/prof_round.sh r03 > gpurun_out/prof_round.log 2>&1
python tools/hipblaslt_table.py > gpurun_out/hipblaslt.log 2>&1; cp gpurun_out/hipblaslt_table.txt gpurun_out/r03/hipblaslt_table.txt
bash tools/pmc_kernels.sh gpurun_out/r03/pmc_per_kernel_train.csv > gpurun_out/pmc_k_train.log 2>&1
bash tools/pmc_kernels.sh gpurun_out/r03/pmc_per_kernel_infer.csv --mode infer > gpurun_out/pmc_k_infer.log 2>&1
bash tools/pmc_step_traffic.sh gpurun_out/r03/train_hbm_traffic.json > gpurun_out/pmc_traffic_train.log 2>&1
bash tools/run_two_stream.sh > gpurun_out/r03/two_stream.txt 2>&1
python tools/quant_probe.py wn > gpurun_out/r03/wn_probe.txt 2>&1; python tools/cold_probe.py >> gpurun_out/r03/wn_probe.txt 2>&1
ls gpurun_out/r03; tail -1 gpurun_out/r03/bench_train.json | cut -c1-400; cat gpurun_out/r03/hipblaslt_table.txt
