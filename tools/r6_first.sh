#!/bin/bash
# round-6 first evidence call: corrected counter collections of the bench forward (headline loop only: --no-full-layers), the training
# step's HBM traffic, and the A/B of compile-time coupling masks for the full layers (-DTPF_HOT_ALL=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/collect_traffic.sh r06 > gpurun_out/r06_collect_traffic.log 2>&1
bash tools/collect_valu.sh r06 > gpurun_out/r06_collect_valu.log 2>&1
bash tools/collect_train_traffic.sh r06 > gpurun_out/r06_train_traffic.txt 2>&1
rm -rf gpurun_out/pmc_* gpurun_out/valu_r06_1 gpurun_out/valu_r06_2
FLAGSETS="|-DTPF_HOT_ALL=1" REPS=2 STEPS=30 bash tools/ab5.sh > gpurun_out/r06_hot_all_ab.txt 2>&1
tail -4 gpurun_out/r06_collect_traffic.log | cut -c1-300
tail -3 gpurun_out/r06_collect_valu.log | cut -c1-400
cat gpurun_out/r06_train_traffic.txt | cut -c1-220
cat gpurun_out/r06_hot_all_ab.txt
