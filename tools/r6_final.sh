#!/bin/bash
# round-6 evidence in one gpurun call (production library as shipped): counter passes of the headline loop (HBM traffic, SQ
# instructions: per layer), rocprofv3 kernel stats, step trace, the training step's kernel stats, the 2-rank rehearsal, the full
# default bench line (CPU baseline + extras) -> gpurun_out/r06_*
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/collect_traffic.sh r06 > gpurun_out/r06_collect_traffic.log 2>&1
bash tools/collect_valu.sh r06 > gpurun_out/r06_collect_valu.log 2>&1
rm -rf gpurun_out/pmc_r06_* gpurun_out/valu_r06_[12]
# the bench line reads the committed summaries: put this run's in place before the line is taken
cp gpurun_out/traffic_r06.json profiles/hbm_traffic.json
cp gpurun_out/valu_r06.json profiles/tp_fused_valu.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r06
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06 -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-calibration --no-full-layers > $R/gpurun_out/r06_bench_under_rocprof.log 2>&1
cp "$(find $R/gpurun_out/prof_r06 -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/r06_kernel_stats.csv
rm -rf $R/gpurun_out/prof_r06
cd $R
bash tools/step_trace.sh > gpurun_out/r06_step_trace.txt 2>&1
bash tools/prof_train_b2048.sh > gpurun_out/r06_train_b2048_profile.txt 2>&1
rm -rf gpurun_out/prof_train2048
python3 bench.py --gpus 2 --backend gloo --share-gpu --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r06_bench_2rank_rehearsal.json
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/r06_bench.json
head -6 gpurun_out/r06_kernel_stats.csv | cut -c1-200
tail -4 gpurun_out/r06_step_trace.txt
head -8 gpurun_out/r06_train_b2048_profile.txt | cut -c1-200
cut -c1-400 gpurun_out/r06_bench_2rank_rehearsal.json
cut -c1-700 gpurun_out/r06_bench.json
