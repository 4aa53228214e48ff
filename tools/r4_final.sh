#!/bin/bash
# round-4 evidence in one call: rocprofv3 kernel stats + PMC passes (traffic, SQ), step trace, 2-rank rehearsal; then the
# bench line with the fresh PMC summaries in place
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
bash tools/round_profile.sh r04 > gpurun_out/r04_round_profile.log 2>&1
cp gpurun_out/traffic_r04.json profiles/hbm_traffic.json
cp gpurun_out/valu_r04.json profiles/tp_fused_valu.json
bash tools/step_trace.sh r04 > gpurun_out/r04_step_trace.txt 2>&1
timeout 600 python3 bench.py --gpus 2 --backend gloo --share-gpu --no-extras --steps 10 --warmup 3 2> gpurun_out/r04_bench_2rank_rehearsal.err | grep '^{"metric"' > gpurun_out/r04_bench_2rank_rehearsal.json; true 2> gpurun_out/r04_bench_2rank_rehearsal.err
timeout 900 python3 bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
cut -c1-250 gpurun_out/r04_bench.json; head -12 gpurun_out/r04_kernel_stats.csv | cut -c1-160
