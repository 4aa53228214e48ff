#!/bin/bash
# round-5 evidence in one gpurun call: rocprofv3 kernel stats + HBM traffic + SQ instruction passes of the bench forward
# (tools/round_profile.sh), the step trace, the per-phase trace of tp_fused waves, the 2-rank rehearsal -> gpurun_out/
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/round_profile.sh r05 > gpurun_out/r05_round_profile.log 2>&1
bash tools/step_trace.sh > gpurun_out/r05_step_trace.txt 2>&1
python3 bench.py --gpus 2 --backend gloo --share-gpu --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r05_bench_2rank_rehearsal.json
TARGETS="model" bash tools/tp_trace.sh > gpurun_out/r05_tp_fused_phase_trace_after.txt 2>&1
tail -3 gpurun_out/r05_round_profile.log | cut -c1-400
tail -5 gpurun_out/r05_step_trace.txt
cut -c1-300 gpurun_out/r05_bench_2rank_rehearsal.json
