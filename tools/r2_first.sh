#!/bin/bash
# round 2 first GPU call: atomic-add throughput ubench, per-kind timing of tp_fused, baseline bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R/tools/ubench && hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_rate.hip -o /tmp/atomic_rate 2>/dev/null
timeout 120 /tmp/atomic_rate > $R/gpurun_out/r2_atomic_rate.txt 2>&1
cd $R
timeout 300 python3 tools/fused_kind_bench.py > gpurun_out/r2_kind_bench.txt 2>&1
timeout 600 python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r2_bench0.json
cat gpurun_out/r2_atomic_rate.txt gpurun_out/r2_kind_bench.txt
cut -c1-400 gpurun_out/r2_bench0.json
