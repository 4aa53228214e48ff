#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "conv_fused_kernel or conv_tile" > gpurun_out/r4b_pytest_tile.log 2>&1; echo "tile tests rc=$?"
tail -8 gpurun_out/r4b_pytest_tile.log
BLOCKS="2048" BLOCKS10="2048" timeout 300 python3 tools/tile_bench.py 2>&1 | grep -v amdgpu.ids
