#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "hub_segments" > gpurun_out/r4e_pytest_hub.log 2>&1; echo "hub test rc=$?"; tail -4 gpurun_out/r4e_pytest_hub.log
for L in 8 12 16 24; do echo "L=$L"; MATTEN_HUB_SPLIT_LEN=$L bash tools/dbg/n100_timeline.sh 2>&1 | grep "kernels "; done
