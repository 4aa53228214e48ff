#!/bin/bash
# kernel-by-kernel timeline of the last forward of a short bench run: name, duration, gap to the previous kernel
#   bash tools/step_trace.sh            (writes gpurun_out/step_trace.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/st_prof
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/st_prof -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-extras --no-calibration --no-full-layers > /tmp/st_bench.log 2>&1
tail -1 /tmp/st_bench.log | cut -c1-200
python3 tools/step_trace.py $(find /tmp/st_prof -name '*kernel_trace.csv' | head -1) | tee gpurun_out/step_trace.txt
