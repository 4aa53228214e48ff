#!/bin/bash
# kernel-by-kernel timeline of one n100 forward (eager and hipGraph replay): bash tools/dbg/n100_timeline.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for MODE in eager graph; do
  rm -rf /tmp/n100tl_$MODE
  N100_MODE=$MODE timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/n100tl_$MODE -o t -- python3 $R/tools/dbg/n100_forward.py > /tmp/n100tl_$MODE.log 2>&1
  f=$(find /tmp/n100tl_$MODE -name "*kernel_trace.csv" | head -1)
  echo "==== $MODE"; python3 $R/tools/step_trace.py "$f"
done
