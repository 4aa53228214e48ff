#!/bin/bash
# gpurun wrapper for this repository: records the commit the snapshot corresponds to (the GPU box has no .git), retries while
# no GPU slot is free (exit code 3: nothing charged) and passes everything on.   tools/gpu.sh --timeout 900 -- '<command>'
cd "$(dirname "$0")/.." || exit 1
c=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- . ':!gpurun_out' || c="$c-dirty"
echo "$c" > .head_commit
for try in $(seq 1 ${GPU_RETRIES:-12}); do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
