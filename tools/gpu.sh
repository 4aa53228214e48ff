#!/bin/bash
# gpurun wrapper for this repository: records the commit the snapshot corresponds to (the GPU box has no .git) and passes
# everything on.   tools/gpu.sh --timeout 900 -- '<command>'
cd "$(dirname "$0")/.." || exit 1
c=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- . ':!gpurun_out' || c="$c-dirty"
echo "$c" > .head_commit
exec /usr/local/graft/bin/gpurun "$@"
