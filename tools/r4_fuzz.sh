#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
echo "== conv_tile forced, poisoned buffers"
MATTEN_CONV_TILE=1 MATTEN_CONV_TILE_MIN_ROWS=0 MATTEN_CONV_TILE_MIN_DMID=0 MATTEN_CONV_TILE_BLOCK=32 NAN_EMPTY=1 timeout 900 python3 tests/fuzz_models.py 60 11 2>&1 | tail -25
echo "== hub split stress (pieces of 3 edges)"
MATTEN_HUB_SPLIT_LEN=3 NAN_EMPTY=1 timeout 600 python3 tests/fuzz_models.py 40 12 2>&1 | tail -6
