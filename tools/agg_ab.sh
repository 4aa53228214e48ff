#!/bin/bash
# A/B of matten_agg_linear build variants (chunks per block x workgroups per CU): bash tools/agg_ab.sh
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
for cfg in "4 3" "4 2" "6 3" "8 2" $EXTRA_CFGS; do
  set -- $cfg
  touch agg_linear.hip
  make -j8 EXTRA_CXXFLAGS="-DAL_BLK_CHUNKS=$1 -DAL_MIN_BLOCKS=$2 $AL_FLAGS" > /dev/null 2>&1 || { echo "build failed $cfg"; continue; }
  echo "== chunks per block $1, workgroups per CU $2 $AL_FLAGS"
  MATTEN_AGG_BLOCK=$1 python3 ../../tools/agg_bench.py 2>&1 | grep d_mid
done
touch agg_linear.hip; make -j8 > /dev/null 2>&1
