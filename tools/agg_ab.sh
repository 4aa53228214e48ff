#!/bin/bash
# A/B of matten_agg_linear build variants (chunks per block x workgroups per CU): bash tools/agg_ab.sh
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
# each configuration: "chunks-per-block min-waves-per-SIMD waves-per-workgroup"
IFS='|' read -ra CFGS <<< "${CFGS:-4 3 8|4 7 8|4 6 12|4 7 14|4 8 16}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  touch agg_linear.hip
  make -j8 EXTRA_CXXFLAGS="-DAL_BLK_CHUNKS=$1 -DAL_MIN_BLOCKS=$2 -DAL_WAVES_PER_WG=$3 $AL_FLAGS" > /dev/null 2>&1 || { echo "build failed $cfg"; continue; }
  echo "== chunks per block $1, min waves per SIMD $2, waves per workgroup $3 $AL_FLAGS"
  MATTEN_AGG_BLOCK=$1 python3 ../../tools/agg_bench.py 2>&1 | grep d_mid
done
touch agg_linear.hip; make -j8 > /dev/null 2>&1
