#!/bin/bash
# conv_tile per-phase ablations (lab builds): bash tools/tile_ablate.sh  -> gpurun_out/tile_ablate.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
build() { (cd matten_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include $1 -c conv_tile.hip -o build/conv_tile.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so); }
: > gpurun_out/tile_ablate.log
IFS='|' read -ra SETS <<< "${FLAGSETS:-|-DMATTEN_LAB -DCT_ABLATE_NO_ALOAD|-DMATTEN_LAB -DCT_ABLATE_NO_LIN2|-DCT_MAXF_N=12|-DCT_MAXF_N=16}"
for FL in "${SETS[@]}"; do
  build "$FL"
  echo "=== flags: $FL" >> gpurun_out/tile_ablate.log
  MATTEN_BENCH_NO_CHECK=1 BLOCKS="2048" BLOCKS10="2048" timeout 300 python3 tools/tile_bench.py 2>&1 | grep -v "amdgpu.ids\|two-kernel, 1 sp" >> gpurun_out/tile_ablate.log
done
build ""
cat gpurun_out/tile_ablate.log
