#!/bin/bash
# conv_tile per-phase ablations (lab builds): bash tools/tile_ablate.sh  -> gpurun_out/tile_ablate.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
build() { (cd matten_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include $1 -c conv_tile.hip -o build/conv_tile.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so); }
: > gpurun_out/tile_ablate.log
for FL in "" "-DMATTEN_LAB -DCT_ABLATE_NO_LIN2" "-DMATTEN_LAB -DCT_ABLATE_NO_LIN2 -DCT_ABLATE_NO_DUMP" "-DMATTEN_LAB -DMATTEN_ABLATE_NO_VALU" "-DMATTEN_LAB -DMATTEN_ABLATE_NO_GATHER" "-DMATTEN_LAB -DMATTEN_ABLATE_NO_H2LOAD" "-DCT_MIN_BLOCKS=2"; do
  build "$FL"
  echo "=== flags: $FL" >> gpurun_out/tile_ablate.log
  MATTEN_BENCH_NO_CHECK=1 BLOCKS="64,2048" BLOCKS10="2048" timeout 300 python3 tools/tile_bench.py >> gpurun_out/tile_ablate.log 2>&1
done
build ""
cat gpurun_out/tile_ablate.log
