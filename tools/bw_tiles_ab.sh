#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for T in "8 -DBW_WAVES_TARGET=2048" "16 -DBW_WAVES_TARGET=1024" "32 -DBW_WAVES_TARGET=512" "16 -DBW_WAVES_TARGET=2048"; do
  (cd matten_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -DBW_TILES_MAX_N=$T -c radial_mlp_bwd.hip -o build/radial_mlp_bwd.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so)
  echo "== BW_TILES_MAX=$T"; timeout 300 python3 tools/train_b2048.py 2>&1 | grep "ms per step\|radial_mlp_bwd"
done
(cd matten_amd/csrc && make -B build/radial_mlp_bwd.o > /dev/null 2>&1 && make > /dev/null 2>&1)
