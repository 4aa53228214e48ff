#!/bin/bash
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
for flag in "" "-DMATTEN_ABLATE_NO_VALU" "-DMATTEN_ABLATE_NO_MFMA"; do
  echo "=== flags: $flag"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include $flag -c tp_fused.hip -o build/tp_fused.o 2>/dev/null && \
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so && python3 ../../tools/tp_bench.py 2>/dev/null | grep fused
done
