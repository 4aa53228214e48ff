#!/bin/bash
# ablations of the species_linear kernel on the lin2 micro-benchmark
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
for fl in "-DSL_ABLATE_NO_W -DSL_ABLATE_NO_MFMA" "-DSL_ABLATE_NO_W -DSL_ABLATE_NO_MFMA -DSL_ABLATE_COALESCED"; do
  for pad in 0 2; do
  echo "=== $fl pad=$pad"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include $fl -c species_linear.hip -o build/species_linear.o 2>/dev/null && \
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so && SL_PAD=$pad python3 ../../tools/sl_bench.py 2>/dev/null | tail -2
  done
done
make -B -j8 >/dev/null 2>&1
