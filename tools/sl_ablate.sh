#!/bin/bash
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 >/dev/null 2>&1
for flag in "" "-DSL_ABLATE_NO_STAGE" "-DSL_ABLATE_NO_MFMA" "-DSL_ABLATE_NO_STAGE -DSL_ABLATE_NO_MFMA"; do
  echo "=== flags: $flag"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include $flag -c node.hip -o build/node.o 2>/dev/null && \
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so && python3 ../../tools/sl_bench.py 2>/dev/null | tail -1
done
