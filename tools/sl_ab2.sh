#!/bin/bash
# species_linear build variants on the lin2 micro-benchmark (tools/sl_bench.py)
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
IFS='|' read -ra SETS <<< "${FLAGSETS:-|-DSL_ABL_NO_MFMA}"
for fl in "${SETS[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $fl -c species_linear.hip -o build/species_linear.o 2>/dev/null || { echo "[$fl] build failed"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  echo "== [$fl]"
  python3 ../../tools/sl_bench.py 2>&1 | grep -v amdgpu.ids
done
touch species_linear.hip; make -j8 > /dev/null 2>&1
