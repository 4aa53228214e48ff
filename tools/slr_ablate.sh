#!/bin/bash
# species_linear_rows_kernel: full vs fill only (no items) vs no stores vs a scalar FMA in place of every matrix instruction; per-call durations of one forward
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
for fl in "" "-DSLR_ABLATE_NO_ITEMS" "-DSLR_ABLATE_NO_STORE" "-DSLR_ABLATE_NO_MFMA"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $fl -c species_linear_rows.hip -o build/species_linear_rows.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  echo "== [$fl]"; MATTEN_BENCH_NO_CHECK=1 bash ../../tools/sl_percall.sh
done
make -B -j8 > /dev/null 2>&1
