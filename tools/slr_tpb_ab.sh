#!/bin/bash
# 16-row tiles per workgroup of species_linear_rows_kernel (weights copied once per workgroup, rows software-pipelined)
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
for tpb in ${TPBS:-2 3 4 6}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. -DSLR_TPB=$tpb -c species_linear_rows.hip -o build/species_linear_rows.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  for rep in 1 2; do
    (cd ../..; bash tools/step_trace.sh > /dev/null; echo "TPB=$tpb: $(grep rows_kernel gpurun_out/step_trace.txt | awk '{s+=$4; printf "%s ", $4} END {printf "= %.1f us", s}')")
  done
done
touch species_linear_rows.hip; make -j8 > /dev/null 2>&1
