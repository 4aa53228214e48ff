#!/bin/bash
# per-group-kind time of tp_fused (last layer) for the full kernel and its ablations
#   FLAGSETS="a|b|c" bash tools/fused_kind_ablate.sh   (each set = extra hipcc flags)
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
IFS='|' read -ra SETS <<< "${FLAGSETS:-|-DMATTEN_LAB -DMATTEN_ABLATE_NO_H2LOAD|-DMATTEN_LAB -DMATTEN_ABLATE_NO_MFMA|-DMATTEN_LAB -DMATTEN_ABLATE_NO_VALU|-DMATTEN_LAB -DMATTEN_ABLATE_NO_STORE|-DMATTEN_LAB -DMATTEN_ABLATE_NO_VALU -DMATTEN_ABLATE_NO_STORE}"
for fl in "${SETS[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $fl -c tp_fused.hip -o build/tp_fused.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  echo "== [$fl]"
  python3 ../../tools/fused_kind_bench.py 2>&1 | grep fused
done
make -B -j8 > /dev/null 2>&1
