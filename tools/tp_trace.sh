#!/bin/bash
# per-phase cycle breakdown of tp_fused waves by group kind (lab build with -DTPF_TRACE); restores the production build
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
source ../../tools/_restore.sh
fl="-DMATTEN_LAB -DTPF_TRACE ${EXTRA:-}"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $fl -c tp_fused.hip -o build/tp_fused.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
for tg in ${TARGETS:-full view l1}; do
  TARGET=$tg python3 ../../tools/tp_trace.py 2>&1 | grep -v Warning
done
# (the production library is restored by the EXIT trap of tools/_restore.sh)
