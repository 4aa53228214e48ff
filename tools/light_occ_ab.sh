#!/bin/bash
# the light (l1 <= 1) kinds of tp_fused compiled alone at 3 / 4 / 5 workgroups per CU (168 / 128 / 96 VGPRs), with the
# wave tile trimmed to what the shared path needs so that the LDS allows it: is occupancy what they lack?
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
for mb in 3 4 5; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. -DMATTEN_LAB -DTPF_ONLY_LIGHT -DTPF_MIN_BLOCKS=$mb -c tp_fused.hip -o build/tp_fused.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  for lds in 1856 1344; do
    echo "== min blocks $mb, lds floats per wave $lds"
    LDS_PER_WAVE=$lds PK=0 python3 ../../tools/fused_kind_bench.py 2>&1 | grep "l1=[01]"
  done
done
touch tp_fused.hip; make -j8 > /dev/null 2>&1
