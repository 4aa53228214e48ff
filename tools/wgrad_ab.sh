#!/bin/bash
# tile-block edge of species_linear_wgrad_kernel (registers vs passes over the rows), batch-2048 training step
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
for tb in ${TBS:-4 2 1}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. -DMATTEN_WG_TB=$tb -c backward.hip -o build/backward.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  echo "== WG_TB=$tb"
  BATCH=2048 bash ../../tools/prof_train.sh 2>&1 | cut -c1-150 | grep -E "total kernel|wgrad"
done
make -B -j8 > /dev/null 2>&1
