#!/bin/bash
# A/B over compile flags of ONE source file on the batch-2048 training step (tools/train_b2048.py): step time and the HIP-event
# times of the tensor-product kernels.   FLAGSETS="|-DX=1" SRC=backward REPS=2 bash tools/ab_train.sh
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
source ../../tools/_restore.sh
SRC=${SRC:-backward}
IFS='|' read -ra SETS <<< "${FLAGSETS:-|}"
i=0
for fl in "${SETS[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $fl -c $SRC.hip -o build/${SRC}_$i.o 2>&1 | grep -i "error"
  i=$((i+1))
done
link() { cp build/${SRC}_$1.o build/$SRC.o; hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "${SRC}_" | grep -v calib) -o ../libmatten_hip.so; }
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for fl in "${SETS[@]}"; do
    link $i
    echo "rep $rep [$fl]: $(env ${ENVS:-A=1} timeout 300 python3 ../../tools/train_b2048.py ${BATCH:-2048} 2>/dev/null | grep 'ms per step\|tp_backward \|tp_scatter ' | tr '\n' ' ' | cut -c1-330)"
    i=$((i+1))
  done
done
# (the production library is restored by the EXIT trap of tools/_restore.sh)
