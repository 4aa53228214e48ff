#!/bin/bash
# gate_bn_kernel: rows per workgroup (-DGB_R) x rows per load batch (-DGB_U), summed over a forward from the step timeline
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
for fl in "" "-DGB_R=32" "-DGB_R=8" "-DGB_R=64" "-DGB_R=32 -DGB_U=16" "-DGB_R=16 -DGB_U=4"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $fl -c node.hip -o build/node.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  for rep in 1 2; do
    (cd ../..; bash tools/step_trace.sh > /dev/null; echo "[$fl]: $(grep gate_bn gpurun_out/step_trace.txt | awk '{s+=$4; printf "%s ", $4} END {printf "= %.1f us", s}')")
  done
done
(cd ../..; timeout 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "conv_layers or model_forward" 2>&1 | tail -1)
touch node.hip; make -j8 > /dev/null 2>&1
