#!/bin/bash
# careful A/B: 16-channel entries for vector (l1 = 1) input blocks (TPF_MAX_COLS_L1 = 112) vs 8-channel ones
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
for c in ${L1_COLS:-112} 64; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. -DTPF_MAX_COLS_L1=$c -c tp_fused.hip -o build/tp_fused_$c.o 2>/dev/null
done
for rep in 1 2 3; do
  for c in ${L1_COLS:-112} 64; do
    cp build/tp_fused_$c.o build/tp_fused.o
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v 'tp_fused_') -o ../libmatten_hip.so
    if [ $rep = 1 ]; then (cd ../.. && MATTEN_TP_MAX_COLS_L1=$c python3 -m pytest tests -m gpu -x -q -k 'conv_layers or config3_fcc64 or tp_kernels' 2>&1 | tail -1); fi
    MATTEN_TP_MAX_COLS_L1=$c python3 ../../bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('rep $rep [l1 cols $c]: step %.3f ms' % d['ms_per_step'], 'tp', ' '.join('%.3f'%v for n,v in k.items() if n.startswith('tp')))"
  done
done
rm -f build/tp_fused_*.o; touch tp_fused.hip; make -j8 > /dev/null 2>&1
