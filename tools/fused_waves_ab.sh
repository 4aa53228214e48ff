#!/bin/bash
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
for wv in 0 3 4; do
  if [ $wv = 0 ]; then cp tp_fused.hip /tmp/tpf_v.hip; else sed "s/__global__ __launch_bounds__(WAVES_PER_BLOCK \* 64) void tp_fused_kernel/__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, $wv) void tp_fused_kernel/" tp_fused.hip > /tmp/tpf_v.hip; fi
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. -c /tmp/tpf_v.hip -o build/tp_fused.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  python3 ../../bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('waves $wv: step %.2f ms' % d['ms_per_step'], 'tp', ' '.join('%.2f'%v for n,v in k.items() if n.startswith('tp')))"
done
