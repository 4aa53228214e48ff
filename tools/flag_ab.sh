#!/bin/bash
# A/B compile flags for the TP kernels on the GPU box:  bash tools/flag_ab.sh "<flags A>" "<flags B>" ...
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
for fl in "$@"; do
  echo "=== extra flags: [$fl]"
  for f in tp_block tp_fused tp_path; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include $fl -c $f.hip -o build/$f.o 2>/dev/null || echo "build failed $f"
  done
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  python3 ../../bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('step %.2f ms' % d['ms_per_step'], 'tp', ' '.join('%.2f'%v for n,v in k.items() if n.startswith('tp')))"
  MATTEN_TP_IMPL=blocks python3 ../../bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('blocks impl: step %.2f ms' % d['ms_per_step'], 'tp', ' '.join('%.2f'%v for n,v in k.items() if n.startswith('tp')), 'mlp', ' '.join('%.2f'%v for n,v in k.items() if n.startswith('radial')))"
done
