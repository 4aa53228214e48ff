#!/bin/bash
# A/B of extra compiler flags on tp_fused.hip (full bench)
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
run() {
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $1 -c tp_fused.hip -o build/tp_fused.o 2>/dev/null || { echo "[$1] build failed"; return; }
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  python3 ../../bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('[$1]: step %.2f ms' % d['ms_per_step'], 'tp', ' '.join('%.2f'%v for n,v in k.items() if n.startswith('tp')))"
}
run ""
run "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1"
run "-mllvm -enable-post-misched=0"
run "-mllvm -amdgpu-schedule-relaxed-occupancy=1"
run "-mllvm -greedy-reverse-local-assignment=1"
run "-mllvm -amdgpu-early-inline-all=true -mllvm -amdgpu-function-calls=false"
run "-O2"
