#!/bin/bash
# tp_lin2_kernel variants on the full bench: per-launch times of the conv kernels + step time
#   FLAGSETS="a|b|c" bash tools/lin2_ablate.sh   (each set = extra hipcc flags for tp_fused.hip)
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
IFS='|' read -ra SETS <<< "${FLAGSETS:-|-DMATTEN_ABLATE_NO_EPI}"
for fl in "${SETS[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $fl -c tp_fused.hip -o build/tp_fused.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  echo "== [$fl]"
  MATTEN_BENCH_NO_CHECK=1 python3 ../../bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  step %.3f ms' % d['ms_per_step'], ' '.join('%s=%.3f' % (k.replace('tp_scatter/', 'H/').replace('tp_lin2/', 'L/'), v) for k, v in d['kernel_ms_per_launch'].items()))"
done
touch tp_fused.hip; make -j8 > /dev/null 2>&1
