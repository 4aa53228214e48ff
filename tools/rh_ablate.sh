#!/bin/bash
# radial_hidden_kernel: full vs without its second-layer matrix instructions / silu evaluations / stores
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
for fl in ${FLAGSETS:-"" "-DMATTEN_LAB,-DRH_ABLATE_NO_SILU"}; do   # (the no-L1 / no-store switches of round 2 were removed with the lab clean-up) fl=${fl//,/ }
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $fl -c tp_fused.hip -o build/tp_fused.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  MATTEN_BENCH_NO_CHECK=1 python3 ../../bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('[$fl] radial_hidden %.4f ms' % k['radial_hidden'])"
done
make -B -j8 > /dev/null 2>&1
