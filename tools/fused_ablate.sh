#!/bin/bash
# ablations of tp_fused on the full bench
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
IFS="|" read -ra SETS <<< "${FLAGSETS:-|-DMATTEN_LAB -DMATTEN_ABLATE_NO_H2LOAD|-DMATTEN_LAB -DMATTEN_ABLATE_NO_MFMA|-DMATTEN_LAB -DMATTEN_ABLATE_NO_VALU|-DMATTEN_LAB -DMATTEN_ABLATE_NO_STORE|-DMATTEN_LAB -DMATTEN_ABLATE_NO_VALU -DMATTEN_ABLATE_NO_STORE}"
for fl in "${SETS[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $fl -c tp_fused.hip -o build/tp_fused.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  MATTEN_BENCH_NO_CHECK=1 python3 ../../bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('[$fl]: step %.2f ms' % d['ms_per_step'], 'tp', ' '.join('%.2f'%v for n,v in k.items() if n.startswith('tp')))"
done
make -B -j8 > /dev/null 2>&1
