#!/bin/bash
# careful A/B of tp_fused.hip build flags: REPS alternating runs of the full bench per flag set, median step time
#   FLAGSETS="a|b" REPS=3 bash tools/ab_flags.sh
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
IFS='|' read -ra SETS <<< "${FLAGSETS:-|-DTPF_SETPRIO=3 -DTPF_SETPRIO_EARLY}"
i=0
for fl in "${SETS[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $fl -c tp_fused.hip -o build/tp_fused_$i.o 2>/dev/null
  i=$((i+1))
done
for rep in $(seq 1 ${REPS:-3}); do
  i=0
  for fl in "${SETS[@]}"; do
    cp build/tp_fused_$i.o build/tp_fused.o
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v 'tp_fused_') -o ../libmatten_hip.so
    python3 ../../bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('rep $rep [$fl]: step %.3f ms' % d['ms_per_step'], 'tp', ' '.join('%.3f'%v for n,v in k.items() if n.startswith('tp')))"
    i=$((i+1))
  done
done
rm -f build/tp_fused_*.o; touch tp_fused.hip; make -j8 > /dev/null 2>&1
