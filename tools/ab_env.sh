#!/bin/bash
# A/B over ENVIRONMENT settings on ONE build (REPS alternating runs of the bench forward, no extras / CPU baseline /
# calibration): step time, full-layer step time, per-layer tp_fused and agg_linear times.
#   ENVSETS="A=1|MATTEN_TP_PERSIST=16:4,8:2" REPS=2 [FLAGS="-D..."] bash tools/ab_env.sh
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
if [ -n "$FLAGS" ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $FLAGS -c tp_fused.hip -o build/tp_fused.o 2>&1 | grep -i error
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v calib) -o ../libmatten_hip.so
fi
IFS='|' read -ra SETS <<< "${ENVSETS:-A=1}"
for rep in $(seq 1 ${REPS:-2}); do
  for ev in "${SETS[@]}"; do
    env $ev python3 ../../bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-extras --no-calibration 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('rep $rep [$ev]: step %.3f ms  full-layers %.3f' % (d['ms_per_step'], d.get('ms_per_step_full_layers', 0)), 'tp', ' '.join('%.3f'%v for n,v in k.items() if n.startswith('tp')), ' agg', ' '.join('%.3f'%v for n,v in k.items() if n.startswith('agg')))"
  done
done
if [ -n "$FLAGS" ]; then touch tp_fused.hip; make -j8 > /dev/null 2>&1; fi
