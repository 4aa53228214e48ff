#!/bin/bash
# round-5 A/B harness: REPS alternating runs per flag set of the bench forward (no extras / CPU baseline / calibration),
# printing the step time and the per-layer tp_fused / agg_linear times; KINDS=1 adds the per-kind launch times once.
#   FLAGSETS="|-DTPF_X=1" REPS=2 KINDS=1 [SRC=agg_linear] bash tools/ab5.sh   (SRC: the source file the flag sets rebuild)
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
source ../../tools/_restore.sh
IFS='|' read -ra SETS <<< "${FLAGSETS:-|}"
i=0
for fl in "${SETS[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $fl -c ${SRC:-tp_fused}.hip -o build/${SRC:-tp_fused}_$i.o 2>&1 | grep -i "error" 
  i=$((i+1))
done
link() { cp build/${SRC:-tp_fused}_$1.o build/${SRC:-tp_fused}.o; hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "${SRC:-tp_fused}_" | grep -v calib) -o ../libmatten_hip.so; }
for rep in $(seq 1 ${REPS:-2}); do
  i=0
  for fl in "${SETS[@]}"; do
    link $i
    env ${ENVS:-A=1} python3 ../../bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-extras --no-calibration 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('rep $rep [$fl]: step %.3f ms  full-layers %.3f' % (d['ms_per_step'], d.get('ms_per_step_full_layers', 0)), 'tp', ' '.join('%.3f'%v for n,v in k.items() if n.startswith('tp')), ' agg', ' '.join('%.3f'%v for n,v in k.items() if n.startswith('agg')), ' radial %.3f' % k.get('radial_hidden_multi', 0))"
    i=$((i+1))
  done
done
if [ -n "$KINDS" ]; then
  i=0
  for fl in "${SETS[@]}"; do
    link $i
    for tg in ${KIND_TARGETS:-full view}; do
      echo "== kinds [$fl] TARGET=$tg"
      TARGET=$tg python3 ../../tools/fused_kind_bench.py 2>&1 | grep fused
    done
    i=$((i+1))
  done
fi
# (the production library is restored by the EXIT trap of tools/_restore.sh)
