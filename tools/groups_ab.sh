#!/bin/bash
# A/B of coupling-group schemes (plan.py _GROUP_SCHEMES; cg_gen.h is generated per scheme): alternating runs of the bench forward
#   SCHEMES="A D" REPS=2 bash tools/groups_ab.sh
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
source ../../tools/_restore.sh
for sch in ${SCHEMES:-A D}; do
  MATTEN_TP_GROUPS=$sch python3 gen_cg.py > cg_gen.h
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. -c tp_fused.hip -o build/tp_fused_$sch.o 2>&1 | grep -i error
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. -c tp_path.hip -o build/tp_path_$sch.o 2>&1 | grep -i error
done
for rep in $(seq 1 ${REPS:-2}); do
  for sch in ${SCHEMES:-A D}; do
    cp build/tp_fused_$sch.o build/tp_fused.o; cp build/tp_path_$sch.o build/tp_path.o
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "_[A-E]\.o" | grep -v calib) -o ../libmatten_hip.so
    MATTEN_TP_GROUPS=$sch python3 ../../bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-extras --no-calibration 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('rep $rep [scheme $sch]: step %.3f ms  full-layers %.3f' % (d['ms_per_step'], d.get('ms_per_step_full_layers', 0)), 'tp', ' '.join('%.3f'%v for n,v in k.items() if n.startswith('tp')))"
  done
done
if [ -n "$TEST_SCHEME" ]; then
  cp build/tp_fused_$TEST_SCHEME.o build/tp_fused.o; cp build/tp_path_$TEST_SCHEME.o build/tp_path.o
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "_[A-E]\.o" | grep -v calib) -o ../libmatten_hip.so
  cd ../..; MATTEN_TP_GROUPS=$TEST_SCHEME python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tp_kernels or conv_layers or golden or independence or isolated or dead_output" 2>&1 | tail -4; cd matten_amd/csrc
fi
# (the production library is restored by the EXIT trap of tools/_restore.sh)
