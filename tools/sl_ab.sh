#!/bin/bash
# A/B of extra compiler flags on species_linear.hip: per-call durations of the species-linear launches + step time
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
make -j8 > /dev/null 2>&1
IFS='|' read -ra SETS <<< "${FLAGSETS:-|-DSL_DUMMY_RESIDENT}"
for fl in "${SETS[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $fl -c species_linear.hip -o build/species_linear.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  echo "== [$fl]"
  python3 ../../bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('  step %.3f ms' % d['ms_per_step'])"
  bash ../../tools/sl_percall.sh
done
touch species_linear.hip; make -j8 > /dev/null 2>&1
