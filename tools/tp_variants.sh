#!/bin/bash
# A/B harness for the fused TP kernel: l2-group scheme (plan.py _GROUP_SCHEMES) x min blocks per CU. Run on the GPU box.
#   tools/tp_variants.sh A:3 B:3 B:4 C:4
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
for variant in "$@"; do
  scheme=${variant%%:*}; mb=${variant##*:}
  export MATTEN_TP_GROUPS=$scheme
  python3 gen_cg.py > cg_gen.h
  rm -f build/tp_fused.o build/tp_block.o
  make -j8 EXTRA_CXXFLAGS="-DTPF_MIN_BLOCKS=$mb" > /dev/null 2>&1 || { echo "build failed $variant"; continue; }
  python3 ../../bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('$variant', 'step %.2f ms' % d['ms_per_step'], 'tp', ' '.join('%.2f'%v for n,v in k.items() if n.startswith('tp')))"
done
