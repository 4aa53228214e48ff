#!/bin/bash
# A/B harness for the block-fused TP kernel: group scheme x waves-per-SIMD cap. Run on the GPU box.
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
for variant in "$@"; do
  scheme=${variant%%:*}; waves=${variant##*:}
  export MATTEN_TP_GROUPS=$scheme
  python3 gen_cg.py > cg_gen.h
  if [ "$waves" != "0" ]; then
    sed -i "s/__launch_bounds__(WAVES_PER_BLOCK \* 64[, 0-9]*) void tp_block_kernel/__launch_bounds__(WAVES_PER_BLOCK * 64, $waves) void tp_block_kernel/" tp_block.hip
  else
    sed -i "s/__launch_bounds__(WAVES_PER_BLOCK \* 64[, 0-9]*) void tp_block_kernel/__launch_bounds__(WAVES_PER_BLOCK * 64) void tp_block_kernel/" tp_block.hip
  fi
  make -j8 > /dev/null 2>&1 || { echo "build failed $variant"; continue; }
  python3 ../../bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('$variant', 'step %.2f ms' % d['ms_per_step'], 'tp', ' '.join('%.2f'%v for n,v in k.items() if n.startswith('tp')), 'mlp', ' '.join('%.2f'%v for n,v in k.items() if n.startswith('radial')))"
done
