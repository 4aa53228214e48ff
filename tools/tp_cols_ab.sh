#!/bin/bash
# tp_fused: entry width (weight columns per entry = A-operand registers) x min blocks per CU x l2-group scheme
#   tools/tp_cols_ab.sh A:64:3 A:32:3 A:32:4 B:32:4 A:64:3:96   (scheme : columns : blocks [: columns of l1 = 0 entries])
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
for variant in "$@"; do
  IFS=: read scheme cols mb cols0 <<< "$variant"
  cols0=${cols0:-$cols}
  export MATTEN_TP_GROUPS=$scheme MATTEN_TP_MAX_COLS=$cols MATTEN_TP_MAX_COLS_L0=$cols0
  python3 gen_cg.py > cg_gen.h
  rm -f build/tp_fused.o build/tp_block.o
  make -j8 EXTRA_CXXFLAGS="-DTPF_MIN_BLOCKS=$mb -DTPF_MAX_COLS=$cols -DTPF_MAX_COLS_L0=$cols0" > /dev/null 2>&1 || { echo "build failed $variant"; continue; }
  python3 ../../bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('$variant', 'step %.2f ms' % d['ms_per_step'], 'tp', ' '.join('%.2f'%v for n,v in k.items() if n.startswith('tp')))"
done
