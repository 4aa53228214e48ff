#!/bin/bash
# A/B of the node-tile size of the TP kernels (TILE_NODES in tp_*.hip, TP_TILE_NODES in plan.py) on the full bench
cd "$GRAFT_REPO_ROOT" || exit 1
for tn in 64 128 256 32; do
  sed -i "s/constexpr int TILE_NODES = [0-9]*;/constexpr int TILE_NODES = $tn;/" matten_amd/csrc/tp_block.hip matten_amd/csrc/tp_fused.hip matten_amd/csrc/tp_path.hip
  sed -i "s/^TP_TILE_NODES = [0-9]*/TP_TILE_NODES = $tn/" matten_amd/plan.py
  (cd matten_amd/csrc && make -j8 > /dev/null 2>&1)
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_launch']
print('tile $tn: step %.2f ms' % d['ms_per_step'], 'tp', ' '.join('%.2f'%v for n,v in k.items() if n.startswith('tp')))"
done
