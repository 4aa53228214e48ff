#!/bin/bash
# crystals/s and edge-TP/s of the bench forward against the batch size (same command as the headline, --crystals B): where the chip fills
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "bench.py --steps 20 --warmup 5 --crystals B --no-cpu-baseline --no-extras --no-calibration --no-full-layers (fcc-64: 64 atoms, 1152 edges per crystal)"
for B in 16 60 125 250 500 1000 2000 4000 8000; do
  python3 bench.py --steps 20 --warmup 5 --crystals $B --no-cpu-baseline --no-extras --no-calibration --no-full-layers 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms_per_launch']
print('B=%5d  %8.3f ms/step  %8.0f crystals/s  %7.1f M edge-TP/s   tp_fused %s   agg_linear %s' % ($B, d['ms_per_step'], d['crystals_per_sec'], d['value'] / 1e6, ' '.join('%.3f' % v for n, v in k.items() if n.startswith('tp_')), ' '.join('%.3f' % v for n, v in k.items() if n.startswith('agg'))))"
done
