#!/bin/bash
# throughput vs batch size: does keeping a layer's agg inside the 256 MB memory-side cache pay?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for B in 60 125 250 500 1000 2000; do
  python3 bench.py --steps 20 --warmup 5 --crystals $B --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('B=%5d  %.3f ms/step  %.0f crystals/s  tp_last %.3f ms  kernels %s' % ($B, d['ms_per_step'], d['crystals_per_sec'], d['roofline']['per_layer'][-1]['ms'], {k: round(v, 3) for k, v in d['kernel_ms_per_launch'].items()}))"
done
