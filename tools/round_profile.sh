#!/bin/bash
# Round-end evidence: kernel stats of the default bench (timeout 600 rocprofv3 --kernel-trace --stats) + HBM traffic PMC passes.
#   bash tools/round_profile.sh r01   -> gpurun_out/{r01_kernel_stats.csv, r01_bench.json, traffic_r01.json}
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-calibration > $R/gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
python3 $R/bench.py 2>/dev/null | tail -1 > $R/gpurun_out/${TAG}_bench.json
bash $R/tools/collect_traffic.sh $TAG
bash $R/tools/collect_valu.sh $TAG
head -8 $R/gpurun_out/${TAG}_kernel_stats.csv
cat $R/gpurun_out/${TAG}_bench.json | cut -c1-600
