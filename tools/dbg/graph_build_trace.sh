#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_build
python3 $R/tools/dbg/graph_build_time.py 2>&1 | grep "per build"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_build -- python3 $R/tools/dbg/graph_build_time.py > $R/gpurun_out/prof_build.log 2>&1
f=$(find $R/gpurun_out/prof_build -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
N = 46
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {calls} launches ({N} builds, two sets) -> {tot/N/1e6:.3f} ms and {calls/N:.0f} launches per build")
for r in rows[:22]:
    print(f"{r['Name'][:80]:80s} calls/build {int(r['Calls'])/N:6.1f} us/build {float(r['TotalDurationNs'])/N/1e3:8.1f} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
