#!/bin/bash
# rocprofv3 kernel stats of the batch-2048 training step (tools/train_b2048.py: 5 + 10 + 3 = 18 steps)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_train2048
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train2048 -- python3 $R/tools/train_b2048.py > $R/gpurun_out/prof_train2048.log 2>&1
grep "ms per step" $R/gpurun_out/prof_train2048.log
f=$(find $R/gpurun_out/prof_train2048 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
N = 18
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {calls} launches ({N} steps) -> {tot/N/1e6:.2f} ms and {calls/N:.0f} launches per step")
for r in rows[:26]:
    print(f"{r['Name'][:72]:72s} calls/step {int(r['Calls'])/N:6.1f} ms/step {float(r['TotalDurationNs'])/N/1e6:7.3f} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
