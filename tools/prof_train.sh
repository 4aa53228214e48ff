#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_train"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_train" -- python3 "$GRAFT_REPO_ROOT/tools/train_bench.py" > "$GRAFT_REPO_ROOT/gpurun_out/prof_train.log" 2>&1
tail -1 "$GRAFT_REPO_ROOT/gpurun_out/prof_train.log"
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof_train" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {calls} launches (39 steps) -> {tot/39/1e6:.2f} ms and {calls/39:.0f} launches per step")
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
