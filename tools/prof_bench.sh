#!/bin/bash
# kernel-level profile of the default bench: prints the per-kernel stats table
cd /tmp && export TMPDIR=/tmp
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.log" 2>&1
tail -1 "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.log"
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof" -name "*kernel_stats.csv" | head -1)
cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/kernel_stats.csv"
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:9.1f} {r['Percentage']:>6s}%")
PY
