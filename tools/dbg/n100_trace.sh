#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/n100prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/n100prof -- python3 "$GRAFT_REPO_ROOT/tools/dbg/n100_forward.py" > /tmp/n100.log 2>&1
f=$(find /tmp/n100prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per forward %.3f ms, launches per forward %.1f" % (tot/30/1e6, sum(int(r['Calls']) for r in rows)/30))
for r in rows[:14]:
    print("%5.1f calls/fwd  %7.1f us avg  %6.1f us/fwd  %s" % (int(r['Calls'])/30, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/30/1e3, r['Name'][:90]))
PY
