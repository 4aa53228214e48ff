#!/bin/bash
# weight block [u][live c] (TPF_COMPACT=1) against [u][c] with zero columns (0): same box, per-layer tp_fused times
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/matten_amd/csrc
run() {  # label, flags, env
    touch tp_fused.hip conv_tile.hip
    make EXTRA_CXXFLAGS="$2" > /dev/null 2>&1 || { echo "build failed: $1"; return; }
    echo "== $1"
    ( cd $R && env $3 python3 bench.py --steps 40 --warmup 10 --no-calibration 2>/dev/null | grep '^{"metric"' | python3 -c '
import json,sys
d=json.loads(sys.stdin.readline())
print("ms_per_step", round(d["ms_per_step"],4), " per layer", [round(p["ms"],4) for p in d["roofline"]["per_layer"]])' )
}
run "compact" "" "MATTEN_TP_COMPACT=1"
run "plain [u][c]" "-DTPF_COMPACT=0" "MATTEN_TP_COMPACT=0"
run "compact" "" "MATTEN_TP_COMPACT=1"
run "plain [u][c]" "-DTPF_COMPACT=0" "MATTEN_TP_COMPACT=0"
touch tp_fused.hip conv_tile.hip; make > /dev/null 2>&1
