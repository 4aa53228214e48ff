#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of tools/ubench/store_shapes.hip's kernels against their known byte counts -> calibration table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-store_calib}
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/store_shapes.hip -o /tmp/store_shapes 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
/tmp/store_shapes > $O/bytes.txt
for c in WRITE_SIZE FETCH_SIZE "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_WRITE_REQ_sum TCC_REQ_sum"; do
  tag=$(echo $c | tr ' ' '+')
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -o p -- /tmp/store_shapes > $O/pmc_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections, re
O="$O"
print(open(O + "/bytes.txt").read())
known = {"st_piece": 65536 * 4320 * 4.0, "st_line": 65536 * 4352 * 4.0, "ld_stream": 65536 * 4352 * 4.0, "ld_rows": 65536 * 4320 * 4.0,
         "ld_gather4": 65536 * 64 * 64 * 4.0}
for d in sorted(glob.glob(O + "/pmc_*/")):
    f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    if not f:
        print(d, "no csv"); continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        per[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (name, c), v in sorted(per.items()):
        base = next((b for k, b in known.items() if name.startswith(k)), None)
        val = v[-1]
        extra = ""
        if base and c in ("WRITE_SIZE", "FETCH_SIZE"):
            extra = f"  = {val * 1024 / base:6.3f} x known bytes (counter in KB)"
        elif base and val:
            extra = f"  -> {base / val:8.1f} known bytes per count"
        print(f"{name:24s} {c:26s} {val:16.0f}{extra}")
PY
