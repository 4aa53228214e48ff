#!/bin/bash
# HBM traffic of the dominant kernel from rocprofv3 PMC counters (separate passes for FETCH_SIZE and
# WRITE_SIZE, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Run on the GPU box:
#   bash tools/collect_traffic.sh <tag>     -> gpurun_out/traffic_<tag>.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${TAG}_$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-calibration --no-full-layers > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1
done
python3 - <<PY
import csv, json, collections, re, os
R="$R"; TAG="$TAG"
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"{R}/gpurun_out/pmc_{TAG}_{c}/p_counter_collection.csv")))
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0) or 0))
    per = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == c:
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            per[m.group(1) if m else r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        res.setdefault(k, {})[c] = {"max_kb": max(v), "mean_kb": sum(v) / len(v), "launches": len(v),
                                    # the kernels launched once per conv layer: the mean per position in the forward (dispatch order)
                                    "by_layer_kb": [sum(v[i::4]) / len(v[i::4]) for i in range(4)] if k in ("tp_fused_kernel", "agg_linear_kernel") and len(v) % 4 == 0 else None}
out = {}
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> x2 (guide, section HBM)
        out[k] = {"fetch_kb_raw": d["FETCH_SIZE"]["max_kb"], "write_kb": d["WRITE_SIZE"]["max_kb"],
                  "hbm_bytes_per_launch": 2 * 1024 * d["FETCH_SIZE"]["max_kb"] + 1024 * d["WRITE_SIZE"]["max_kb"],
                  "hbm_bytes_mean_launch": 2 * 1024 * d["FETCH_SIZE"]["mean_kb"] + 1024 * d["WRITE_SIZE"]["mean_kb"],
                  "launches": d["FETCH_SIZE"]["launches"]}
        if d["FETCH_SIZE"].get("by_layer_kb") and d["WRITE_SIZE"].get("by_layer_kb"):
            out[k]["hbm_bytes_by_layer"] = [2 * 1024 * f + 1024 * w for f, w in zip(d["FETCH_SIZE"]["by_layer_kb"], d["WRITE_SIZE"]["by_layer_kb"])]
            out[k]["write_bytes_by_layer"] = [1024 * w for w in d["WRITE_SIZE"]["by_layer_kb"]]
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_traffic.sh via "
                 "tools/round_profile.sh), bench.py --steps 2; gfx950 correction: FETCH_SIZE x2 (MI355X_MICROARCH.md "
                 "section HBM); KB -> bytes x1024; 'per_launch' = largest launch (last conv layer), 'mean_launch' = "
                 "average over all launches of the kernel in a forward (bench.py --no-full-layers: the launches of the headline loop only; until round 5 the "
                 "second loop with the full last layer was averaged in); by_layer = per position in the forward",
       "tag": TAG, "kernels": out}
try:
    import subprocess
    # the GPU box has no .git: tools/gpu.sh writes the commit (+ "-dirty") into .head_commit before every gpurun call
    doc["commit"] = (subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
                     or (open(os.path.join(R, ".head_commit")).read().strip() if os.path.exists(os.path.join(R, ".head_commit")) else None))
except Exception:
    pass
json.dump(doc, open(f"{R}/gpurun_out/traffic_{TAG}.json", "w"), indent=1)
for k, v in out.items():
    if any(s in k for s in ("tp_", "radial", "species_linear", "agg_linear")):
        print(k, v)
PY
