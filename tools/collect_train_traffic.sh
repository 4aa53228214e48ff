#!/bin/bash
# HBM traffic of the batch-2048 training step's kernels (tools/train_b2048.py: 18 steps) from rocprofv3 PMC counters, FETCH_SIZE and
# WRITE_SIZE in separate passes, gfx950 corrections as in tools/collect_traffic.sh (FETCH x2 for wide streams; KB -> bytes x1024).
#   bash tools/collect_train_traffic.sh <tag> [batch]    -> gpurun_out/train_traffic_<tag>.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
BATCH=${2:-2048}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_train_${TAG}_$c -o p -- python3 $R/tools/train_b2048.py $BATCH > $R/gpurun_out/pmc_train_${TAG}_$c.log 2>&1
done
python3 - <<PY
import csv, json, collections, re, os
R="$R"; TAG="$TAG"
N_STEPS = 18
def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(_ZN[0-9A-Za-z_]*?N_1\d+)?([A-Za-z_][A-Za-z0-9_]*)(<[^(]*>)?", name)
    if name.startswith("_ZN"):
        m2 = re.search(r"\d+([a-z_0-9]+_kernel)", name)
        return m2.group(1) if m2 else name[:48]
    return (m.group(2) + (m.group(3) or "")) if m else name[:48]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"{R}/gpurun_out/pmc_train_{TAG}_{c}/p_counter_collection.csv")))
    per = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == c:
            per[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in per.items():
        res.setdefault(k, {})[c] = {"sum_kb": sum(v), "mean_kb": sum(v) / len(v), "launches": len(v)}
out = {}
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        out[k] = {"launches_per_step": d["FETCH_SIZE"]["launches"] / N_STEPS,
                  "fetch_bytes_mean_launch_x2": 2 * 1024 * d["FETCH_SIZE"]["mean_kb"], "write_bytes_mean_launch": 1024 * d["WRITE_SIZE"]["mean_kb"],
                  "hbm_bytes_mean_launch": 2 * 1024 * d["FETCH_SIZE"]["mean_kb"] + 1024 * d["WRITE_SIZE"]["mean_kb"],
                  "hbm_bytes_per_step": (2 * 1024 * d["FETCH_SIZE"]["sum_kb"] + 1024 * d["WRITE_SIZE"]["sum_kb"]) / N_STEPS}
doc = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_train_traffic.sh) of "
                 "tools/train_b2048.py (18 steps: lmax-2 model, $BATCH crystals); FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md section HBM), KB x1024",
       "tag": TAG, "batch": int("$BATCH"), "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"]))}
try:
    doc["commit"] = open(os.path.join(R, ".head_commit")).read().strip() if os.path.exists(os.path.join(R, ".head_commit")) else None
except Exception:
    pass
json.dump(doc, open(f"{R}/gpurun_out/train_traffic_{TAG}.json", "w"), indent=1)
tot = sum(v["hbm_bytes_per_step"] for v in out.values())
print(f"HBM bytes per step (all kernels): {tot/1e9:.3f} GB")
for k, v in list(doc["kernels"].items())[:14]:
    print(f"{k[:56]:56s} launches/step {v['launches_per_step']:5.1f}  mean launch: fetch x2 {v['fetch_bytes_mean_launch_x2']/1e6:8.1f} MB  write {v['write_bytes_mean_launch']/1e6:8.1f} MB   per step {v['hbm_bytes_per_step']/1e6:8.1f} MB")
PY
