#!/bin/bash
# SQ instruction / wait counters of every kernel of the bench forward (rocprofv3 PMC passes of `bench.py --steps 2`,
# 8 SQ counters per pass, never combined with trace domains other than --kernel-trace).  Run on the GPU box:
#   bash tools/collect_valu.sh <tag>   -> gpurun_out/valu_<tag>.json  (copy to profiles/tp_fused_valu.json: bench.py's
#                                          roofline.valu reads the committed file)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES"
P2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/valu_${TAG}_$i -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-calibration --no-full-layers > $R/gpurun_out/valu_${TAG}_$i.log 2>&1
done
python3 - <<PY
import csv, json, collections, re, subprocess, os
R="$R"; TAG="$TAG"
per = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    try:
        rows = list(csv.DictReader(open(f"{R}/gpurun_out/valu_{TAG}_{i}/p_counter_collection.csv")))
    except Exception as e:
        print("pass", i, "failed:", e); continue
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0) or 0))
    for r in rows:
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        per[m.group(1) if m else r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in per.items():
    out[k] = {}
    for c, v in d.items():
        out[k][c + "_mean_launch"] = sum(v) / len(v)
        out[k][c + "_max_launch"] = max(v)
        if k in ("tp_fused_kernel", "agg_linear_kernel") and len(v) % 4 == 0:
            out[k][c + "_by_layer"] = [sum(v[i::4]) / len(v[i::4]) for i in range(4)]   # per position in the forward (dispatch order)
    out[k]["launches"] = max(len(v) for v in d.values())
doc = {"source": "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes, tools/collect_valu.sh), bench.py --steps 2 "
                 "--warmup 1 --no-extras --no-full-layers (the launches of the headline loop only; until round 5 the second loop with the full last "
                 "layer was averaged in); *_mean_launch = average over all launches of the kernel (tp_fused: one per conv "
                 "layer and forward), *_max_launch = the largest; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count "
                 "quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES cycles (MI355X_MICROARCH.md)",
       "tag": TAG, "kernels": out}
try:
    # the GPU box has no .git: tools/gpu.sh writes the commit (+ "-dirty") into .head_commit before every gpurun call
    doc["commit"] = (subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
                     or (open(os.path.join(R, ".head_commit")).read().strip() if os.path.exists(os.path.join(R, ".head_commit")) else None))
except Exception:
    pass
json.dump(doc, open(f"{R}/gpurun_out/valu_{TAG}.json", "w"), indent=1)
for k in ("tp_fused_kernel", "agg_linear_kernel", "conv_fused_kernel"):
    if k in out:
        print(k, {c: "%.4g" % v for c, v in out[k].items() if c.endswith("_mean_launch")})
PY
