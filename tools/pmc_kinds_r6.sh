#!/bin/bash
# SQ / TA / TCP / TCC counters of tp_fused per group kind on the PRODUCTION library (no rebuild): tools/fused_kind_bench.py launches
# the last conv layer's entries kind by kind (7 launches each: "all", then every kind); five --pmc passes.
#   bash tools/pmc_kinds_r6.sh <tag> [TARGET=view]     -> gpurun_out/<tag>_tp_fused_pmc_by_kind.txt
R=$GRAFT_REPO_ROOT; TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
P4="TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
P5="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmck_${TAG}_$i -o p -- python3 $R/tools/fused_kind_bench.py > $R/gpurun_out/pmck_${TAG}_$i.log 2>&1
done
python3 - > $R/gpurun_out/${TAG}_tp_fused_pmc_by_kind.txt <<PY
import csv, collections, re
labels = [l.split()[1:3] for l in open("$R/gpurun_out/pmck_${TAG}_1.log") if l.startswith("fused ")]
labels = [" ".join(x) if x[0] != "all" else "all" for x in labels]
times = [float(l.split()[-2]) for l in open("$R/gpurun_out/pmck_${TAG}_1.log") if l.startswith("fused ")]
agg = collections.defaultdict(list)
for i in (1, 2, 3, 4, 5):
    try:
        rows = list(csv.DictReader(open("$R/gpurun_out/pmck_${TAG}_%d/p_counter_collection.csv" % i)))
    except Exception as e:
        print("pass", i, "failed", e); continue
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0) or 0))
    for r in rows:
        if "tp_fused" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("tp_fused per group kind, last conv layer (TARGET=${TARGET:-full}), 1000 fcc-64 crystals, production library; launch 4 of 7 per kind (rocprofv3 --pmc, five passes)")
print("%-34s " % "counter" + " ".join("%11s" % l.replace("l1=", "(").replace(" g=", ",") .replace("all", "all") for l in labels))
print("%-34s " % "ms under the profiler (pass 1)" + " ".join("%11.3f" % t for t in times))
for k, v in agg.items():
    print("%-34s " % k + " ".join("%11.4g" % v[7 * j + 3] for j in range(len(labels)) if 7 * j + 3 < len(v)) + "  (n=%d)" % len(v))
PY
rm -rf $R/gpurun_out/pmck_${TAG}_[1-5]
cat $R/gpurun_out/${TAG}_tp_fused_pmc_by_kind.txt | cut -c1-200
