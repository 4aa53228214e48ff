#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_WAVES_EQ_64"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmcf_$i -o p -- python3 $R/tools/tp_bench.py > $R/gpurun_out/pmcf_$i.log 2>&1
done
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for i in (1,2,3):
    try: rows=list(csv.DictReader(open("$R/gpurun_out/pmcf_%d/p_counter_collection.csv"%i)))
    except Exception as e: print("pass",i,"failed",e); continue
    for r in rows:
        if "tp_fused" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    # launches: (2 warm + 5 timed) x [all, l1=0, ..., l1=4] => indices 3, 10, 17, 24, 31, 38
    print("%-26s all=%.4g  l1=0..4: %s  (n=%d)" % (k, v[3], " ".join("%.3g" % v[i] for i in (10, 17, 24, 31, 38) if i < len(v)), len(v)))
PY
