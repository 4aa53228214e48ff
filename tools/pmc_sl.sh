#!/bin/bash
# PMC counters of species_linear_kernel on the lin2 micro-benchmark (one pass per counter group)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
P4="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_TA_BUSY_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmcs_$i -o p -- python3 $R/tools/sl_bench.py > $R/gpurun_out/pmcs_$i.log 2>&1
done
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for i in (1,2,3,4):
    try: rows=list(csv.DictReader(open("$R/gpurun_out/pmcs_%d/p_counter_collection.csv"%i)))
    except Exception as e: print("pass",i,"failed",e); continue
    for r in rows:
        if "species_linear" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print("%-28s %.4g  (n=%d)" % (k, v[3], len(v)))
PY
