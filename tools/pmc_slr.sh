#!/bin/bash
# SQ counters of species_linear_rows_kernel over one bench forward (largest launch = lin1 + self-connection of the last layer)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA"
P3="GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmcr_$i -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcr_$i.log 2>&1
done
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for i in (1,2,3):
    rows=list(csv.DictReader(open("$R/gpurun_out/pmcr_%d/p_counter_collection.csv"%i)))
    for r in rows:
        if "species_linear_rows" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    n=len(v)//3  # launches per forward
    last=v[-n:]
    print("%-26s per forward launch: %s" % (k, " ".join("%.3g"%x for x in last)))
PY
