#!/bin/bash
# PMC counters of agg_linear_kernel (and the mul_ir kernel beside it) on tools/agg_bench.py, one pass per group
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum TCC_WRITE_sum"
P6="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmca_$i -o p -- python3 $R/tools/agg_bench.py > $R/gpurun_out/pmca_$i.log 2>&1
done
python3 - <<PY
import csv, collections
for kern in ("agg_linear_kernel", "species_linear_kernel"):
    agg=collections.defaultdict(list)
    for i in range(1,7):
        try: rows=list(csv.DictReader(open("$R/gpurun_out/pmca_%d/p_counter_collection.csv"%i)))
        except Exception as e: print("pass",i,"failed",e); continue
        for r in rows:
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", kern, "(largest launch of each counter)")
    for k,v in agg.items():
        print("%-32s %.5g  (n=%d)" % (k, max(v), len(v)))
PY
