#!/bin/bash
# SQ / TA / TCP counters of tp_fused per group kind (tools/fused_kind_bench.py: launches = 7 x [all, 9 kinds])
#   bash tools/pmc_kinds.sh <tag> [extra hipcc flags]
R=$GRAFT_REPO_ROOT; TAG=${1:-k}; FL="$2"
cd $R/matten_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. $FL -c tp_fused.hip -o build/tp_fused.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
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
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for i in (1,2,3,4,5):
    try: rows=list(csv.DictReader(open("$R/gpurun_out/pmck_${TAG}_%d/p_counter_collection.csv"%i)))
    except Exception as e: print("pass",i,"failed",e); continue
    for r in rows:
        if "tp_fused" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-34s %10s | %s" % ("counter", "all", "kinds (0,0) (1,0) (1,1) (2,0) (2,1) (3,0) (3,1) (4,0) (4,1)"))
for k,v in agg.items():
    print("%-34s %10.4g | %s  (n=%d)" % (k, v[3], " ".join("%9.3g" % v[7*j+3] for j in range(1,10) if 7*j+3 < len(v)), len(v)))
PY
cd $R/matten_amd/csrc && make -B -j8 > /dev/null 2>&1
