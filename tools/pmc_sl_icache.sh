#!/bin/bash
# instruction-cache / fetch counters of species_linear_kernel on the lin2 micro-benchmark
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|SQC_" | head -30 > $R/gpurun_out/sqc_counters.txt
P1="SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"
P2="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmci_$i -o p -- python3 $R/tools/sl_bench.py > $R/gpurun_out/pmci_$i.log 2>&1
done
python3 - <<PY
import csv, collections
agg=collections.defaultdict(list)
for i in (1,2):
    try: rows=list(csv.DictReader(open("$R/gpurun_out/pmci_%d/p_counter_collection.csv"%i)))
    except Exception as e: print("pass",i,"failed",e); continue
    for r in rows:
        if "species_linear" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print("%-28s %.4g  (n=%d)" % (k, v[min(3,len(v)-1)], len(v)))
PY
head -20 $R/gpurun_out/sqc_counters.txt
