#!/bin/bash
# per-call durations of the species-linear launches of one forward (rocprofv3 kernel trace of the default bench)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_slc
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_slc -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/prof_slc/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
sl=[r for r in rows if "species_linear" in r["Kernel_Name"]]
per=len(sl)//23
d=collections.defaultdict(list); names={}
for i,r in enumerate(sl):
    d[i%per].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3); names[i%per]="rows" if "rows" in r["Kernel_Name"] else "stream"
tot=0; out=[]
for k in sorted(d):
    v=sorted(d[k]); m=v[len(v)//2]; tot+=m; out.append("%s %.0f" % (names[k], m))
print(" | ".join(out), "| sum %.0f us" % tot)
PY
