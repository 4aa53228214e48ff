#!/usr/bin/env python3
"""Timeline of the last forward in a rocprofv3 kernel trace (see step_trace.sh)."""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"]
# a forward starts at the species embedding kernel
starts = [i for i, r in enumerate(rows) if "species_embed_kernel" in name(r)]
i0 = starts[-1]
i1 = len(rows)
for i in range(i0 + 1, len(rows)):   # the step ends with the last kernel before the next embedding / end of trace
    if "species_embed_kernel" in name(rows[i]):
        i1 = i
        break
step = rows[i0:i1]
# what follows the forward in the process (the bench's finiteness check, the final read-back) is not part of it: cut at the
# first host-paced hole behind the conv stack
for k in range(10, len(step)):
    if int(step[k]["Start_Timestamp"]) - int(step[k - 1]["End_Timestamp"]) > 100_000:
        step = step[:k]
        break
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
tot_busy = tot_gap = 0
ours = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    n = name(r)
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if "at::native" in n:
        import re
        m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", n[n.index("<"):] if "<" in n else n)
        short = "torch:" + (m.group(1) if m else short[-40:])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  {short[:90]}")
    tot_busy += e - s
    tot_gap += max(gap, 0)
    prev_end = max(prev_end, e)
print(f"kernels {len(step)}  busy {tot_busy / 1e3:.1f} us  gaps {tot_gap / 1e3:.1f} us  span {(prev_end - t0) / 1e3:.1f} us")
