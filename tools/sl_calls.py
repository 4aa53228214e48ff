"""Median duration of each species_linear launch of one forward, from a rocprofv3 kernel trace."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
sl = [r for r in rows if "species_linear" in r["Kernel_Name"]]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
per = len(sl) // steps
d = collections.defaultdict(list)
for i, r in enumerate(sl):
    d[i % per].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
out = []
for k in sorted(d):
    v = sorted(d[k]); m = v[len(v) // 2]; tot += m
    out.append(f"{m:.0f}")
print("species_linear per call (us):", " ".join(out), "| sum", f"{tot:.0f}")
