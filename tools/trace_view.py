#!/usr/bin/env python3
"""Timeline of a rocprofv3 kernel trace (tools/gpu_trace.sh): per queue, the kernels of a few consecutive scans with start offset,
duration and the gap to the previous kernel on the same queue."""
import csv, glob, sys, collections
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace"
import os
f = max(glob.glob(d + "/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].split("(")[0].replace("scal::", "")
rows.sort(key=lambda r: r["s"])
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_map_begin"
nscan = int(sys.argv[3]) if len(sys.argv) > 3 else 2
gc = [r["s"] for r in rows if r["n"] == anchor]
per = [round((gc[i + 1] - gc[i]) / 1000) for i in range(len(gc) - 1)]
print("period us:", per[-40:])
t0, t1 = gc[-4 - nscan], gc[-4]
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items()):
    names = collections.Counter(r["n"] for r in rs)
    print("--- queue", q, "busy us in window:", round(sum(min(r["e"], t1) - max(r["s"], t0) for r in rs if r["e"] > t0 and r["s"] < t1) / 1000), "of", round((t1 - t0) / 1000))
    prev = None
    for r in rs:
        if t0 - 20000 <= r["s"] <= t1:
            gap = (r["s"] - prev) / 1000 if prev else 0
            print(f"  {(r['s']-t0)/1000:9.1f} +{(r['e']-r['s'])/1000:7.1f} gap {gap:6.1f}  {r['n']}  grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{r['Workgroup_Size_X']}")
        prev = r["e"]
