#!/usr/bin/env python3
"""Summarise one forward from a rocprofv3 kernel trace CSV: per-kernel-name count / total / avg, in launch order."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))  # the CSV is not in launch order
idx = [i for i, r in enumerate(rows) if "im2col" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
s = idx[which]
e = [i for i, r in enumerate(rows) if i > s and "rowdot" in r["Kernel_Name"]][0]
span = (int(rows[e]["End_Timestamp"]) - int(rows[s]["Start_Timestamp"])) / 1e3
agg = collections.OrderedDict()
busy = 0
for r in rows[s:e + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy += d
    nm = r["Kernel_Name"].split("(")[0].replace("void sm::", "").replace("sm::", "")
    key = (nm, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += d
print(f"forward span {span:.1f} us, busy {busy:.1f} us, {e - s + 1} kernels")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:9.1f} us  {n:3d} x {t / n:8.2f}  {k[0]}  grid=({k[1]},{k[2]},{k[3]})")
