#!/usr/bin/env python3
"""Durations (us) of the kernels whose name contains argv[2], in launch order, from a rocprofv3 --kernel-trace CSV directory;
a blank line where two launches are more than argv[3] (default 1000) us apart (= between calls)."""
import csv
import glob
import sys

d, flt = sys.argv[1], sys.argv[2]
gap = float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)
rows = sorted((r for r in csv.DictReader(open(f[0])) if flt in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
line, last = [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last is not None and (s - last) / 1e3 > gap:
        print(" ".join(line)); line = []
    line.append("%.0f" % ((e - s) / 1e3))
    last = e
print(" ".join(line))
