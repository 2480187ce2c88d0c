#!/usr/bin/env python3
"""Summary of tools/profile_mixed_horizontal.sh: duration distribution of the timed launches + one line per counter."""
import csv
import glob
import json
import os
import sys


def rows(d, pat):
    f = glob.glob(d + "/**/" + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def main():
    o, which = sys.argv[1], sys.argv[2]
    print("# %s: rocprofv3 kernel trace + one --pmc counter per pass (tools/profile_mixed_horizontal.sh)" % which)
    for name in ("unprofiled", "trace"):
        try:
            print(name + ":", open(os.path.join(o, name + ".json")).read().strip())
        except OSError:
            pass
    kt = sorted((r for r in rows(os.path.join(o, "trace"), "*_kernel_trace.csv") if "ist_stitch" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in kt[-50:])
    if d:
        q = lambda f: d[min(len(d) - 1, int(f * len(d)))]      # noqa: E731
        print("kernel-trace durations of the last %d launches, us: min %.1f  median %.1f  p95 %.1f  max %.1f  mean %.1f" % (len(d), d[0], q(0.5), q(0.95), d[-1], sum(d) / len(d)))
        gaps = [(int(kt[i + 1]["Start_Timestamp"]) - int(kt[i]["End_Timestamp"])) / 1e3 for i in range(len(kt) - 50, len(kt) - 1)]
        print("gaps between consecutive launches, us: median %.1f max %.1f" % (sorted(gaps)[len(gaps) // 2], max(gaps)))
    print()
    print("%-32s %16s   (mean per dispatch over the last 10 dispatches)" % ("counter", "value"))
    for p in sorted(glob.glob(os.path.join(o, "pmc_*"))):
        if not os.path.isdir(p):
            continue
        cc = sorted((r for r in rows(p, "*_counter_collection.csv") if "ist_stitch" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
        by = {}
        for r in cc[-10 * max(1, len({x["Counter_Name"] for x in cc})):]:
            by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in by.items():
            print("%-32s %16.1f" % (k, sum(v) / len(v)))
    f = os.path.join(o, "failed.txt")
    if os.path.exists(f):
        print(open(f).read())


if __name__ == "__main__":
    main()
