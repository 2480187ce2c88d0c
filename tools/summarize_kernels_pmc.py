#!/usr/bin/env python3
"""Summary of tools/profile_kernels_pmc.sh: per kernel (name filter) and counter, the mean per dispatch over the dispatches
of the program's last quarter (the warm calls)."""
import csv
import glob
import os
import re
import sys


def main():
    o, flt = sys.argv[1], sys.argv[2]
    table, kernels = {}, []
    for p in sorted(glob.glob(os.path.join(o, "pmc_*"))):
        if not os.path.isdir(p):
            continue
        f = glob.glob(p + "/**/*_counter_collection.csv", recursive=True)
        if not f:
            continue
        rows = sorted((r for r in csv.DictReader(open(f[0])) if flt in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
        rows = rows[-max(1, len(rows) // 4):]
        for r in rows:
            m = re.search(r"(ist_\w+|__amd_\w+)", r["Kernel_Name"])
            k = (m.group(1) if m else r["Kernel_Name"])[:28]
            if k not in kernels:
                kernels.append(k)
            table.setdefault((r["Counter_Name"], k), []).append(float(r["Counter_Value"]))
    print("# per dispatch, mean over the warm calls (tools/profile_kernels_pmc.sh); kernels matching %r" % flt)
    print("%-26s" % "counter" + "".join("%30s" % k for k in kernels))
    for c in sorted({c for c, _ in table}):
        print("%-26s" % c + "".join("%30.1f" % (sum(table[(c, k)]) / len(table[(c, k)])) if (c, k) in table else "%30s" % "-" for k in kernels))
    f = os.path.join(o, "failed.txt")
    if os.path.exists(f):
        print(open(f).read())


if __name__ == "__main__":
    main()
