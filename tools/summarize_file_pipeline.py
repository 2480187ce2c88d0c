#!/usr/bin/env python3
"""Timeline of the LAST ist_stitch_files_png call of tools/exp_pipeline.py from a rocprofv3 kernel + memory-copy trace: per
kernel / copy kind the count, the busy time and the first start / last end relative to the first device activity of the call,
then the activities in time order (merged per kind into runs), so that what overlaps what can be read off."""
import csv
import glob
import json
import os
import sys


def rows(d, pat):
    f = glob.glob(d + "/**/" + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    for k in ("ist_jpeg_sync_kernel", "ist_jpeg_write_kernel", "ist_jpeg_idct_kernel", "ist_jpeg_color_kernel", "ist_stitch_kernel",
              "ist_stitch_area_kernel", "ist_png_deflate_kernel", "ist_png_gather_kernel", "ist_jpeg_scatter_kernel"):
        if k in name:
            return k
    return name.replace("void ", "").replace("ist::", "")[-40:]


def main():
    o = sys.argv[1]
    run = json.loads(open(os.path.join(o, "run.json")).read().strip().splitlines()[-1])
    print("# file pipeline (nine photo-like 12 MP JPEGs -> 4032x27216 PNG): rocprofv3 --kernel-trace --memory-copy-trace of tools/exp_pipeline.py")
    print("wall ms per call (profiled):", run["ms_per_call"])
    ev = []
    for r in rows(os.path.join(o, "trace"), "*_kernel_trace.csv"):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), 0))
    for r in rows(os.path.join(o, "trace"), "*_memory_copy_trace.csv"):
        kind = r.get("Direction") or r.get("Kind") or "copy"
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + kind.replace("MEMORY_COPY_", ""), int(float(r.get("Bytes") or r.get("Size") or 0))))
    if not ev:
        print("no trace rows found")
        return
    ev.sort()
    # the last call = the activities after the largest idle gap... simpler: split the run into calls by gaps > 0.5 ms of device idleness
    calls, cur, last_end = [], [], None
    for e in ev:
        if last_end is not None and e[0] - last_end > 400000 and cur:
            calls.append(cur)
            cur = []
        cur.append(e)
        last_end = e[1] if last_end is None else max(last_end, e[1])
    if cur:
        calls.append(cur)
    call = calls[-1]
    t0 = call[0][0]
    span = (max(e[1] for e in call) - t0) / 1e3
    print("last call: %d device activities, first to last %.0f us" % (len(call), span))
    print()
    print("%-28s %6s %10s %10s %10s %12s" % ("kind", "count", "busy us", "first us", "last us", "bytes"))
    agg = {}
    for s, e, k, b in call:
        a = agg.setdefault(k, [0, 0, s, e, 0])
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], s); a[3] = max(a[3], e); a[4] += b
    for k, a in sorted(agg.items(), key=lambda kv: kv[1][2]):
        print("%-28s %6d %10.1f %10.1f %10.1f %12d" % (k, a[0], a[1] / 1e3, (a[2] - t0) / 1e3, (a[3] - t0) / 1e3, a[4]))
    print()
    print("timeline (consecutive activities of one kind merged):")
    runs = []
    for s, e, k, b in call:
        if runs and runs[-1][2] == k and s - runs[-1][1] < 50000:
            runs[-1][1] = max(runs[-1][1], e); runs[-1][3] += 1
        else:
            runs.append([s, e, k, 1])
    for s, e, k, n in runs[:120]:
        print("  %8.1f .. %8.1f us  %-28s x%d" % ((s - t0) / 1e3, (e - t0) / 1e3, k, n))


if __name__ == "__main__":
    main()
