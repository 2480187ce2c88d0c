#!/usr/bin/env python3
"""Host side of the last ist_stitch_files_png call of tools/exp_pipeline.py from a rocprofv3 --hip-trace run: the HIP API calls
that took longer than 30 us, in time order (relative to the first device activity of the call), per thread."""
import csv
import glob
import json
import os
import sys

o = sys.argv[1]
run = json.loads(open(os.path.join(o, "run.json")).read().strip().splitlines()[-1])
f = glob.glob(o + "/trace/**/*_hip_api_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
t0, t1 = run["last_call_window_ns"]
# the profiler's timestamps and CLOCK_MONOTONIC share a base on Linux; fall back to "the last 9 ms" if they do not overlap
rows = [r for r in rows]
ts = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "")) for r in rows]
inwin = [x for x in ts if t0 <= x[0] <= t1]
if not inwin:
    last = max(x[1] for x in ts)
    inwin = [x for x in ts if x[0] >= last - 9_000_000]
    t0 = min(x[0] for x in inwin)
print("HIP API calls of the last call: %d; longer than 30 us:" % len(inwin))
for s, e, fn, th in sorted(inwin):
    if e - s > 30000:
        print("  %8.1f us  +%7.1f us  %-34s thread %s" % ((s - t0) / 1e3, (e - s) / 1e3, fn, th))
agg = {}
for s, e, fn, th in inwin:
    a = agg.setdefault(fn, [0, 0]); a[0] += 1; a[1] += e - s
print("totals:")
for fn, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-34s x%-4d %9.1f us" % (fn, a[0], a[1] / 1e3))
