#!/usr/bin/env python3
"""Names what the RARE slow call of the file pipeline waits for: runs the pipeline `calls` times in a process
started with IST_TUNING=1 IST_TIMELINE=1 IST_TIMELINE_SLOW_MS=<ms>: the library keeps host-side marks of every call (thread-local,
no printing) and dumps them to stderr only for a call that took longer than the threshold.
usage: IST_TUNING=1 IST_TIMELINE=1 IST_TIMELINE_SLOW_MS=7.5 python tools/exp_slow_call.py [calls]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import imagestitching_amd as ist  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 600
tmp = tempfile.mkdtemp()
paths = []
for k, (w, h) in enumerate(bench.UNIFORM):
    p = os.path.join(tmp, "in%d.jpg" % k)
    with open(p, "wb") as f:
        f.write(bench.photo_jpeg(k, w, h))
    paths.append(p)
ts = []
for i in range(calls + 3):
    t0 = time.perf_counter()
    r = ist.stitch_files(paths, "vertical", copy=False)
    dt = (time.perf_counter() - t0) * 1e3
    if i >= 3:
        ts.append(dt)
        if dt > float(os.environ.get("IST_TIMELINE_SLOW_MS", "7.5")):
            print("call %d took %.2f ms (its marks are above, on stderr)" % (i - 3, dt), file=sys.stderr, flush=True)
    del r
s = sorted(ts)
print("files -> PNG, %d calls: median %.2f  p10 %.2f  p90 %.2f  p99 %.2f  max %.2f ms; calls over 1.5 x median: %d" %
      (len(s), s[len(s) // 2], s[len(s) // 10], s[9 * len(s) // 10], s[99 * len(s) // 100], s[-1], sum(1 for t in s if t > 1.5 * s[len(s) // 2])), flush=True)
