#!/usr/bin/env python3
"""The file pipeline a few times (nine photo-like 12 MP JPEGs -> one PNG), for tools/profile_file_pipeline.sh.  Prints the wall
time of every call and the wall-clock window of the LAST call (ns, same clock as the profiler's timestamps)."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import imagestitching_amd as ist  # noqa: E402

tmp = tempfile.mkdtemp()
paths = []
for k, (w, h) in enumerate(bench.UNIFORM):
    p = os.path.join(tmp, "in%d.jpg" % k)
    with open(p, "wb") as f:
        f.write(bench.photo_jpeg(k, w, h))
    paths.append(p)
times = []
win = None
for i in range(6):
    t0 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
    r = ist.stitch_files(paths, "vertical", copy=False)
    t1 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
    times.append(round((t1 - t0) / 1e6, 3))
    win = (t0, t1)
    del r
print(json.dumps({"ms_per_call": times, "last_call_window_ns": win}))
