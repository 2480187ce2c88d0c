#!/usr/bin/env python3
"""Spread of the file pipeline's wall time over many calls (nine photo-like 12 MP JPEGs -> one PNG), and of the decode-only call:
median, p10 / p90, and every call that took much longer than the rest (the tail a mean hides).  usage: python tools/exp_pipeline_dist.py [calls]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import imagestitching_amd as ist  # noqa: E402


def spread(name, ts, slow):
    s = sorted(ts)
    print("%-22s %d calls: median %.2f  p10 %.2f  p90 %.2f  max %.2f ms; over %.0f ms: %s" % (name, len(ts), s[len(s) // 2], s[len(s) // 10], s[9 * len(s) // 10], s[-1], slow, [t for t in ts if t > slow]))


calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
blobs = [bench.photo_jpeg(k, w, h) for k, (w, h) in enumerate(bench.UNIFORM)]
tmp = tempfile.mkdtemp()
paths = []
for k, b in enumerate(blobs):
    p = os.path.join(tmp, "in%d.jpg" % k)
    with open(p, "wb") as f:
        f.write(b)
    paths.append(p)
ts = []
for i in range(calls + 3):
    t0 = time.perf_counter()
    r = ist.stitch_files(paths, "vertical", copy=False)
    ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    del r
spread("files -> PNG", ts[3:], 8.0)
out, _ = ist.decode_files_device(blobs)
ts = []
for i in range(calls + 3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ist.decode_files_device(blobs, out=out)
    torch.cuda.synchronize()
    ts.append(round((time.perf_counter() - t0) * 1e3, 2))
spread("files -> bitmaps (HBM)", ts[3:], 4.0)
