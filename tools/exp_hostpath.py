#!/usr/bin/env python3
"""The bench's PCIe-inclusive host path leg alone (numpy in -> ist_stitch_rgba8 -> pinned block out).  IST_TIMING=1 prints
the band pipeline's host-side timeline.  usage: python tools/exp_hostpath.py [vertical|horizontal] [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import imagestitching_amd as ist  # noqa: E402

direction = sys.argv[1] if len(sys.argv) > 1 else "vertical"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
px = [bench.synth_np(k, 4032, 3024) for k in range(9)]
imgs = [{"width": 4032, "height": 3024, "data": a, "opaque": True} for a in px]
ist.stitch(imgs, direction, {"filter": "bilinear"})
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    r = ist.stitch(imgs, direction, {"filter": "bilinear"})
    ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    del r
print(json.dumps({"direction": direction, "ms": ts}), flush=True)
