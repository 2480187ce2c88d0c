#!/usr/bin/env python3
"""Why does BASELINE configs[4] (64 x 8000x6000, one launch, 24.6 GB) run at 0.76 of 8 TB/s when configs[1] runs at 0.81?
Same kernel path (COPY tiles).  Varies the image count (working set: 1.5 .. 24.6 GB), the width (8000 vs 8192: partial last
tile, row pitch) and prints the runtime's own copy of the same bytes beside each.  usage: python tools/exp_config5.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

dev = torch.device("cuda", 0)
st = ist.Stitcher(0)


def run(n, w, h, reps=12):
    imgs = [{"width": w, "height": h, "opaque": True}] * n
    p, job = st.compile(imgs, "vertical", {"filter": "bilinear"})
    srcs = [torch.empty((h + 1, w, 4), dtype=torch.uint8, device=dev)[:h].random_(0, 256) for _ in range(n)]
    out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev)
    for _ in range(max(3, int(0.06 / (n * w * h * 8 / 6e12)))):
        job.launch(srcs, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        job.launch(srcs, out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    B = job.info["algorithmic_bytes"]
    flat = torch.empty(B // 2, dtype=torch.uint8, device=dev)
    flat2 = torch.empty(B // 2, dtype=torch.uint8, device=dev)
    for _ in range(3):
        flat2.copy_(flat)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        flat2.copy_(flat)
    e1.record()
    torch.cuda.synchronize()
    cus = e0.elapsed_time(e1) * 1e3 / reps
    print("%3d x %dx%d: %8.1f us  %.3f of 8 TB/s   (%.1f GB moved; tiles %d)   torch copy_ of the same bytes: %8.1f us %.3f" %
          (n, w, h, us, B / (us * 1e-6) / 8e12, B / 1e9, job.info["n_tiles"], cus, B / (cus * 1e-6) / 8e12), flush=True)
    del srcs, out, flat, flat2, job
    torch.cuda.empty_cache()


for n, w, h in ((9, 4032, 3024), (36, 4032, 3024), (128, 4032, 3024), (4, 8000, 6000), (16, 8000, 6000), (64, 8000, 6000), (64, 8192, 6000), (64, 7936, 6000)):
    run(n, w, h)
