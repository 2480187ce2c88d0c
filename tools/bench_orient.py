#!/usr/bin/env python3
"""EXIF orientation cost: 9 x 12 MP vertical stitch with every image at orientation o (utils/canvas.js:160-200)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
for o, (w, h) in ((1, (4032, 3024)), (3, (4032, 3024)), (2, (4032, 3024)), (6, (3024, 4032)), (8, (3024, 4032)), (5, (3024, 4032))):
    # for 5..8 the bitmap is stored rotated: natural size = oriented size, bitmap = transposed
    imgs = [{"width": w, "height": h, "orientation": o, "opaque": True, "bmpWidth": (h if o >= 5 else w), "bmpHeight": (w if o >= 5 else h)} for _ in range(9)]
    p, job = st.compile(imgs, "vertical", {"filter": "bilinear"})
    bw, bh = (h, w) if o >= 5 else (w, h)
    srcs = [torch.randint(0, 256, (bh, bw, 4), dtype=torch.uint8, device=dev) for _ in range(9)]
    out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev)
    for _ in range(3):
        job.launch(srcs, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        job.launch(srcs, out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    B = job.info["algorithmic_bytes"]
    t = job.info
    print("orientation %d: canvas %dx%d  %.1f us  %.0f GB/s (%.1f%%)  tiles copy/sample/general = %d/%d/%d" % (o, p.canvas_w, p.canvas_h, us, B / us / 1e3, B / us / 1e3 / 80, t["tiles_copy"], t["tiles_sample"], t["tiles_general"]), flush=True)
