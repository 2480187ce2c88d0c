#!/usr/bin/env python3
"""EXIF orientations on a mixed-size strip (every image resampled AND turned/mirrored): the SWAP_LDS / flipped SAMPLE paths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagestitching_amd as ist
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
NAT = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
for o in (1, 2, 3, 6, 5):
    imgs = [{"width": w, "height": h, "orientation": o, "opaque": True, "bmpWidth": (h if o >= 5 else w), "bmpHeight": (w if o >= 5 else h)} for (w, h) in NAT]
    p, job = st.compile(imgs, "vertical", {"filter": "bilinear"})
    srcs = [torch.randint(0, 256, ((w if o >= 5 else h), (h if o >= 5 else w), 4), dtype=torch.uint8, device=dev) for (w, h) in NAT]
    out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev)
    ts = []
    for r in range(4):
        for _ in range(3):
            job.launch(srcs, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            job.launch(srcs, out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    ts.sort()
    B = job.info["algorithmic_bytes"]; t = job.info
    print("orientation %d mixed: canvas %dx%d  %.1f us  %.0f GB/s (%.1f%%)  tiles copy/sample/general = %d/%d/%d" % (o, p.canvas_w, p.canvas_h, ts[1], B / ts[1] / 1e3, B / ts[1] / 1e3 / 80, t["tiles_copy"], t["tiles_sample"], t["tiles_general"]), flush=True)
