#!/usr/bin/env python3
"""Kernel time of plans other than the BASELINE ones, 9 x 12 MP inputs resident in HBM (supplementary table in DESIGN.md)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagestitching_amd as ist
MIXED = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
UNI = [(4032, 3024)] * 9
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
cases = [
    ("mixed, mode max (upscale to 4032 wide)", MIXED, "vertical", {"mode": "max"}),
    ("mixed, mode original (copies + white margins)", MIXED, "vertical", {"mode": "original", "gap": 16}),
    ("uniform, gap 24 (copies + white gaps)", UNI, "vertical", {"gap": 24}),
    ("uniform, iOS caps (shrink 6.6x), bilinear", UNI, "vertical", {"platform": "ios", "superSample": 1}),
    ("uniform, iOS caps (shrink 6.6x), nearest", UNI, "vertical", {"platform": "ios", "superSample": 1, "filter": "nearest"}),
    ("uniform, android caps, reference superSample rule", UNI, "vertical", {"platform": "android"}),
    ("3 x 640x480, iOS, reference superSample 2.2 (configs[0] as the phone plans it)", [(640, 480)] * 3, "vertical", {"platform": "ios"}),
    ("uniform, nearest (identity)", UNI, "horizontal", {"filter": "nearest"}),
]
for name, sizes, direction, opts in cases:
    o = dict({"filter": "bilinear"}, **opts)
    imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in sizes]
    p, job = st.compile(imgs, direction, o)
    sets = [[torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev) for (w, h) in sizes] for _ in range(2)]
    outs = [torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
    ts = []
    for r in range(5):
        for i in range(3):
            job.launch(sets[i % 2], outs[i % 2])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            job.launch(sets[i % 2], outs[i % 2])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    ts.sort()
    B = job.info["algorithmic_bytes"]
    t = {k: job.info[k] for k in ("tiles_fill", "tiles_copy", "tiles_sample", "tiles_general")}
    print("%-80s canvas %6dx%-6d ss %.2f  %7.1f us  %5.0f GB/s (%4.1f%%)  tiles %s" % (name, p.canvas_w, p.canvas_h, p.super_sample, ts[2], B / ts[2] / 1e3, B / ts[2] / 1e3 / 80, t), flush=True)
    del sets, outs, job
    torch.cuda.empty_cache()
