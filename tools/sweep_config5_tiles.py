#!/usr/bin/env python3
"""COPY tile shapes (IST_COPY_TILE, tuning mode) on BASELINE configs[4] (64 x 8000x6000) and configs[1] (9 x 4032x3024), interleaved rounds.
usage: python tools/sweep_config5_tiles.py [rounds]"""
import os
import sys

os.environ["IST_TUNING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
SHAPES = ["256x8", "256x4", "256x16", "512x4", "512x8", "1024x2", "1024x4", "2048x1", "2048x2"]
for name, n, w, h, reps in (("configs[4]", 64, 8000, 6000, 6), ("configs[1]", 9, 4032, 3024, 60), ("configs[2]", 9, 4032, 3024, 60)):
    direction = "horizontal" if name == "configs[2]" else "vertical"
    imgs = [{"width": w, "height": h, "opaque": True}] * n
    jobs = {}
    for shp in SHAPES:
        os.environ["IST_COPY_TILE"] = shp
        p, jobs[shp] = st.compile(imgs, direction, {"filter": "bilinear"})
    srcs = [torch.empty((h + 1, w, 4), dtype=torch.uint8, device=dev)[:h].random_(0, 256) for _ in range(n)]
    out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev)
    for _ in range(max(3, 400 // max(1, n * w * h // 12000000))):
        jobs["256x8"].launch(srcs, out)
    torch.cuda.synchronize()
    res = {s: [] for s in SHAPES}
    for r in range(rounds):
        for shp in SHAPES:
            job = jobs[shp]
            for _ in range(2):
                job.launch(srcs, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                job.launch(srcs, out)
            e1.record()
            torch.cuda.synchronize()
            res[shp].append(e0.elapsed_time(e1) * 1e3 / reps)
    B = jobs["256x8"].info["algorithmic_bytes"]
    for shp in SHAPES:
        t = sorted(res[shp])
        print("%-11s tile %-7s median %9.1f us  min %9.1f  -> %.3f of 8 TB/s  (%d tiles)" % (name, shp, t[len(t) // 2], t[0], B / (t[len(t) // 2] * 1e-6) / 8e12, jobs[shp].info["n_tiles"]), flush=True)
    del srcs, out, jobs
    torch.cuda.empty_cache()
