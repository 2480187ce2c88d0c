#!/usr/bin/env python3
"""Tuning sweep for the COPY path tile shape (IST_COPY_TILE knob), interleaved rounds in ONE process
(cdna_hip_programming.md rule 24).  usage: python tools/sweep_copy.py [rounds]"""
import os
import sys

os.environ["IST_TUNING"] = "1"      # the launch-time knobs are only read in tuning mode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

TILES = ["256x32", "256x16", "256x64", "512x16", "512x8", "1024x8", "1024x4", "2048x4", "4096x2", "4096x1", "1024x16", "512x32"]
if len(sys.argv) > 2:
    TILES = sys.argv[2].split(",")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
imgs = [{"width": 4032, "height": 3024, "opaque": True} for _ in range(9)]
nsets = 3
sets = [[torch.randint(0, 256, (3024, 4032, 4), dtype=torch.uint8, device=dev) for _ in range(9)] for _ in range(nsets)]
res = {}
for direction in ("vertical", "horizontal"):
    jobs = {}
    for t in TILES:
        os.environ["IST_COPY_TILE"] = t
        p, job = st.compile(imgs, direction, {"filter": "bilinear"})
        jobs[t] = job
    outs = [torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev) for _ in range(nsets)]
    for r in range(rounds):
        for t in TILES:
            job = jobs[t]
            for i in range(3):
                job.launch(sets[i % nsets], outs[i % nsets])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 30
            e0.record()
            for i in range(n):
                job.launch(sets[i % nsets], outs[i % nsets])
            e1.record()
            torch.cuda.synchronize()
            res.setdefault((direction, t), []).append(e0.elapsed_time(e1) * 1e3 / n)
    for t in TILES:
        v = sorted(res[(direction, t)])
        print("%-10s tile %-8s tiles=%6d  median %.1f us  min %.1f us  -> %.0f GB/s" % (direction, t, jobs[t].info["n_tiles"], v[len(v) // 2], v[0], 877879296 / v[len(v) // 2] / 1e3), flush=True)
    del outs, jobs
