#!/usr/bin/env python3
"""Sustained kernel time of the mixed-size strips: 300 back-to-back launches, the event time of launches 0-50 (burst),
100-200 and 200-300 (after the chip has settled its clock).  usage: python tools/exp_sustained.py [label]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

MIXED = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
label = sys.argv[1] if len(sys.argv) > 1 else ""
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
for direction in ("vertical", "horizontal"):
    imgs = [{"width": w, "height": h, "opaque": True} for w, h in MIXED]
    p, job = st.compile(imgs, direction, {"filter": "bilinear"})
    sets = [[torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev) for w, h in MIXED] for _ in range(3)]
    outs = [torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    marks = [0, 50, 100, 200, 300]
    ev = [torch.cuda.Event(enable_timing=True) for _ in marks]
    k = 0
    for i in range(300):
        if i == marks[k]:
            ev[k].record(); k += 1
        job.launch(sets[i % 3], outs[i % 3])
    ev[k].record()
    torch.cuda.synchronize()
    B = job.info["algorithmic_bytes"]
    seg = {"%d-%d" % (marks[j], marks[j + 1]): round(ev[j].elapsed_time(ev[j + 1]) * 1e3 / (marks[j + 1] - marks[j]), 1) for j in range(len(marks) - 1)}
    print(json.dumps({"label": label, "direction": direction, "us_per_launch": seg, "frac_settled": round(B / (seg["200-300"] * 1e-6) / 8e12, 3)}), flush=True)
    del sets, outs, job
    torch.cuda.empty_cache()
