#!/usr/bin/env python3
"""Settled kernel time of the resampling workloads (200 untimed launches, then 100 timed) under whatever tuning knobs the
environment carries.  usage: IST_TUNING=1 IST_... python tools/exp_paths.py [label]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

MIXED = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
UNI = [(4032, 3024)] * 9


def img(w, h, o=1):
    return {"width": w, "height": h, "orientation": o, "opaque": True, "bmpWidth": (h if o >= 5 else w), "bmpHeight": (w if o >= 5 else h)}


WORKLOADS = [
    ("mixed_v", [img(w, h) for w, h in MIXED], "vertical", {"filter": "bilinear"}),
    ("mixed_h", [img(w, h) for w, h in MIXED], "horizontal", {"filter": "bilinear"}),
    ("mixed_v_max", [img(w, h) for w, h in MIXED], "vertical", {"filter": "bilinear", "mode": "max"}),
    ("ios_plan", [img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "platform": "ios", "superSample": 1}),
    ("android_plan", [img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "platform": "android", "superSample": 1}),
    ("k1.6", [img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "maxSide": 17010}),
    ("k3", [img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "maxSide": 9072}),
    ("k4", [img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "maxSide": 6804}),
    ("k5", [img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "maxSide": 5443}),
    ("k10", [img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "maxSide": 2722}),
    ("h_k1.4", [img(4032, 3024)] * 8 + [img(3840, 2160)], "horizontal", {"filter": "bilinear"}),
    ("v_k1.4", [img(4032, 3024)] * 8 + [img(2880, 2160)], "vertical", {"filter": "bilinear"}),
    ("h_k1.87", [img(3024, 4032)] * 8 + [img(3840, 2160)], "horizontal", {"filter": "bilinear"}),
    ("v_k1.87", [img(4032, 3024)] * 8 + [img(2160, 1620)], "vertical", {"filter": "bilinear"}),
    ("exif6", [img(w, h, 6) for w, h in MIXED], "vertical", {"filter": "bilinear"}),
    ("exif6_unit", [img(w, h, 6) for w, h in UNI], "vertical", {"filter": "bilinear"}),
    ("exif8_unit_h", [img(w, h, 8) for w, h in UNI], "horizontal", {"filter": "bilinear"}),
    ("exif3", [img(w, h, 3) for w, h in MIXED], "vertical", {"filter": "bilinear"}),
]
label = sys.argv[1] if len(sys.argv) > 1 else ""
only = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
res = {"label": label}
for name, imgs, direction, opts in WORKLOADS:
    if only and name not in only:
        continue
    p, job = st.compile(imgs, direction, opts)
    sets = [[torch.randint(0, 256, (i["bmpHeight"], i["bmpWidth"], 4), dtype=torch.uint8, device=dev) for i in imgs] for _ in range(2)]
    outs = [torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for i in range(200):
        job.launch(sets[i % 2], outs[i % 2])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(100):
        job.launch(sets[i % 2], outs[i % 2])
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 10.0
    res[name] = [round(us, 1), round(job.info["algorithmic_bytes"] / (us * 1e-6) / 8e12, 3)]
    del sets, outs, job
    torch.cuda.empty_cache()
print(json.dumps(res), flush=True)
