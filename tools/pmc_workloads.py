#!/usr/bin/env python3
"""Launches each supplementary workload N times, in a fixed order, for the rocprofv3 passes of tools/profile_plans.sh
(kernel trace, FETCH_SIZE, WRITE_SIZE: each in its own pass).  Prints one JSON line per workload with the algorithmic
bytes; profiles/summarize_plans.py joins that with the counters (dispatches are grouped N at a time, in this order)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

N = 10
MIXED = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
UNI = [(4032, 3024)] * 9


def img(w, h, o=1):
    return {"width": w, "height": h, "orientation": o, "opaque": True, "bmpWidth": (h if o >= 5 else w), "bmpHeight": (w if o >= 5 else h)}


WORKLOADS = [
    ("ios_plan_bilinear (9x12MP vertical, iOS caps -> 1820x12288)", [img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "platform": "ios", "superSample": 1}),
    ("ios_plan_nearest", [img(w, h) for w, h in UNI], "vertical", {"filter": "nearest", "platform": "ios", "superSample": 1}),
    ("android_plan_bilinear (606x4096)", [img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "platform": "android", "superSample": 1}),
    ("shrink_4x_bilinear (9x12MP vertical, maxSide 6804 -> 1008x6804)", [img(w, h) for w, h in UNI], "vertical", {"filter": "bilinear", "maxSide": 6804}),
    ("exif6_scaled (mixed sizes, every image quarter-turned and resampled)", [img(w, h, 6) for w, h in MIXED], "vertical", {"filter": "bilinear"}),
    ("exif5_scaled", [img(w, h, 5) for w, h in MIXED], "vertical", {"filter": "bilinear"}),
    ("exif8_scaled", [img(w, h, 8) for w, h in MIXED], "vertical", {"filter": "bilinear"}),
    ("exif3_scaled (half turn)", [img(w, h, 3) for w, h in MIXED], "vertical", {"filter": "bilinear"}),
    ("mixed_horizontal", [img(w, h) for w, h in MIXED], "horizontal", {"filter": "bilinear"}),
    ("ios_plan_area (box filter: every source byte is used)", [img(w, h) for w, h in UNI], "vertical", {"filter": "area", "platform": "ios", "superSample": 1, "edgeAA": False}),
    ("android_plan_area", [img(w, h) for w, h in UNI], "vertical", {"filter": "area", "platform": "android", "superSample": 1, "edgeAA": False}),
    ("shrink_4x_area", [img(w, h) for w, h in UNI], "vertical", {"filter": "area", "maxSide": 6804}),
]


def main():
    dev = torch.device("cuda", 0)
    st = ist.Stitcher(0)
    for name, imgs, direction, opts in WORKLOADS:
        p, job = st.compile(imgs, direction, opts)
        srcs = [torch.randint(0, 256, (i["bmpHeight"], i["bmpWidth"], 4), dtype=torch.uint8, device=dev) for i in imgs]
        out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for _ in range(N):
            job.launch(srcs, out)
        torch.cuda.synchronize()
        print(json.dumps({"workload": name, "launches": N, "canvas": [p.canvas_w, p.canvas_h], "algorithmic_bytes": job.info["algorithmic_bytes"],
                          "src_bytes_touched": 4 * job.info["src_pixels_touched"], "out_bytes": 4 * job.info["out_pixels"],
                          "tiles": [job.info[k] for k in ("tiles_fill", "tiles_copy", "tiles_sample", "tiles_general")]}), flush=True)
        del srcs, out, job
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
