#!/usr/bin/env python3
"""One bench configuration (default: the mixed-size horizontal strip) launched back to back: PREROLL untimed launches so that
the chip holds its working clock, then N launches.  Driven by tools/profile_mixed_horizontal.sh under rocprofv3 (kernel trace
in one run, each --pmc counter in its own run).  Prints one JSON line: the job's geometry, tiles per band, bytes."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

MIXED = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
UNIFORM = [(4032, 3024)] * 9
CONFIG5 = [(8000, 6000)] * 64            # BASELINE configs[4]: 24.6 GB moved per launch; run it with --sets 1 --preroll 20 --launches 10
CONFIGS = {"mixed_horizontal": (MIXED, "horizontal"), "mixed_vertical": (MIXED, "vertical"),
           "uniform_horizontal": (UNIFORM, "horizontal"), "uniform_vertical": (UNIFORM, "vertical"), "config5": (CONFIG5, "vertical"),
           "exif6_scaled": (MIXED, "vertical"), "exif6_unit": (UNIFORM, "vertical")}     # every image quarter-turned (EXIF 6)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="mixed_horizontal")
    ap.add_argument("--launches", type=int, default=50)
    ap.add_argument("--preroll", type=int, default=400)
    ap.add_argument("--sets", type=int, default=3)
    args = ap.parse_args()
    sizes, direction = CONFIGS[args.which]
    dev = torch.device("cuda", 0)
    st = ist.Stitcher(0)
    turned = args.which.startswith("exif6")
    if turned:                              # natural size w x h, stored bitmap h x w (what a phone writes for a portrait shot)
        imgs = [{"width": w, "height": h, "orientation": 6, "opaque": True, "bmpWidth": h, "bmpHeight": w} for (w, h) in sizes]
        sizes = [(h, w) for (w, h) in sizes]
    else:
        imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in sizes]
    p, job = st.compile(imgs, direction, {"filter": "bilinear"})
    sets = [[torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev) for (w, h) in sizes] for _ in range(args.sets)]
    outs = [torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev) for _ in range(args.sets)]
    torch.cuda.synchronize()
    for i in range(args.preroll):
        job.launch(sets[i % args.sets], outs[i % args.sets])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(args.launches):
        job.launch(sets[i % args.sets], outs[i % args.sets])
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"workload": args.which, "preroll": args.preroll, "launches": args.launches, "canvas": [p.canvas_w, p.canvas_h],
                      "event_us_per_launch": round(e0.elapsed_time(e1) * 1e3 / args.launches, 2),
                      "algorithmic_bytes": job.info["algorithmic_bytes"], "n_tiles": job.info["n_tiles"], "n_cells": job.info["n_cells"],
                      "tiles": {k: job.info[k] for k in ("tiles_fill", "tiles_copy", "tiles_sample", "tiles_general")}}), flush=True)


if __name__ == "__main__":
    main()
