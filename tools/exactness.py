#!/usr/bin/env python3
"""How far the HIP bilinear path is from the CPU oracle, in LSBs (the contract allows 1; the arithmetic is meant to be identical)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import imagestitching_amd as ist
from oracle import oracle as O
sizes = [(640, 480), (480, 640), (600, 450), (333, 517), (801, 200)]
for opaque in (True, False):
    px = [O.synth_image(k, h, w, opaque=opaque) for k, (w, h) in enumerate(sizes)]
    for orient in (None, [1, 6, 3, 8, 5], [2, 4, 7, 1, 6]):
        for direction in ("vertical", "horizontal"):
            for mode in ("min", "max"):
                imgs = [{"width": a.shape[1], "height": a.shape[0], "data": a, "orientation": (orient[i] if orient else 1), "opaque": opaque} for i, a in enumerate(px)]
                # natural sizes follow the orientation (quarter turns swap them), as the reference's getImageInfo reports
                for d in imgs:
                    if d["orientation"] >= 5:
                        d["width"], d["height"] = d["height"], d["width"]
                        d["bmp_w"], d["bmp_h"] = d["data"].shape[1], d["data"].shape[0]
                try:
                    got = ist.stitch(imgs, direction, {"filter": "bilinear", "mode": mode, "gap": 3})
                except Exception as e:
                    print("skip", opaque, orient, direction, mode, e); continue
                descs = [{"width": d["width"], "height": d["height"], "orientation": d["orientation"], "bmp_w": d.get("bmp_w", 0), "bmp_h": d.get("bmp_h", 0)} for d in imgs]
                rc, pd, rl = O.plan(descs, direction, mode, 3, O.lifted_limits(1.0))
                ref = O.render(pd, rl, descs, px, "bilinear")
                diff = np.abs(got["data"].astype(np.int16) - ref.astype(np.int16))
                print("opaque=%s orient=%s %s %s: max diff %d, differing bytes %d of %d" % (opaque, orient, direction, mode, diff.max(), int((diff > 0).sum()), diff.size), flush=True)
