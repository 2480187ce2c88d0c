#!/usr/bin/env python3
"""The host entry point (ist_stitch_rgba8 through imagestitching_amd.stitch: pageable numpy in, pooled pinned block out) on BASELINE configs[1]
and [2]'s geometry, median of 7 calls.  Run once as is (row bands, uploads and downloads overlapped) and once with IST_TUNING=1
IST_HOST_DUPLEX=0 (upload everything, one launch, download everything) on the same box.  usage: python tools/exp_host_path.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import imagestitching_amd as ist  # noqa: E402
from imagestitching_amd import _lib as L  # noqa: E402

px = [bench.synth_np(k, 4032, 3024) for k in range(9)]
imgs = [{"width": 4032, "height": 3024, "data": a, "opaque": True} for a in px]
mixed = [bench.synth_np(20 + k, w, h) for k, (w, h) in enumerate(bench.MIXED)]
mimgs = [{"width": a.shape[1], "height": a.shape[0], "data": a, "opaque": True} for a in mixed]
for name, im, direction in (("9 x 4032x3024 vertical", imgs, "vertical"), ("9 x 4032x3024 horizontal", imgs, "horizontal"),
                            ("mixed sizes vertical (resampled)", mimgs, "vertical"), ("mixed sizes horizontal (resampled)", mimgs, "horizontal")):
    ist.stitch(im, direction, {"filter": "bilinear"})
    ts = []
    before = L.lib.ist_debug_duplex_stitches()
    for _ in range(7):
        t0 = time.perf_counter()
        r = ist.stitch(im, direction, {"filter": "bilinear"})
        ts.append(time.perf_counter() - t0)
        nbytes = r["data"].nbytes
        del r
    t = sorted(ts)[3]
    print("%-36s %6.2f ms per stitch  (%.1f GB/s of payload, in + out)  %s" % (name, t * 1e3, (sum(a.nbytes for a in (px if im is imgs else mixed)) + nbytes) / t / 1e9,
          "bands, both directions busy" if L.lib.ist_debug_duplex_stitches() > before else "upload all, launch, download all"), flush=True)
