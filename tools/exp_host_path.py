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

# the same nine images kept in page-locked memory by the host (torch pinned tensors): no staging, one copy per band and image
import torch  # noqa: E402

pinned = [torch.from_numpy(a).pin_memory() for a in px]
pimgs = [{"width": 4032, "height": 3024, "data": t.numpy(), "opaque": True} for t in pinned]
for direction in ("vertical", "horizontal"):
    ist.stitch(pimgs, direction, {"filter": "bilinear"})
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        r = ist.stitch(pimgs, direction, {"filter": "bilinear"})
        ts.append(time.perf_counter() - t0)
        del r
    print("9 x 4032x3024 %-10s from page-locked memory: %6.2f ms per stitch" % (direction, sorted(ts)[3] * 1e3), flush=True)

# photo-like images to a PNG file in host memory (ist_stitch_png, called through ctypes so that no Python copy of the file is timed):
# the next band's rows go up while the encoder compresses the last one and its slabs come down
import ctypes as C  # noqa: E402
import io  # noqa: E402

import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

import importlib  # noqa: E402

S = importlib.import_module("imagestitching_amd.stitch")

photos = [np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(bench.photo_jpeg(k, 4032, 3024))).convert("RGBA"))) for k in range(9)]
n = len(photos)
descs = S._descs([{"width": 4032, "height": 3024, "data": a, "opaque": True} for a in photos])
ptrs, pitches = (C.c_void_p * n)(), (C.c_size_t * n)()
for i, a in enumerate(photos):
    ptrs[i] = a.ctypes.data
    pitches[i] = a.strides[0]
o = S._merge({"filter": "bilinear"})
lim = S._limits(o)
ctx = S._ctx_png(0, 1)
for direction in ("vertical", "horizontal"):
    ts, ln = [], 0
    before = L.lib.ist_debug_duplex_stitches()
    for r in range(6):
        cplan, out, length = L.Plan(), C.POINTER(C.c_uint8)(), C.c_int64(0)
        t0 = time.perf_counter()
        L.check(L.lib.ist_stitch_png(ctx, descs, ptrs, pitches, n, S._DIRECTIONS[direction], S._MODES[o["mode"]], 0.0, C.byref(lim), S._filter_of(o), C.byref(cplan), C.byref(out), C.byref(length)))
        t = time.perf_counter() - t0
        if r:
            ts.append(t)
        ln = length.value
        L.lib.ist_plan_free(C.byref(cplan))
        L.lib.ist_free(C.cast(out, C.c_void_p))
    print("9 photo-like 12 MP images %-10s -> PNG: %6.2f ms per call  (file %d bytes)  %s" % (direction, sorted(ts)[len(ts) // 2] * 1e3, ln,
          "bands" if L.lib.ist_debug_duplex_stitches() > before else "upload all, launch, encode"), flush=True)
