#!/usr/bin/env python3
"""File to file: 9 x 12 MP JPEG (photo-like) -> one PNG, phase times (IST_TIMING=1 prints them from the C side)."""
import os, sys, time, tempfile
os.environ["IST_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
import imagestitching_amd as ist
tmp = tempfile.mkdtemp()
paths = []
yy, xx = np.mgrid[0:3024, 0:4032]
for k in range(9):
    a = np.stack([128 + 90 * np.sin(xx / (37.0 + k) + yy / 91.0), 128 + 80 * np.cos(xx / 53.0 - yy / (29.0 + k)), 100 + 0.03 * xx + 0.02 * yy], -1)
    a = (a + np.random.default_rng(k).normal(0, 3.0, a.shape)).clip(0, 255).astype(np.uint8)
    p = os.path.join(tmp, "in%d.jpg" % k)
    Image.fromarray(a).save(p, "JPEG", quality=90, subsampling=2)
    paths.append(p)
print("inputs: %.1f MB of JPEG" % (sum(os.path.getsize(p) for p in paths) / 1e6), flush=True)
for level in (0, 1):
    for rep in range(3):
        t0 = time.perf_counter()
        res = ist.stitch_files(paths, "vertical", {"pngLevel": level}, copy=False)
        dt = time.perf_counter() - t0
        print("level %d rep %d: %.1f ms end to end (incl. file reads, Python), PNG %.1f MB" % (level, rep, dt * 1e3, len(res["png"]) / 1e6), file=sys.stderr, flush=True)

# ---- the screenshot case: nine 1179 x 2556 PNG screenshots (phone screen size), flat areas + "text"
rng = np.random.default_rng(5)
spaths = []
for k in range(9):
    h, w = 2556, 1179
    a = np.full((h, w, 3), 255, np.uint8)
    for _ in range(600):
        y = int(rng.integers(0, h - 12)); x = int(rng.integers(0, w - 200))
        a[y:y + int(rng.integers(2, 12)), x:x + int(rng.integers(20, 200))] = rng.integers(0, 120, 3, dtype=np.uint8)
    a[200:420] = (40 + 10 * k, 120, 200)
    p = os.path.join(tmp, "shot%d.png" % k)
    Image.fromarray(a).save(p, "PNG")
    spaths.append(p)
print("screenshots: %.2f MB of PNG" % (sum(os.path.getsize(p) for p in spaths) / 1e6), file=sys.stderr, flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    res = ist.stitch_files(spaths, "vertical", {"pngLevel": 1})
    dt = time.perf_counter() - t0
    print("screenshots rep %d: %.1f ms end to end, %dx%d, PNG %.2f MB" % (rep, dt * 1e3, res["width"], res["height"], len(res["png"]) / 1e6), file=sys.stderr, flush=True)
