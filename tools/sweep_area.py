#!/usr/bin/env python3
"""A/B of the streamed box filter's tile shape (IST_TUNING knobs), every arm in its own child process, interleaved rounds.
gpurun -- 'python3 tools/sweep_area.py > gpurun_out/r03_sweep_area.jsonl'"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import torch
import imagestitching_amd as ist
UNI = [(4032, 3024)] * 9
plans = [("ios", {"filter": "area", "platform": "ios", "superSample": 1, "edgeAA": False}),
         ("android", {"filter": "area", "platform": "android", "superSample": 1, "edgeAA": False}),
         ("4x", {"filter": "area", "maxSide": 6804}),
         ("ios_bilinear", {"filter": "bilinear", "platform": "ios", "superSample": 1})]
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
srcs = [torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev) for (w, h) in UNI]
imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in UNI]
out = {}
for name, o in plans:
    p, job = st.compile(imgs, "vertical", o)
    dst = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev)
    for _ in range(300):
        job.launch(srcs, dst)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        job.launch(srcs, dst)
    e1.record()
    torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) * 10, 1)
print(json.dumps(out))
''' % ROOT

V = os.path.join(ROOT, "tools", "exp", "variants")
ARMS = [("default (5 waves per SIMD, 96 VGPRs)", {}), ("h32", {"IST_AREA_TILE_H": "32"}), ("p2", {"IST_AREA_PASSES": "2"})]
# variant builds of the library (make ... CXXFLAGS+=-DIST_AREA_WAVES=6 OUT=tools/exp/variants/libimagestitch_areaw6.so), when present
for w in (6, 7):
    lib = os.path.join(V, "libimagestitch_areaw%d.so" % w)
    if os.path.exists(lib):
        ARMS.append(("%d waves per SIMD (spills)" % w, {"IST_LIB_PATH": lib}))
if "--with-general" in sys.argv:
    ARMS.append(("per_pixel_general_path", {"IST_NO_LDS": "1"}))      # what filter 'area' cost before the streamed path existed
for rnd in range(2):
    for name, env in ARMS:
        e = dict(os.environ, IST_TUNING="1", **env)
        r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=600)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:]
        print(json.dumps({"round": rnd, "arm": name, "us": line}), flush=True)
