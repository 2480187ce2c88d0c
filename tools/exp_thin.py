"""Where the flat form of a vertical strip stops paying: images too short for rows of 32 KiB (head and tail rows of every image are tiles with one
row in eight used).  Run once as is and once with IST_TUNING=1 IST_FLAT=0 (row form).  usage: python tools/exp_thin.py"""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
import imagestitching_amd as ist
from imagestitching_amd import _lib as L
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
for (n, w, h) in ((9, 5120, 3000), (12, 3072, 3000), (20, 2048, 3000), (9, 4096, 3000), (128, 4032, 100), (100, 4032, 250), (50, 4032, 500), (25, 4032, 1000), (12, 4032, 2000), (128, 8000, 100), (64, 8000, 400), (120, 2000, 400), (60, 2000, 1600)):
    imgs = [{"width": w, "height": h, "opaque": True}] * n
    p, job = st.compile(imgs, "vertical", {"filter": "bilinear"})
    srcs = [torch.empty((h, w, 4), dtype=torch.uint8, device=dev).random_(0, 256) for _ in range(n)]
    out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev)
    b = L.lib.ist_debug_flat_launches()
    for _ in range(200): job.launch(srcs, out)
    torch.cuda.synchronize()
    flat = L.lib.ist_debug_flat_launches() - b
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): job.launch(srcs, out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 10)
    print("%d x %dx%d: %s form, %.1f us, %.3f of 8 TB/s" % (n, w, h, "flat" if flat else "row", sorted(ts)[2], job.info["algorithmic_bytes"] / (sorted(ts)[2] * 1e-6) / 8e12), flush=True)
