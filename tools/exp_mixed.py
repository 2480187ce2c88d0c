#!/usr/bin/env python3
"""Times the bilinear SAMPLE_LDS path on three workloads (whatever libimagestitch.so is currently built)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagestitching_amd as ist
MIXED = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
ALL75 = [(4032, 3024)] * 8 + [(3024, 2268)]
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
tag = sys.argv[1] if len(sys.argv) > 1 else "base"
for name, sizes, direction in (("mixed_v", MIXED, "vertical"), ("mixed_h", MIXED, "horizontal"), ("all75_v", ALL75, "vertical"), ("all75_h", ALL75, "horizontal")):
    imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in sizes]
    p, job = st.compile(imgs, direction, {"filter": "bilinear"})
    sets = [[torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev) for (w, h) in sizes] for _ in range(3)]
    outs = [torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev) for _ in range(3)]
    ts = []
    for r in range(5):
        for i in range(3):
            job.launch(sets[i % 3], outs[i % 3])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30):
            job.launch(sets[i % 3], outs[i % 3])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 30)
    ts.sort()
    B = job.info["algorithmic_bytes"]
    print("%-6s %-8s median %.1f us min %.1f  %.0f GB/s  tiles copy=%d sample=%d" % (tag, name, ts[2], ts[0], B / ts[2] / 1e3, job.info["tiles_copy"], job.info["tiles_sample"]), flush=True)
    del sets, outs, job
    torch.cuda.empty_cache()
