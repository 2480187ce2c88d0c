#!/usr/bin/env python3
"""Does the ROW PITCH of the caller's buffers matter to the copy path?  (tools/exp_config5.py: 64 x 8192x6000 runs at 0.845 of 8 TB/s,
64 x 8000x6000 at 0.76.)  Same images, same canvas size; only the pitch of sources and canvas varies (padding columns are never touched).
usage: python tools/exp_pitch.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

dev = torch.device("cuda", 0)
st = ist.Stitcher(0)


def pitched(h, w, pitch_px):
    base = torch.empty((h + 1, pitch_px, 4), dtype=torch.uint8, device=dev)
    base.random_(0, 256)
    return base[:h, :w]


def run(n, w, h, direction, pitches, reps):
    imgs = [{"width": w, "height": h, "opaque": True}] * n
    p, job = st.compile(imgs, direction, {"filter": "bilinear"})
    B = job.info["algorithmic_bytes"]
    for sp, dp in pitches:
        srcs = [pitched(h, w, sp) for _ in range(n)]
        out = pitched(p.canvas_h, p.canvas_w, dp)
        for _ in range(max(3, 300 // max(1, n * w * h // 12000000))):
            job.launch(srcs, out)
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                job.launch(srcs, out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        ts.sort()
        print("%2d x %dx%d %-10s source pitch %6d px, canvas pitch %6d px: median %9.1f us  -> %.3f of 8 TB/s" % (n, w, h, direction, sp, dp, ts[2], B / (ts[2] * 1e-6) / 8e12), flush=True)
        del srcs, out
        torch.cuda.empty_cache()


if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    # which property of the pitch is it?  (bytes = 4 * px: 4032 px = 15.75 KiB, 4096 = 16 KiB, 4352 = 17 KiB, 4608 = 18 KiB, 5120 = 20 KiB, 6144 = 24 KiB)
    px = [4032, 4048, 4064, 4096, 4128, 4160, 4224, 4352, 4608, 4864, 5120, 6144, 7168, 8192, 12288, 16384]
    run(9, 4032, 3024, "vertical", [(q, q) for q in px], 40)
else:
    run(9, 4032, 3024, "vertical", [(4032, 4032), (4096, 4096), (4064, 4064), (4160, 4160), (5120, 5120), (4096, 4032), (4032, 4096)], 40)
    run(9, 4032, 3024, "horizontal", [(4032, 36288), (4096, 36864), (4096, 36288), (4032, 36864), (4096, 40960)], 40)
    run(64, 8000, 6000, "vertical", [(8000, 8000), (8192, 8192), (8064, 8064), (8192, 8000), (8000, 8192)], 5)
