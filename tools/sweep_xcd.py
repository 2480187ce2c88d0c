#!/usr/bin/env python3
"""A/B of the XCD-aware tile order (IST_XCD_ROTATE, ist_compile.cpp tile table) on the resampling strips, interleaved rounds in
one process, outputs compared bit for bit.  usage: python tools/sweep_xcd.py [rounds]"""
import os
import sys

os.environ["IST_TUNING"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

MIXED = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
for name, sizes, direction, opts in (("mixed_horizontal", MIXED, "horizontal", {"filter": "bilinear"}), ("mixed_vertical", MIXED, "vertical", {"filter": "bilinear"}),
                                     ("mixed_vertical_max", MIXED, "vertical", {"filter": "bilinear", "mode": "max"}),
                                     ("mixed_horizontal_max", MIXED, "horizontal", {"filter": "bilinear", "mode": "max"})):
    imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in sizes]
    jobs = {}
    for v in (0, 1):
        os.environ["IST_XCD_ROTATE"] = str(v)
        p, jobs[v] = st.compile(imgs, direction, opts)
    sets = [[torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev) for (w, h) in sizes] for _ in range(3)]
    outs = [torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev) for _ in range(3)]
    jobs[0].launch(sets[0], outs[0]); jobs[1].launch(sets[0], outs[1])
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]), name
    for _ in range(300):
        jobs[0].launch(sets[0], outs[0])
    torch.cuda.synchronize()
    res = {0: [], 1: []}
    for r in range(rounds):
        for v in (0, 1):
            for i in range(10):
                jobs[v].launch(sets[i % 3], outs[i % 3])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(40):
                jobs[v].launch(sets[i % 3], outs[i % 3])
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) * 1e3 / 40)
    B = jobs[0].info["algorithmic_bytes"]
    for v in (0, 1):
        t = sorted(res[v])
        print("%-22s xcd_rotate=%d  median %.1f us  min %.1f  max %.1f  -> %.3f of 8 TB/s" % (name, v, t[len(t) // 2], t[0], t[-1], B / (t[len(t) // 2] * 1e-6) / 8e12), flush=True)
    del sets, outs, jobs
    torch.cuda.empty_cache()
