#!/usr/bin/env python3
"""Sweep of the LDS footprint budget (IST_LDS_BUDGET bytes) for the bilinear SAMPLE_LDS path on the mixed-size
variant of BASELINE configs 2/3; interleaved rounds in one process."""
import os
import sys

os.environ["IST_TUNING"] = "1"      # the launch-time knobs are only read in tuning mode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

MIXED = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
budgets = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8192, 12288, 16384, 20480, 24576, 32768, 40960, 65536]
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in MIXED]
nsets = 3
sets = [[torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev) for (w, h) in MIXED] for _ in range(nsets)]
for direction in ("vertical", "horizontal"):
    jobs, res = {}, {}
    for b in budgets:
        os.environ["IST_LDS_BUDGET"] = str(b)
        p, jobs[b] = st.compile(imgs, direction, {"filter": "bilinear"})
    os.environ["IST_NO_LDS"] = "1"
    p, jobs["direct"] = st.compile(imgs, direction, {"filter": "bilinear"})
    del os.environ["IST_NO_LDS"]
    outs = [torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev) for _ in range(nsets)]
    ref = None
    for k, job in jobs.items():
        outs[0].fill_(0)
        job.launch(sets[0], outs[0])
        torch.cuda.synchronize()
        if ref is None:
            ref = outs[0].clone()
        else:
            assert int((outs[0].to(torch.int16) - ref.to(torch.int16)).abs().max()) <= 1, k
    for r in range(rounds):
        for k, job in jobs.items():
            for i in range(3):
                job.launch(sets[i % nsets], outs[i % nsets])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 30
            e0.record()
            for i in range(n):
                job.launch(sets[i % nsets], outs[i % nsets])
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(k, []).append(e0.elapsed_time(e1) * 1e3 / n)
    for k, job in jobs.items():
        v = sorted(res[k])
        B = job.info["algorithmic_bytes"]
        print("%-10s budget %-7s tiles=%6d  median %.1f us  min %.1f  -> %.0f GB/s (%.1f%%)" % (direction, k, job.info["n_tiles"], v[len(v) // 2], v[0], B / v[len(v) // 2] / 1e3, B / v[len(v) // 2] / 1e3 / 80), flush=True)
    del outs, jobs, ref
