#!/usr/bin/env python3
"""The canvas pitch and the paths that are not a flat copy (tools/exp/hbm_ceiling.cpp: a workgroup's eight 1 KiB stores run at 0.87 of
8 TB/s when their addresses are congruent mod 16 KiB and at 0.73 when they are not).  bench.py's four plans with the canvas rows padded
to the next multiple of 4 KiB / 16 KiB; sources dense.  usage: python tools/exp_canvas_pitch.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import imagestitching_amd as ist  # noqa: E402

dev = torch.device("cuda", 0)
st = ist.Stitcher(0)


def up(v, m):
    return (v + m - 1) // m * m


for name, sizes, direction in (("uniform_vertical", bench.UNIFORM, "vertical"), ("uniform_horizontal", bench.UNIFORM, "horizontal"),
                               ("mixed_vertical", bench.MIXED, "vertical"), ("mixed_horizontal", bench.MIXED, "horizontal")):
    imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in sizes]
    p, job = st.compile(imgs, direction, {"filter": "bilinear"})
    B = job.info["algorithmic_bytes"]
    srcs = [bench.synth(k, w, h, dev) for k, (w, h) in enumerate(sizes)]
    row = p.canvas_w * 4
    want = None
    for label, pitch in (("dense", row), ("4 KiB", up(row, 4096)), ("16 KiB", up(row, 16384)), ("32 KiB", up(row, 32768))):
        raw = torch.empty((p.canvas_h * pitch + 65536,), dtype=torch.uint8, device=dev)
        off = (-raw.data_ptr()) % 32768
        out = raw[off:off + p.canvas_h * pitch].view(p.canvas_h, pitch // 4, 4)[:, :p.canvas_w]
        for _ in range(400):
            job.launch(srcs, out)
        torch.cuda.synchronize()
        if want is None:
            want = out.clone()
        same = bool(torch.equal(out, want))
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                job.launch(srcs, out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 40)
        t = sorted(ts)[2]
        print("%-19s canvas rows %7d B (%-6s): %7.1f us  %.3f of 8 TB/s  same pixels: %s" % (name, pitch, label, t, B / (t * 1e-6) / 8e12, same), flush=True)
        del raw, out
    del srcs, want
    torch.cuda.empty_cache()
