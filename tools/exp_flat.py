#!/usr/bin/env python3
"""A vertical strip of equal-width images with DENSE rows (pitch = 4 * width on both sides) is a copy of contiguous byte ranges, so
the rows the kernel walks need not be the image's rows.  tools/exp_pitch.py: a 16 KiB / 32 KiB row pitch runs at 0.83-0.84 of 8 TB/s where
4032- and 8000-pixel rows run at 0.81 / 0.77.  This emulates "walk the same bytes as rows of VW pixels" through the public op-list API
(ist_job_create): per image a head row, whole rows and a tail row of a virtual canvas VW pixels wide.  The result must be the same bytes.
usage: python tools/exp_flat.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402
from imagestitching_amd import _lib as L  # noqa: E402

dev = torch.device("cuda", 0)
st = ist.Stitcher(0)


def time_it(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[2]


def run(n, w, h, reps, vws):
    Lb = w * h * 4
    total = n * Lb
    # the images, each in its own allocation placed so that (address - D_i) is a multiple of every VW*4 tried (<= 64 KiB)
    arena = torch.empty(n * (Lb + (1 << 17)) + (1 << 17), dtype=torch.uint8, device=dev)
    arena.random_(0, 256)
    base = (arena.data_ptr() + 65535) // 65536 * 65536 - arena.data_ptr()
    offs = []
    at = base
    for i in range(n):
        D = i * Lb
        at = (at + 65535) // 65536 * 65536 + D % 65536
        offs.append(at)
        at += Lb
    srcs = [arena[o:o + Lb].view(h, w, 4) for o in offs]
    canvas = torch.empty(total + (1 << 16), dtype=torch.uint8, device=dev)
    out = canvas[:total].view(n * h, w, 4)
    # (a) the shipped plan
    imgs = [{"width": w, "height": h, "opaque": True}] * n
    p, job = st.compile(imgs, "vertical", {"filter": "bilinear"})
    B = job.info["algorithmic_bytes"]
    t = time_it(lambda: job.launch(srcs, out), reps)
    want = out.clone()
    print("%2d x %dx%d shipped plan (rows of %d px): %9.1f us -> %.3f of 8 TB/s  (%d tiles)" % (n, w, h, w, t, B / (t * 1e-6) / 8e12, job.info["n_tiles"]), flush=True)
    for vw in vws:
        P = vw * 4
        vh = (total + P - 1) // P
        ops = []
        fill = L.Op()
        fill.kind = 0
        fill.d[0], fill.d[1], fill.d[2], fill.d[3] = 0, 0, vw, vh
        fill.m[0] = fill.m[3] = 1.0
        fill.rgba[0] = fill.rgba[1] = fill.rgba[2] = fill.rgba[3] = 255
        ops.append(fill)
        descs = (L.ImageDesc * n)()
        ptrs, pitches = [], []
        for i in range(n):
            D = i * Lb
            hx = (D % P) // 4
            y0 = D // P
            rows = (D % P + Lb + P - 1) // P                  # rows of the virtual image (first one starts at hx)
            descs[i].width, descs[i].height, descs[i].opaque = vw, rows, 1
            ptrs.append(arena.data_ptr() + offs[i] - (D % P))
            pitches.append(P)
            pieces = []
            first_full = 0
            if hx:
                pieces.append((hx, 0, min(vw - hx, Lb // 4), 1))
                first_full = 1
            left = Lb // 4 - (pieces[0][2] if pieces else 0)
            nfull = left // vw
            if nfull:
                pieces.append((0, first_full, vw, nfull))
            tail = left - nfull * vw
            if tail:
                pieces.append((0, first_full + nfull, tail, 1))
            for (x, y, ww, hh) in pieces:
                o = L.Op()
                o.kind, o.image = 1, i
                o.m[0] = o.m[3] = 1.0
                o.s[0], o.s[1], o.s[2], o.s[3] = x, y, ww, hh
                o.d[0], o.d[1], o.d[2], o.d[3] = x, y0 + y, ww, hh
                ops.append(o)
        arr = (L.Op * len(ops))(*ops)
        vjob = st.compile_ops(vw, vh, arr, len(ops), descs, n, "bilinear", clear=(0, 0, 0, 0))
        canvas.zero_()
        fn = lambda: vjob.launch_ptrs(ptrs, pitches, canvas.data_ptr(), P, torch.cuda.current_stream().cuda_stream)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        same = bool(torch.equal(out, want))
        t = time_it(fn, reps)
        i = vjob.info
        print("%2d x %dx%d as rows of %5d px (%d ops): %9.1f us -> %.3f of 8 TB/s  same bytes: %s  (tiles %d: copy %d sample %d general %d fill %d)" % (
            n, w, h, vw, len(ops), t, B / (t * 1e-6) / 8e12, same, i["n_tiles"], i["tiles_copy"], i["tiles_sample"], i["tiles_general"], i["tiles_fill"]), flush=True)
        vjob.close()
    del srcs, out, canvas, arena
    torch.cuda.empty_cache()


if len(sys.argv) > 1 and sys.argv[1] == "shapes":
    # IST_TUNING=1 IST_COPY_TILE=<shape> python tools/exp_flat.py shapes   (one tile shape per process: the knob is read once)
    print("IST_COPY_TILE=%s" % os.environ.get("IST_COPY_TILE", "(default 256x8)"), flush=True)
    run(9, 4032, 3024, 40, [6144, 8192, 12288])
    run(64, 8000, 6000, 5, [8192, 12288])
else:
    run(9, 4032, 3024, 40, [4096, 8192, 2048, 16384])
    run(64, 8000, 6000, 5, [4096, 8192, 16384])
