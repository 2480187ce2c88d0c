#!/usr/bin/env python3
"""Extra measurements quoted in DESIGN.md (not the driver's bench line):
  config5  BASELINE configs[4] geometry on ONE GPU: 64 x 8000x6000 vertical -> 8000x384000 (3072 MP, 24.6 GB moved)
  host     PCIe-inclusive rate of the host-buffer path (ist_stitch_rgba8) on configs[1]
usage: python tools/bench_extra.py [config5] [host]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

what = sys.argv[1:] or ["config5", "host", "png", "files"]
dev = torch.device("cuda", 0)

if "config5" in what:
    st = ist.Stitcher(0)
    n, w, h = 64, 8000, 6000
    p, job = st.compile([{"width": w, "height": h, "opaque": True}] * n, "vertical", {"filter": "bilinear"})
    srcs = [torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev) for _ in range(n)]
    out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev)
    job.launch(srcs, out)
    torch.cuda.synchronize()
    for k in (0, 17, 63):
        assert torch.equal(out[h * k:h * (k + 1)], srcs[k])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        job.launch(srcs, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    B = job.info["algorithmic_bytes"]
    print("config5 1-GPU: canvas %dx%d, %d tiles, %.3f ms per stitch, %.0f GB/s (%.1f%% of 8 TB/s), %.0f MP/s"
          % (p.canvas_w, p.canvas_h, job.info["n_tiles"], ms, B / ms / 1e6, B / ms / 1e6 / 80, p.canvas_w * p.canvas_h / 1e6 / (ms * 1e-3)), flush=True)
    del srcs, out, job
    torch.cuda.empty_cache()

if "host" in what:
    px = [np.random.default_rng(1000 + k).integers(0, 256, (3024, 4032, 4), dtype=np.uint8) for k in range(9)]
    imgs = [{"width": 4032, "height": 3024, "data": a, "opaque": True} for a in px]
    r = ist.stitch(imgs, "vertical", {"filter": "bilinear"})          # warm-up: scratch allocation
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = ist.stitch(imgs, "vertical", {"filter": "bilinear"})
        ts.append(time.perf_counter() - t0)
    assert np.array_equal(r["data"][:3024], px[0])
    t = sorted(ts)[len(ts) // 2]
    print("host path (pageable numpy in -> HIP -> numpy out, incl. plan, H2D, D2H, output copy): %.1f ms per stitch = %.0f MP/s, %.1f GB/s of PCIe payload"
          % (t * 1e3, 109.734912 / t, 2 * 438.94e6 / t / 1e9), flush=True)

if "png" in what:
    canvas = torch.randint(0, 256, (27216, 4032, 4), dtype=torch.uint8, device=dev)
    out, n = ist.encode_png_device(canvas)               # warm-up
    buf = torch.empty(out.numel() + 64, dtype=torch.uint8, device=dev)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, n = ist.encode_png_device(canvas, out=buf)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    print("png device-resident 4032x27216: %d bytes (%.4f x raw), %.3f ms per encode incl. host checksum combine = %.0f MP/s, %.0f GB/s read+write"
          % (n, n / canvas.numel(), t * 1e3, 109.734912 / t, (canvas.numel() + n) / t / 1e9), flush=True)

if "files" in what:
    import io
    import tempfile
    from PIL import Image
    tmp = tempfile.mkdtemp()
    paths = []
    yy, xx = np.mgrid[0:3024, 0:4032]
    for k in range(9):
        a = np.stack([(xx * (k + 1) // 7 + yy) % 256, (xx + yy * (k + 2) // 5) % 256, (xx * 3 + yy * 2 + 31 * k) % 256], -1).astype(np.uint8)
        a = np.clip(a.astype(np.int16) + np.random.default_rng(k).integers(-12, 13, a.shape), 0, 255).astype(np.uint8)
        p = os.path.join(tmp, "in%d.jpg" % k)
        Image.fromarray(a).save(p, "JPEG", quality=90, subsampling=2)
        paths.append(p)
    size = sum(os.path.getsize(p) for p in paths)
    res = ist.stitch_files(paths, "vertical")            # warm-up
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        res = ist.stitch_files(paths, "vertical", out_path=os.path.join(tmp, "out.png"))
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    print("files: 9 x 12 MP JPEG (%.1f MB total) -> %dx%d PNG (%.0f MB): %.0f ms end to end (file read, Huffman on %d host threads, GPU "
          "reconstruct + stitch + PNG, PNG write) = %.0f MP/s" % (size / 1e6, res["width"], res["height"], len(res["png"]) / 1e6, t * 1e3, 9, 109.734912 / t), flush=True)
    t0 = time.perf_counter()
    one = ist.decode_image(open(paths[0], "rb").read())
    print("one 12 MP JPEG decode (host Huffman + H2D + GPU + D2H): %.0f ms" % ((time.perf_counter() - t0) * 1e3))
