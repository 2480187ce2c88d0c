#!/usr/bin/env python3
"""PNG export of the 4032 x 27216 canvas of BASELINE configs[1]: stored vs compressed form, three kinds of content."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagestitching_amd as ist

dev = torch.device("cuda", 0)
W, H = 4032, 27216
yy = torch.arange(H, device=dev, dtype=torch.float32)[:, None]
xx = torch.arange(W, device=dev, dtype=torch.float32)[None, :]
photo = torch.stack([128 + 90 * torch.sin(xx / 37 + yy / 91), 128 + 80 * torch.cos(xx / 53 - yy / 29), 100 + 0.03 * xx + 0.002 * yy, torch.full((H, W), 255.0, device=dev)], -1)
photo[..., :3] += torch.randn((H, W, 3), device=dev) * 2.0
photo = photo.clamp(0, 255).to(torch.uint8).contiguous()
rnd = torch.randint(0, 256, (H, W, 4), dtype=torch.uint8, device=dev)
flat = torch.full((H, W, 4), 255, dtype=torch.uint8, device=dev)
flat[::37, 100:900, :3] = 30
for name, canvas in (("random", rnd), ("photo-like", photo), ("flat", flat)):
    for level in (0, 1):
        out = None
        ts = []
        for r in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            buf, n = ist.encode_png_device(canvas, out=out, level=level)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            out = None
        ts.sort()
        print("%-10s level %d: %8.2f ms (min of 6; includes scratch hipMalloc + host combine)  %12d bytes  ratio %.4f" % (name, level, ts[0], n, n / canvas.numel()), flush=True)
