#!/usr/bin/env python3
"""A few launches of the mixed-size vertical stitch (for rocprofv3 --pmc runs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

MIXED = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
direction = sys.argv[1] if len(sys.argv) > 1 else "vertical"
p, job = st.compile([{"width": w, "height": h, "opaque": True} for (w, h) in MIXED], direction, {"filter": "bilinear"})
srcs = [torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device=dev) for (w, h) in MIXED]
out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev)
for _ in range(5):
    job.launch(srcs, out)
torch.cuda.synchronize()
print(job.info)
