#!/usr/bin/env python3
"""A/B sweep of copy-kernel variants (IST_VARIANT / IST_COPY_TILE / IST_PERSIST_BLOCKS / IST_FULL_KERNEL knobs),
interleaved rounds in one process.  usage: python tools/sweep_variants.py [rounds] [direction]"""
import os
import sys

os.environ["IST_TUNING"] = "1"      # the launch-time knobs are only read in tuning mode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import imagestitching_amd as ist  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
directions = sys.argv[2].split(",") if len(sys.argv) > 2 else ["vertical", "horizontal"]
# (label, tile, variant, persist_blocks, full_kernel)
# (label, IST_COPY_TILE, IST_VARIANT (0 shipped, 1 = 8 rows/wave, 2 = consecutive rows, 3 = no nt hints; +100 persistent),
#  IST_PERSIST_BLOCKS, IST_FULL_KERNEL, IST_DYN_LDS)
CONFIGS = [("shipped 256x8", "256x8", 0, 0, 0, 0), ("8 rows/wave 256x32", "256x32", 1, 0, 0, 0), ("consecutive 256x32", "256x32", 2, 0, 0, 0),
           ("no nt 256x32", "256x32", 3, 0, 0, 0), ("persistent 256x8", "256x8", 100, 2048, 0, 0), ("shipped, full kernel", "256x8", 0, 0, 1, 0),
           ("shipped 512x4", "512x4", 0, 0, 0, 0), ("1 row/wave 256x4", "256x4", 0, 0, 0, 0), ("4 rows/wave 256x16", "256x16", 1, 0, 0, 0)]
dev = torch.device("cuda", 0)
st = ist.Stitcher(0)
imgs = [{"width": 4032, "height": 3024, "opaque": True} for _ in range(9)]
nsets = 3
sets = [[torch.randint(0, 256, (3024, 4032, 4), dtype=torch.uint8, device=dev) for _ in range(9)] for _ in range(nsets)]


def setenv(cfg):
    _, tile, v, pb, full, dyn = cfg
    if "nobands" in cfg[0]:
        os.environ["IST_NO_BANDS"] = "1"
    else:
        os.environ.pop("IST_NO_BANDS", None)
    os.environ["IST_DYN_LDS"] = str(dyn)
    os.environ["IST_COPY_TILE"] = tile
    os.environ["IST_VARIANT"] = str(v)
    os.environ["IST_PERSIST_BLOCKS"] = str(pb or 2048)
    if full:
        os.environ["IST_FULL_KERNEL"] = "1"
    else:
        os.environ.pop("IST_FULL_KERNEL", None)


for direction in directions:
    jobs, res = {}, {}
    for cfg in CONFIGS:
        setenv(cfg)
        p, jobs[cfg[0]] = st.compile(imgs, direction, {"filter": "bilinear"})
    outs = [torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev) for _ in range(nsets)]
    # correctness of every variant once
    for cfg in CONFIGS:
        setenv(cfg)
        outs[0].fill_(0)
        jobs[cfg[0]].launch(sets[0], outs[0])
        torch.cuda.synchronize()
        ref = torch.cat(sets[0], 0 if direction == "vertical" else 1)
        assert torch.equal(outs[0], ref), cfg
        del ref
    for r in range(rounds):
        for cfg in CONFIGS:
            setenv(cfg)
            job = jobs[cfg[0]]
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
            res.setdefault(cfg[0], []).append(e0.elapsed_time(e1) * 1e3 / n)
    for cfg in CONFIGS:
        v = sorted(res[cfg[0]])
        print("%-10s %-22s median %.1f us  min %.1f us  -> %.0f GB/s (%.1f%% of 8 TB/s)" % (direction, cfg[0], v[len(v) // 2], v[0], 877879296 / v[len(v) // 2] / 1e3, 877879296 / v[len(v) // 2] / 1e3 / 80), flush=True)
    del outs, jobs
