#!/usr/bin/env python3
"""GPU JPEG decode of the bench's nine photo-like 12 MP files, with the C side's lap timing (IST_TIMING=1 prints
huffman/* laps to stderr).  usage: IST_TIMING=1 python tools/exp_huff.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import imagestitching_amd as ist  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
paths = [bench.photo_jpeg(k, w, h) for k, (w, h) in enumerate(bench.UNIFORM)]     # the file bytes
for r in range(reps):
    t0 = time.perf_counter()
    imgs = ist.decode_files_device(paths)
    import torch
    torch.cuda.synchronize()
    print("rep %d: decode_files_device %.2f ms" % (r, (time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
    del imgs
