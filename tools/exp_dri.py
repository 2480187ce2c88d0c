#!/usr/bin/env python3
"""Nine 12 MP photo-like JPEGs -> bitmaps in HBM (ist_decode_files_device), with and without restart intervals.
With DRI every interval is a unit of the GPU Huffman batch (ist_jpeg_gpu.hip); the table says what that costs or saves."""
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

import bench  # noqa: E402
import imagestitching_amd as ist  # noqa: E402
from imagestitching_amd import _lib as L  # noqa: E402


def reencode(blob, **kw):
    a = np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"))
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", quality=90, subsampling=2, **kw)
    return b.getvalue()


base = [bench.photo_jpeg(k, w, h) for k, (w, h) in enumerate(bench.UNIFORM)]
for name, kw in (("no restart intervals", None), ("one interval per MCU row", {"restart_marker_rows": 1}), ("one interval per 4 MCU rows", {"restart_marker_rows": 4}),
                 ("one interval per 32 MCUs", {"restart_marker_blocks": 32})):
    blobs = base if kw is None else [reencode(b, **kw) for b in base]
    out, _ = ist.decode_files_device(blobs)
    ist.set_phase_timing(True)
    before = L.lib.ist_debug_gpu_entropy_files()
    times, phases = [], None
    for i in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ist.decode_files_device(blobs, out=out)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        phases = ist.last_phase_times()
    ist.set_phase_timing(False)
    print(json.dumps({"files": name, "bytes": sum(len(b) for b in blobs), "ms_best": round(min(times), 3), "ms_median": round(sorted(times)[4], 3),
                      "on_gpu_per_call": (L.lib.ist_debug_gpu_entropy_files() - before) // 8,
                      "entropy_gpu_ms": round(phases["entropy_gpu"], 3), "host_decode_ms": round(phases["host_decode"], 3)}))
