#!/usr/bin/env python3
"""bench.py — stitched output megapixels / second on MI355X, with the HBM roofline of the resample+blit kernel and the
CPU oracle timed beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`--gpus N` with N > 1 ALWAYS runs N ranks, one per GPU: under torchrun the ranks exist already (WORLD_SIZE is set); invoked
plainly, this process — before it imports torch or touches HIP — starts N rank processes of itself with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set, relays rank 0's JSON line and exits with the ranks' status (launch_ranks).  The line proves the
rank count: extra.ranks_seen is an all-gather of every rank's device and PCI bus id.

N = 1  : BASELINE.json configs[1] — 9 x 4032x3024 RGBA vertical stitch, bilinear, inputs resident in HBM, one fused
         launch per step (a "step" = one whole stitch).  Three buffer sets are rotated.
N > 1  : BASELINE.json configs[3] — the same stitch with image i on GPU i mod N, bands gathered to GPU 0 over RCCL
         (strong scaling: total work fixed).
`value` is always the region BASELINE's contract names: inputs already resident in HBM.  DESIGN.md section 6 explains why
that region cannot beat one GPU (a 0.14 ms job against a >= 0.3 ms gather over one xGMI link per sender) and on which
region the >= 6x target is claimed instead; every run therefore also reports, in extra.regions, the SAME stitch timed
  from_pinned_host  each GPU first uploads the source rows it renders from pinned host memory (its own PCIe link)
  host_in_host_out  ... and delivers its finished bands into pinned host memory (no gather: the export is host-destined)
  from_jpeg         each GPU first decodes its images from JPEG bytes (GPU Huffman + IDCT + colour)
and, in extra.scaling, the ONE-GPU form of every region timed by rank 0 ALONE in the same process on the same box while the
other ranks block on the rendezvous store (no GPU work) - so every region carries {ms_1gpu, ms_Ngpu, speedup} from one lease.
BASELINE configs[4] (64 x 8000x6000 -> 8000x384000, 24.6 GB moved per stitch) has its own legs: extra.config5_single_gpu in
the N = 1 line, extra.config5 (8 images per GPU, resident and host_in_host_out, + the same-lease one-GPU comparators) in the
N > 1 line, and configs[2]'s horizontal strip runs cut by rows (extra.regions[*/rows_horizontal]).
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this pool (set before HIP loads)
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X spec peak, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
UNIFORM = [(4032, 3024)] * 9
CONFIG5 = [(8000, 6000)] * 64            # BASELINE configs[4] (SURVEY 8d "Config 5"): 64 x 48 MP -> 8000x384000, 12.288 GB in + as much out
MIXED = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
KERNEL_SOURCES = ["imagestitching_amd/csrc/ist_kernels.hip", "imagestitching_amd/csrc/ist_compile.cpp", "imagestitching_amd/csrc/ist_internal.h",
                  "imagestitching_amd/csrc/ist_launch.h"]


class StdoutGuard:
    """Native libraries print banners to fd 1 (RCCL: its version block at communicator creation).  Everything written to
    stdout while the guard is active goes to stderr; emit() restores fd 1 and prints the ONE JSON line of the contract."""

    def __init__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        print(line, flush=True)


def kernel_source_sha():
    """identifies the kernel + tiling a PMC measurement belongs to (profiles/*_pmc.json carry it)"""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def synth_np(k, w, h):
    """image k = default_rng(1000+k) uniform bytes, alpha forced to 255."""
    import numpy as np
    a = np.random.default_rng(1000 + k).integers(0, 256, (h, w, 4), dtype=np.uint8)
    a[..., 3] = 255
    return a


def synth(k, w, h, device):
    import torch
    return torch.from_numpy(synth_np(k, w, h)).to(device)


def photo_jpeg(k, w, h):
    """a photo-like 12 MP JPEG (smooth structure + sensor-like noise, quality 90, 4:2:0): what a phone hands the page"""
    import io
    import numpy as np
    from PIL import Image
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([128 + 90 * np.sin(xx / (37.0 + k) + yy / 91.0), 128 + 80 * np.cos(xx / 53.0 - yy / (29.0 + k)), 100 + 0.03 * xx + 0.02 * yy], -1)
    a = (a + np.random.default_rng(k).normal(0, 3.0, a.shape)).clip(0, 255).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", quality=90, subsampling=2)
    return b.getvalue()


PREROLL_MS = 60.0     # untimed back-to-back launches before every timed region: after the seconds of host-side input synthesis the
                      # GPU sits in a low power state and needs ~10 ms of load to reach the clock it then holds (measured,
                      # tools/exp_sustained.py: launches 0-50 of the mixed strip run 156 us, launches 200-300 117 us)


PREROLL_LAUNCHES = 400


def preroll(fn, torch, ms=PREROLL_MS):
    """brings the chip to its sustained clock: `fn()` (one step) repeated for at least `ms` of GPU time, untimed"""
    t0 = time.perf_counter()
    while True:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        if (time.perf_counter() - t0) * 1e3 >= ms:
            return


def time_job(job, sets, outs, steps, warmup, torch):
    """steps launches, rotating buffer sets; returns (wall seconds, event milliseconds)."""
    n = len(sets)
    for i in range(PREROLL_LAUNCHES):        # a fixed count here (about PREROLL_MS at 0.14 ms each): the rocprofv3 summaries count dispatches
        job.launch(sets[i % n], outs[i % n])
    torch.cuda.synchronize()
    for i in range(warmup):
        job.launch(sets[i % n], outs[i % n])
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                       # the launches below go to torch's current stream (passed through the C-ABI)
    for i in range(steps):
        job.launch(sets[i % n], outs[i % n])
    ev1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return t1 - t0, ev0.elapsed_time(ev1)


# ---------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(budget_s=10.0):
    """The CPU oracle (a port: the reference's raster is the closed WeChat client) on the same workload, all host
    cores by output row bands, bounded to ~budget_s of CPU work.  Rebuilt on this box with -march=native (BASELINE.md
    section 3); falls back to the portable build that travelled with the repo."""
    import subprocess
    import numpy as np
    build = "-O3 -march=x86-64-v3 (portable build)"
    try:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        os.environ["IST_ORACLE_LIB"] = os.path.join(ROOT, "oracle", "libist_oracle_native.so")
        build = "-O3 -march=native, built on this box"
    except Exception:
        pass
    from oracle import oracle as O
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    px = [synth_np(k, 4032, 3024) for k in range(9)]
    descs = [{"width": 4032, "height": 3024} for _ in px]
    rc, pd, rl = O.plan(descs, "vertical", "min", 0, O.lifted_limits(1.0))
    out = np.empty((int(pd["canvas_h"]), int(pd["canvas_w"]), 4), np.uint8)
    O.render(pd, rl, descs, px, "bilinear", threads, out=out)          # warm-up (page faults)
    reps, t0 = 0, time.perf_counter()
    while True:
        O.render(pd, rl, descs, px, "bilinear", threads, out=out)
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= 50:
            break
    dt = (time.perf_counter() - t0) / reps
    mp = pd["canvas_w"] * pd["canvas_h"] / 1e6
    t1 = time.perf_counter()
    O.render(pd, rl, descs, px, "bilinear", 1, out=out)                # single-thread figure on one repetition, for the record
    st = time.perf_counter() - t1
    quota = None                                   # the container's CPU-time quota, if any (cgroup v2 cpu.max = "<quota us> <period us>")
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        quota = None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        pass
    res = {"value": round(mp / dt, 1), "unit": "MP/s", "cores": threads, "cpu_time_quota_cpus": quota, "kind": "port", "build": build,
           "sample": "%d x the full 9x4032x3024 vertical bilinear stitch (109.7 MP each), oracle/ist_oracle.c, %d threads by row bands" % (reps, threads),
           "single_thread_MPs": round(mp / st, 1)}
    res["cairo"] = cairo_leg(px)
    return res


def cairo_leg(px, budget_s=4.0):
    """Optional: the same nine 1:1 draws on a cairo image surface — a real Canvas-2D backend — on
    one host thread, if libcairo.so.2 can be dlopen'ed on this box.  Never required; None when absent."""
    import ctypes as C
    import numpy as np
    try:
        cairo = C.CDLL("libcairo.so.2")
    except OSError:
        return None
    try:
        for name, res, args in (("cairo_image_surface_create", C.c_void_p, [C.c_int, C.c_int, C.c_int]),
                                ("cairo_image_surface_create_for_data", C.c_void_p, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
                                ("cairo_create", C.c_void_p, [C.c_void_p]), ("cairo_set_source_surface", None, [C.c_void_p, C.c_void_p, C.c_double, C.c_double]),
                                ("cairo_get_source", C.c_void_p, [C.c_void_p]), ("cairo_pattern_set_filter", None, [C.c_void_p, C.c_int]),
                                ("cairo_rectangle", None, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]), ("cairo_fill", None, [C.c_void_p]),
                                ("cairo_set_source_rgb", None, [C.c_void_p, C.c_double, C.c_double, C.c_double]), ("cairo_paint", None, [C.c_void_p]),
                                ("cairo_surface_flush", None, [C.c_void_p]), ("cairo_destroy", None, [C.c_void_p]), ("cairo_surface_destroy", None, [C.c_void_p]),
                                ("cairo_surface_status", C.c_int, [C.c_void_p])):
            f = getattr(cairo, name)
            f.restype, f.argtypes = res, args
        w, h, n = 4032, 3024, len(px)
        FORMAT_ARGB32, FILTER_BILINEAR = 0, 4
        # cairo's ARGB32 is premultiplied BGRA in memory; the inputs are opaque, so a channel swap is the whole conversion
        srcs = [np.ascontiguousarray(a[..., [2, 1, 0, 3]]) for a in px]
        ssurf = [cairo.cairo_image_surface_create_for_data(a.ctypes.data, FORMAT_ARGB32, w, h, w * 4) for a in srcs]
        dst = cairo.cairo_image_surface_create(FORMAT_ARGB32, w, h * n)
        if cairo.cairo_surface_status(dst) != 0:
            return None

        def once():
            cr = cairo.cairo_create(dst)
            cairo.cairo_set_source_rgb(cr, 1.0, 1.0, 1.0)
            cairo.cairo_paint(cr)                                       # fillRect('#ffffff') over the canvas (index.js:1423-1424)
            for k, s in enumerate(ssurf):
                cairo.cairo_set_source_surface(cr, s, 0.0, float(h * k))
                cairo.cairo_pattern_set_filter(cairo.cairo_get_source(cr), FILTER_BILINEAR)
                cairo.cairo_rectangle(cr, 0.0, float(h * k), float(w), float(h))
                cairo.cairo_fill(cr)                                    # drawImage(img, 0,0,w,h, 0,h*k,w,h)
            cairo.cairo_surface_flush(dst)
            cairo.cairo_destroy(cr)
        once()
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s and reps < 20:
            once()
            reps += 1
        dt = (time.perf_counter() - t0) / max(reps, 1)
        for s in ssurf:
            cairo.cairo_surface_destroy(s)
        cairo.cairo_surface_destroy(dst)
        return {"MPs": round(w * h * n / 1e6 / dt, 1), "threads": 1, "reps": reps,
                "what": "cairo image surface (libcairo.so.2, pixman): white paint + nine 1:1 FILTER_BILINEAR draws, one thread"}
    except Exception as ex:       # informational only
        return {"error": repr(ex)}


def canvas_pitch_leg(st, dev, torch):
    """Informational: the three plans that are not a flat copy, with the caller's canvas rows padded to the next multiple of 4 KiB (the
    canvas pitch is the caller's choice, ist_job_launch takes it as it is; INTEGRATION.md "Row pitch").  A workgroup's 1 KiB stores are
    cheapest when they start on 256-byte boundaries and sit a multiple of 4 KiB apart (tools/exp/hbm_ceiling.cpp, tools/exp_canvas_pitch.py);
    a 25 320-pixel canvas row (101 280 bytes) offers neither."""
    out = {"what": "same jobs and sources as extra.<plan>, canvas rows padded to a multiple of 4096 bytes; kernel_us by events, 40 launches, median of 5"}
    for name, sizes, direction in (("uniform_horizontal", UNIFORM, "horizontal"), ("mixed_vertical", MIXED, "vertical"), ("mixed_horizontal", MIXED, "horizontal")):
        imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in sizes]
        p, job = st.compile(imgs, direction, {"filter": "bilinear"})
        srcs = [synth(k, w, h, dev) for k, (w, h) in enumerate(sizes)]
        canvas = job.empty_canvas()                  # the pitch the job itself asks for (ist_job_preferred_dst_pitch): here the next multiple of 4 KiB
        pitch = canvas.stride(0)
        assert pitch % 4096 == 0 and pitch >= p.canvas_w * 4
        for _ in range(PREROLL_LAUNCHES):
            job.launch(srcs, canvas)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                job.launch(srcs, canvas)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 40)
        us = sorted(ts)[2]
        out[name] = {"canvas_row_bytes": pitch, "dense_row_bytes": p.canvas_w * 4, "kernel_us": round(us, 2),
                     "frac": round(job.info["algorithmic_bytes"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        del srcs, canvas, job
        torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------- N = 1
def run_single(args):
    import numpy as np
    import torch
    import imagestitching_amd as ist
    from imagestitching_amd import _lib as L
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    st = ist.Stitcher(0)
    nsets = 3
    results = {}
    head_job = None
    for name, sizes, direction in (("uniform_vertical", UNIFORM, "vertical"), ("uniform_horizontal", UNIFORM, "horizontal"),
                                   ("mixed_vertical", MIXED, "vertical"), ("mixed_horizontal", MIXED, "horizontal")):
        if args.quick and name != "uniform_vertical":
            continue
        imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in sizes]
        p, job = st.compile(imgs, direction, {"filter": "bilinear"})
        sets = [[synth(9 * s + k, w, h, dev) for k, (w, h) in enumerate(sizes)] for s in range(nsets)]
        outs = [torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev) for _ in range(nsets)]
        steps = args.steps if name == "uniform_vertical" else max(10, args.steps // 2)
        flat0 = L.lib.ist_debug_flat_launches()
        wall, ev_ms = time_job(job, sets, outs, steps, args.warmup, torch)
        flat = L.lib.ist_debug_flat_launches() - flat0 == PREROLL_LAUNCHES + args.warmup + steps
        mp = p.canvas_w * p.canvas_h / 1e6
        k_us = ev_ms * 1e3 / steps
        results[name] = {"canvas": [p.canvas_w, p.canvas_h], "out_MP": round(mp, 3), "steps": steps,
                         "ms_per_step": wall * 1e3 / steps, "kernel_us": k_us, "MPs": mp / (wall / steps),
                         "algorithmic_bytes": job.info["algorithmic_bytes"],
                         "GBs": job.info["algorithmic_bytes"] / (k_us * 1e-6) / 1e9,
                         "tiles": {k: job.info[k] for k in ("tiles_fill", "tiles_copy", "tiles_sample", "tiles_general")},
                         "rows_walked": "flat form: the strip's bytes as rows of 32 KiB (dense rows on both sides; DESIGN.md section 3)" if flat else "the canvas's rows"}
        del sets, outs, job
        torch.cuda.empty_cache()
    head = results["uniform_vertical"]
    extra = {k: {"MPs": round(v["MPs"], 1), "kernel_us": round(v["kernel_us"], 2), "GBs": round(v["GBs"], 1),
                 "frac": round(v["GBs"] / HBM_PEAK_GBS, 4), "canvas": v["canvas"], "tiles": v["tiles"], "rows_walked": v["rows_walked"]} for k, v in results.items()}
    yard = cpu = None
    if not args.kernels_only:
        yard = d2d_yardstick(head["algorithmic_bytes"] // 2, dev, torch)
        try:
            extra["regions"] = single_gpu_regions(st, ist, dev, torch, head)
        except Exception as ex:      # informational legs never take the headline down
            extra["regions"] = {"error": repr(ex)}
        try:
            extra["canvas_rows_padded_to_4KiB"] = canvas_pitch_leg(st, dev, torch)
        except Exception as ex:
            extra["canvas_rows_padded_to_4KiB"] = {"error": repr(ex)}
        try:
            extra["config5_single_gpu"] = config5_single_leg(st, dev, torch)
        except Exception as ex:
            extra["config5_single_gpu"] = {"error": repr(ex)}
        try:
            extra["end_to_end_host_path"] = host_path_leg(ist, np)
        except Exception as ex:
            extra["end_to_end_host_path"] = {"error": repr(ex)}
        try:
            extra["file_pipeline"] = file_pipeline_leg(ist)
        except Exception as ex:
            extra["file_pipeline"] = {"error": repr(ex)}
        cpu = None if args.no_cpu else cpu_baseline()
    # HBM traffic per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE
    # doubled per MI355X_MICROARCH.md), collected by profiles/summarize.py.  Only quoted when the measurement belongs to
    # THIS kernel + tiling (the JSON carries a hash of the kernel sources); otherwise null.
    traffic, traffic_src = None, "no PMC measurement of this kernel revision is committed (profiles/*_pmc.json)"
    sha = kernel_source_sha()
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc.json")), reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pm = json.load(f)
            if pm.get("kernel_source_sha") != sha:
                continue
            row = pm["uniform_vertical (BASELINE configs[1])"]
            traffic = int(row["FETCH_SIZE_bytes"] + row["WRITE_SIZE_bytes"])
            traffic_src = "profiles/%s (rocprofv3 --pmc passes of this bench at kernel revision %s; bytes per launch)" % (name, sha)
            break
        except Exception:
            continue
    line = {
        "metric": "stitched megapixels/sec (9x12 MP vertical)", "value": round(head["MPs"], 1), "unit": "MP/s",
        "n_gpus": 1, "steps": head["steps"], "warmup": args.warmup, "ms_per_step": round(head["ms_per_step"], 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 9 x 4032x3024 RGBA8 vertical stitch, bilinear resample to common width, "
                               "caps lifted -> 4032x27216 (109.73 MP); inputs and output resident in HBM, one fused launch per stitch",
                   "buffer_sets_rotated": nsets, "timed_region": "kernel launches only (no H2D/D2H, no PNG)",
                   "untimed_before_the_warmup": "%.0f ms of back-to-back launches per configuration, so that the timed steps run at the clock the chip holds under load" % PREROLL_MS},
        "roofline": {"bound": "hbm", "achieved": round(head["GBs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(head["GBs"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "ist_stitch_kernel", "kernel_us": round(head["kernel_us"], 2),
                     "algorithmic_bytes_per_launch": head["algorithmic_bytes"], "kernel_source_sha": sha},
        "cpu_baseline": cpu,
        "d2d_copy_yardstick": yard,
        "extra": extra,
    }
    args.out.emit(json.dumps(line))


def device_noise(torch, h, w, dev):
    """an h x w RGBA8 image of uniform random bytes with alpha 255, synthesised on the device (the 64 x 48 MP inputs of
    configs[4] are 12.3 GB: seconds on the GPU, minutes in numpy); one spare row behind it (dist.alloc_rows)"""
    t = torch.empty((h + 1, w, 4), dtype=torch.uint8, device=dev)[:h]
    t.random_(0, 256)
    t[..., 3] = 255
    return t


def config5_single_leg(st, dev, torch, steps=12):
    """BASELINE configs[4] on ONE GPU at its own size: 64 x 8000x6000 -> 8000x384000 in ONE launch (24.6 GB moved), inputs and
    output resident, event-timed like the headline.  No buffer rotation: one set is 96 x the Infinity Cache."""
    need = 2 * sum(w * h * 4 for w, h in CONFIG5)
    free, _ = torch.cuda.mem_get_info(dev)
    if free < need * 1.15:
        return {"skipped": "needs %.1f GB of HBM, %.1f GB free" % (need / 1e9, free / 1e9)}
    imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in CONFIG5]
    p, job = st.compile(imgs, "vertical", {"filter": "bilinear"})
    srcs = [device_noise(torch, h, w, dev) for (w, h) in CONFIG5]
    out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev)
    for _ in range(15):                              # ~60 ms of launches: the chip's sustained clock
        job.launch(srcs, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        job.launch(srcs, out)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    k_us = e0.elapsed_time(e1) * 1e3 / steps
    # identity property (every draw is 1:1): the strip is the images one under the other - checked on the device, three of them
    ok = all(bool(torch.equal(out[6000 * k:6000 * (k + 1)], srcs[k])) for k in (0, 31, 63))
    B = int(job.info["algorithmic_bytes"])
    mp = p.canvas_w * p.canvas_h / 1e6
    res = {"canvas": [p.canvas_w, p.canvas_h], "out_MP": round(mp, 1), "steps": steps, "kernel_us": round(k_us, 1), "ms_per_step": round(wall * 1e3, 4),
           "MPs": round(mp / wall, 1), "algorithmic_bytes": B, "GBs": round(B / (k_us * 1e-6) / 1e9, 1), "frac": round(B / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
           "tiles": {k: job.info[k] for k in ("tiles_fill", "tiles_copy", "tiles_sample", "tiles_general")}, "strip_is_the_images_in_order": ok,
           "what": "BASELINE configs[4]: 64 x 8000x6000 RGBA8 vertical stitch, bilinear, caps lifted -> 8000x384000 (3072 MP), ONE launch on one GPU; "
                   "inputs synthesised on the device, inputs and output resident (24.6 GB moved per launch)"}
    del srcs, out, job
    torch.cuda.empty_cache()
    return res


def d2d_yardstick(nbytes, dev, torch):
    """the runtime's own device-to-device copy (torch copy_) of the same number of bytes, same box, same process"""
    try:
        a = [torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 256) for _ in range(2)]
        b = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        preroll(lambda: b[0].copy_(a[0]), torch)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for i in range(reps):
            b[i % 2].copy_(a[i % 2])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        return {"us": round(us, 2), "GBs": round(2 * nbytes / (us * 1e-6) / 1e9, 1), "frac": round(2 * nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "what": "torch Tensor.copy_ device-to-device, %d bytes read + as many written" % nbytes}
    except Exception as ex:       # informational only
        return {"error": repr(ex)}


def single_gpu_regions(st, ist, dev, torch, head, reps=8):
    """the three timed regions of the N > 1 lines, on ONE GPU (the N = 1 point of their scaling curves)"""
    imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in UNIFORM]
    p, job = st.compile(imgs, "vertical", {"filter": "bilinear"})
    mp = p.canvas_w * p.canvas_h / 1e6
    out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev)
    host = [torch.from_numpy(synth_np(k, w, h)).pin_memory() for k, (w, h) in enumerate(UNIFORM)]
    srcs = [torch.empty((h + 1, w, 4), dtype=torch.uint8, device=dev)[:h] for (w, h) in UNIFORM]

    def timed(fn):
        preroll(fn, torch)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def from_host():
        for d, s in zip(srcs, host):
            d.copy_(s, non_blocking=True)
        job.launch(srcs, out)
    t_host = timed(from_host)
    hout = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8).pin_memory()

    def host_sink():
        from_host()
        hout.copy_(out, non_blocking=True)
    t_sink = timed(host_sink)
    del hout
    blobs = [photo_jpeg(k, w, h) for k, (w, h) in enumerate(UNIFORM)]

    def from_jpeg():
        ist.decode_files_device(blobs, out=srcs)
        job.launch(srcs, out)
    t_jpeg = timed(from_jpeg)
    return {"resident": {"ms_per_step": round(head["ms_per_step"], 4), "MPs": round(head["MPs"], 1), "what": "= value: inputs resident in HBM"},
            "from_pinned_host": {"ms_per_step": round(t_host * 1e3, 3), "MPs": round(mp / t_host, 1), "h2d_bytes": int(sum(w * h * 4 for w, h in UNIFORM)),
                                 "what": "9 x 48.8 MB from pinned host memory over this GPU's PCIe link, then the launch"},
            "host_in_host_out": {"ms_per_step": round(t_sink * 1e3, 3), "MPs": round(mp / t_sink, 1),
                                 "what": "pinned host in -> H2D, launch, D2H of the canvas into pinned host memory, all on ONE stream: the two directions do not overlap (the N = 1 point of host_in_host_out/*, whose ranks do the same; the library's own host entry point overlaps them band by band: extra.end_to_end_host_path)"},
            "from_jpeg": {"ms_per_step": round(t_jpeg * 1e3, 3), "MPs": round(mp / t_jpeg, 1), "jpeg_bytes": int(sum(len(b) for b in blobs)),
                          "what": "nine 12 MP photo-like JPEGs (q90, 4:2:0) decoded on the GPU (Huffman + IDCT + colour), then the launch"}}


def host_path_leg(ist, np, reps=4):
    """BASELINE.md section 4 / SURVEY 8d 'end to end, reported separately and labelled': numpy in -> plan -> staged H2D ->
    launch -> D2H into a pooled pinned block, through ist_stitch_rgba8 (what the N-API addon binds)"""
    px = [synth_np(k, 4032, 3024) for k in range(9)]
    imgs = [{"width": 4032, "height": 3024, "data": a, "opaque": True} for a in px]
    ist.stitch(imgs, "vertical", {"filter": "bilinear"})            # warm-up: scratch, staging ring, result pool
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = ist.stitch(imgs, "vertical", {"filter": "bilinear"})
        ts.append(time.perf_counter() - t0)
        del r
    t = sorted(ts)[len(ts) // 2]
    return {"ms_per_stitch": round(t * 1e3, 2), "MPs": round(109.734912 / t, 1), "pcie_payload_GBs": round(2 * 438.939648e6 / t / 1e9, 2),
            "what": "PCIe-inclusive: pageable numpy in -> plan -> row bands of ~40 MB: the source rows a band samples are packed into 32 MiB pinned pieces and sent up, the band is launched, its rows come down into a pooled pinned block while the next band's rows go up (never `value`)"}


def file_pipeline_leg(ist, reps=5):
    """SURVEY 8f ranks 2-3 measured: nine 12 MP JPEGs -> one PNG through ist_stitch_files_png.  Two runs of the same call:
    PIPELINED (production: every image decodes on its own thread + stream, band k is rendered and its PNG slabs cross PCIe
    while later images are still in the Huffman decoder) = ms_end_to_end; and PHASE-TIMED (ist_ctx_set_timing: the same steps
    with a barrier and a stream sync between them) = stages_ms, each stage with the bound it is priced against."""
    import tempfile
    tmp = tempfile.mkdtemp()
    paths = []
    for k, (w, h) in enumerate(UNIFORM):
        p = os.path.join(tmp, "in%d.jpg" % k)
        with open(p, "wb") as f:
            f.write(photo_jpeg(k, w, h))
        paths.append(p)
    in_bytes = sum(os.path.getsize(p) for p in paths)
    for _ in range(2):
        ist.stitch_files(paths, "vertical", copy=False)             # warm-up: arena, streams, scratch, result pool
    ts, png_len = [], 0
    for _ in range(reps):
        t0 = time.perf_counter()
        res = ist.stitch_files(paths, "vertical", copy=False)
        ts.append(time.perf_counter() - t0)
        png_len = len(res["png"])
        del res
    best = min(ts)
    ist.set_phase_timing(True)
    try:
        tbest, phases = None, None
        for _ in range(3):
            t0 = time.perf_counter()
            res = ist.stitch_files(paths, "vertical", copy=False)
            dt = time.perf_counter() - t0
            if tbest is None or dt < tbest:
                tbest, phases = dt, ist.last_phase_times()
            del res
    finally:
        ist.set_phase_timing(False)
    canvas = 4032 * 27216 * 4
    px = 4032 * 27216
    # what each stage is priced against (the line, not prose, says which stage is furthest from its ceiling):
    #   entropy_gpu   instruction issue: 3 decodes of the scan (guess, re-decode from the neighbour's exit state, writing pass) x ~22 M
    #                 symbols x ~200 instructions per symbol step, 64 lanes per wave, 1024 SIMDs issuing one wave-instruction per
    #                 4 cycles at 2.4 GHz
    #   reconstruct   HBM: coefficients in (3 B/px at 4:2:0) + sample planes out and in (1.5 + 1.5 B/px) + RGBA out (4 B/px)
    #   stitch        HBM: 8 B per output pixel (the headline kernel)
    #   png           PCIe payload: the file's bytes over this GPU's link (57 GB/s: the rate one pinned DMA sustains on these boxes),
    #                 which the encoder must hide behind
    symbols = in_bytes * 8 / 6.0
    issue_rate = 1024 * 2.4e9 / 4 * 64 / 200.0
    bounds = {"entropy_gpu": ("instruction issue (3 decodes x symbols x ~200 instr)", 3 * symbols / issue_rate * 1e3),
              "reconstruct": ("hbm (10 B per pixel)", 10.0 * px / 8e12 * 1e3),
              "stitch": ("hbm (8 B per pixel)", 8.0 * px / 8e12 * 1e3),
              "png": ("pcie payload (file bytes at 57 GB/s)", png_len / 57e9 * 1e3)}
    roof = {k: {"bound": b, "bound_ms": round(ms, 3), "achieved_ms": round(phases[k], 3), "frac": round(ms / phases[k], 3) if phases[k] > 0 else None}
            for k, (b, ms) in bounds.items()}
    return {"ms_end_to_end": round(best * 1e3, 2), "ms_end_to_end_all": [round(t * 1e3, 2) for t in ts],
            "ms_phase_timed_run": round(tbest * 1e3, 2), "jpeg_bytes_in": in_bytes, "png_bytes_out": png_len,
            "stages_ms": {k: round(v, 3) for k, v in phases.items()}, "stage_rooflines": roof,
            "what": "nine photo-like 12 MP JPEGs -> 4032x27216 PNG (level 1); includes reading the files.  ms_end_to_end = the pipelined call "
                    "(images decode on their own threads + streams, bands are rendered and exported as their image arrives); stages_ms = the same call "
                    "with a barrier + stream sync after every stage (ist_ctx_set_timing), whose sum is therefore larger than the pipelined time; "
                    "`png` holds the D2H (slabs cross PCIe while later slabs compress)"}


# ---------------------------------------------------------------------------------------------------- N > 1
SCALING_NOTE = ("value = resident/image: the contract's region (inputs resident in HBM) with BASELINE's split.  That region CANNOT scale on this "
                "hardware: one GPU moves the whole job in ~0.14 ms, the gather alone needs 48.8 MB over one xGMI link per sender (>= 0.3 ms) - "
                "extra.scaling shows its speed-up below 1 by construction.  The >= 6x target is claimed on host_in_host_out/band (each GPU's own "
                "PCIe link in both directions, no gather) and, for 64 x 48 MP, on extra.config5.scaling[host_in_host_out/image]; every entry of "
                "extra.scaling pairs the N-GPU time with the ONE-GPU form of the same region timed by rank 0 alone in this process.")


def scaling_table(regions_n, regions_1):
    """{region/split: {ms_1gpu, ms_Ngpu, speedup}}: the N-rank time of a region against the one-GPU form of the same region
    (key without the split) measured by rank 0 alone in the same process"""
    out = {}
    for name, r in regions_n.items():
        one = regions_1.get(name.split("/")[0])
        if not isinstance(r, dict) or "ms_per_step" not in r or not isinstance(one, dict) or "ms_per_step" not in one:
            continue
        out[name] = {"ms_1gpu": one["ms_per_step"], "ms_Ngpu": r["ms_per_step"], "speedup": round(one["ms_per_step"] / r["ms_per_step"], 3) if r["ms_per_step"] > 0 else None}
    return out


class Solo:
    """rank 0 works ALONE while the other ranks block on the rendezvous store - on the host, with no kernel on their GPUs (a
    collective barrier would park a spinning kernel on every waiting GPU)"""

    def __init__(self, dist, rank):
        from datetime import timedelta
        self.dist, self.rank, self.n, self.timeout = dist, rank, 0, timedelta(seconds=3000)
        try:
            self.store = dist.distributed_c10d._get_default_store()
        except Exception:
            self.store = None

    def __call__(self, fn):
        self.n += 1
        key = "ist_bench_solo_%d" % self.n
        self.dist.barrier()
        res = None
        if self.rank == 0:
            try:
                res = fn()
            except Exception as ex:          # an informational comparator never takes the line (or the other ranks) down
                res = ({"error": repr(ex)}, 0)
            finally:
                if self.store is not None:
                    self.store.set(key, "done")
        elif self.store is not None:
            self.store.wait([key], self.timeout)
        self.dist.barrier()
        return res


def run_sharded(args):
    import torch
    import torch.distributed as dist
    import imagestitching_amd as ist
    from imagestitching_amd import dist as D
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not os.environ.get("IST_BENCH_FORCE_SHARDED"):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus %d` (it launches its own ranks) or under "
                         "torchrun with --nproc-per-node %d" % (args.gpus, world, args.gpus, args.gpus))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    local = int(os.environ.get("LOCAL_RANK", rank))
    if torch.cuda.device_count() <= local:
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d are visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # RCCL, or - SURVEY 8e "fallback: if RCCL init fails -> replicas only" - gloo for the control plane: the regions that exchange
    # nothing between GPUs (host_in_host_out: every GPU delivers its own bands) and the replicas leg still run, the gather regions
    # are reported as unavailable, and `value` is the replicas' aggregate.  (--force-gloo rehearses this path.)
    rccl, rccl_error = True, None
    try:
        if args.force_gloo:
            raise RuntimeError("forced by --force-gloo")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist.barrier()                                     # the first collective makes (and proves) the communicator
    except Exception as ex:
        rccl, rccl_error = False, repr(ex)
        sys.stderr.write("bench.py: rank %d: RCCL is not usable (%s): gloo control plane, no gather regions\n" % (rank, rccl_error))
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ctl = dev if rccl else torch.device("cpu")             # where the control tensors of the collectives live
    ranks_seen = gather_ranks(dist, rank, local, torch)
    solo = Solo(dist, rank)

    def timed(step, steps, warmup):
        """the contract's bracket: barrier + synchronize on both sides, MAX over ranks"""
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0], device=ctl, dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item()) / steps

    def timed_alone(step, steps, warmup):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def sharded_suite(sizes, direction, split, want, steps, big):
        """one workload cut one way: its regions on all ranks.  Returns (regions, gather facts, seconds of `resident`)."""
        imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in sizes]
        # SET-UP (no collective inside): a rank that cannot allocate says so, and ALL ranks leave the suite together - a rank that
        # raised alone would leave the others in the next collective until the launch times out, and the line would be lost
        err = None
        try:
            sh = D.ShardedStitch(imgs, direction, {"filter": "bilinear"}, rank, world, 0, split=split)
            mp = sh.plan.canvas_w * sh.plan.canvas_h / 1e6
            be = D.HipBackend(sh, local)
            need = sh.rows_needed()
            # this rank's holdings: only the source rows its parts sample (+ one spare row), in HBM and in pinned host memory
            dsrc, hsrc = [None] * len(sizes), {}
            for i, (a, b) in need.items():
                if big:                                   # synthesised on the device (12.3 GB in numpy would take minutes)
                    t = device_noise(torch, b - a, sizes[i][0], dev)
                    hsrc[i] = torch.empty((b - a, sizes[i][0], 4), dtype=torch.uint8).pin_memory()
                    hsrc[i].copy_(t)
                else:
                    t = D.alloc_rows(torch, b - a, sizes[i][0], dev)
                    hsrc[i] = torch.from_numpy(synth_np(i, *sizes[i])[a:b]).pin_memory()
                    t.copy_(hsrc[i])
                dsrc[i] = D.SourceRows(t, a)
            canvas = be.new_canvas() if rank == 0 else None
        except Exception as ex:
            err = repr(ex)
        bad = torch.tensor([1 if err else 0], device=ctl, dtype=torch.int32)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            torch.cuda.empty_cache()
            return {"set-up": {"error": err or "another rank could not set this suite up"}}, {"split": split}, None
        regions = {}

        def upload():
            for i in need:
                dsrc[i].tensor.copy_(hsrc[i], non_blocking=True)

        def resident():
            D.run_step(sh, be, dsrc, canvas, dist)
        t_res = None
        if rccl or not sh.remote:
            t_res = timed(resident, steps, max(args.warmup, 3 if big else 20))     # (>= 20 untimed steps: every rank's chip reaches its sustained clock)
            regions["resident"] = {"ms_per_step": round(t_res * 1e3, 4), "MPs": round(mp / t_res, 1)}
        else:
            regions["resident"] = {"unavailable": "the gather needs RCCL: " + str(rccl_error)}
            want = want - {"from_pinned_host", "from_jpeg"}                       # (they end in the same gather)
        few = max(3, steps // 10)
        if "from_pinned_host" in want:
            t = timed(lambda: (upload(), resident()), few, 2)
            regions["from_pinned_host"] = {"ms_per_step": round(t * 1e3, 4), "MPs": round(mp / t, 1), "h2d_bytes_this_rank": int(sum(v.numel() for v in hsrc.values()))}
        # host in, host out with a HOST SINK: no gather - every rank copies its finished bands into pinned host memory over its
        # own PCIe link (full-width bands are contiguous byte ranges of the host canvas)
        if "host_in_host_out" in want and sh.root_rows() is not None:
            hbands = {p.index: torch.empty(p.shape, dtype=torch.uint8).pin_memory() for p in sh.mine if sh.slot != 0}
            hcanvas = D.HostRows(torch, sh) if rank == 0 else None
            t = timed(lambda: (upload(), D.run_step_host_sink(sh, be, dsrc, canvas, hbands, hcanvas)), few, 2)
            regions["host_in_host_out"] = {"ms_per_step": round(t * 1e3, 4), "MPs": round(mp / t, 1),
                                           "what": "pinned host in -> each rank's H2D, band launches, D2H of its bands into pinned host memory; no gather"}
            del hbands, hcanvas
        if "from_jpeg" in want:                       # JPEG inputs are whole files: by image only
            blobs = [photo_jpeg(i, *sizes[i]) for i in sorted(need)]
            outs = [dsrc[i].tensor for i in sorted(need)]

            def from_jpeg():
                if blobs:
                    ist.decode_files_device(blobs, device=local, out=outs)
                resident()
            t = timed(from_jpeg, max(3, steps // 20), 1)
            regions["from_jpeg"] = {"ms_per_step": round(t * 1e3, 4), "MPs": round(mp / t, 1)}
        per_sender = {}
        for p in sh.remote:
            per_sender[sh.rank_of(p)] = per_sender.get(sh.rank_of(p), 0) + p.nbytes
        px = [0] * world
        for q in sh.parts:
            px[q.slot] += (q.X1 - q.X0) * (q.Y1 - q.Y0)
        gather = {"split": sh.split, "bytes_into_gpu0_per_step": int(sum(per_sender.values())), "sending_gpus": len(per_sender),
                  "busiest_link_bytes": int(max(per_sender.values()) if per_sender else 0),
                  "bands_in_place": sum(1 for p in sh.remote if p.in_place), "bands_staged": sum(1 for p in sh.remote if not p.in_place),
                  "host_sink_available": sh.root_rows() is not None,
                  "output_pixels_per_rank_max_over_mean": round(max(px) / (sum(px) / world), 3) if sum(px) else None}
        del be, sh, dsrc, hsrc, canvas
        torch.cuda.empty_cache()
        return regions, gather, t_res

    def single_suite(sizes, direction, want, steps, big):
        """rank 0 ALONE: the one-GPU form of the same regions (one fused launch; all inputs over this GPU's one PCIe link)"""
        imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in sizes]
        st = ist.Stitcher(local)
        p, job = st.compile(imgs, direction, {"filter": "bilinear"})
        mp = p.canvas_w * p.canvas_h / 1e6
        srcs = [device_noise(torch, h, w, dev) for (w, h) in sizes]
        out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev)
        regions = {}
        t = timed_alone(lambda: job.launch(srcs, out), steps, 3 if big else 400)
        regions["resident"] = {"ms_per_step": round(t * 1e3, 4), "MPs": round(mp / t, 1)}
        few = max(3, steps // 10)
        host = None
        if "from_pinned_host" in want or "host_in_host_out" in want:
            host = [torch.empty((h, w, 4), dtype=torch.uint8).pin_memory() for (w, h) in sizes]
            for hs, d in zip(host, srcs):
                hs.copy_(d)

            def from_host():
                for d, hs in zip(srcs, host):
                    d.copy_(hs, non_blocking=True)
                job.launch(srcs, out)
            if "from_pinned_host" in want:
                t = timed_alone(from_host, few, 2)
                regions["from_pinned_host"] = {"ms_per_step": round(t * 1e3, 4), "MPs": round(mp / t, 1)}
            if "host_in_host_out" in want:
                hout = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8).pin_memory()
                t = timed_alone(lambda: (from_host(), hout.copy_(out, non_blocking=True)), few, 2)
                regions["host_in_host_out"] = {"ms_per_step": round(t * 1e3, 4), "MPs": round(mp / t, 1)}
                del hout
        if "from_jpeg" in want:
            blobs = [photo_jpeg(k, w, h) for k, (w, h) in enumerate(sizes)]
            t = timed_alone(lambda: (ist.decode_files_device(blobs, device=local, out=srcs), job.launch(srcs, out)), max(3, steps // 20), 1)
            regions["from_jpeg"] = {"ms_per_step": round(t * 1e3, 4), "MPs": round(mp / t, 1)}
        B = int(job.info["algorithmic_bytes"])
        del srcs, out, host, job, st
        torch.cuda.empty_cache()
        return regions, B

    # ---- BASELINE configs[3]: 9 x 12 MP vertical, by image (= value) and by band; configs[2]'s horizontal strip by rows
    regions, gather = {}, {}
    value_sec = None
    all_regions = {"from_pinned_host", "host_in_host_out", "from_jpeg"}
    for split in ("image", "band"):
        r, g, t_res = sharded_suite(UNIFORM, "vertical", split, all_regions if split == "image" else all_regions - {"from_jpeg"}, args.steps, False)
        if split == "image":
            if t_res is None and rccl:
                raise SystemExit("bench.py: the headline suite could not be set up: %s" % r)
            value_sec = t_res
        for k, v in r.items():
            regions["%s/%s" % (k, split)] = v
        gather[split] = g
    r, g, _ = sharded_suite(UNIFORM, "horizontal", "auto", {"host_in_host_out"}, max(10, args.steps // 4), False)
    for k, v in r.items():
        regions["%s/rows_horizontal" % k] = v
    gather["rows_horizontal"] = g
    one = solo(lambda: single_suite(UNIFORM, "vertical", all_regions, args.steps, False))
    one_h = solo(lambda: single_suite(UNIFORM, "horizontal", {"host_in_host_out"}, max(10, args.steps // 4), False))
    scaling = None
    if rank == 0:
        scaling = scaling_table({k: v for k, v in regions.items() if not k.endswith("/rows_horizontal")}, one[0])
        scaling.update(scaling_table({k: v for k, v in regions.items() if k.endswith("/rows_horizontal")}, one_h[0]))

    # ---- BASELINE configs[4]: 64 x 48 MP vertical, 8 images per rank at 8 GPUs (by image), resident + host in / host out
    config5 = None
    if not args.no_config5:
        per_rank = 2 * max(sum(w * h * 4 for k, (w, h) in enumerate(CONFIG5) if k % world == s) for s in range(world))
        need0 = per_rank + sum(w * h * 4 for w, h in CONFIG5)                     # the root also holds the canvas
        free = torch.tensor([torch.cuda.mem_get_info(dev)[0]], device=ctl, dtype=torch.float64)
        dist.all_reduce(free, op=dist.ReduceOp.MIN)
        if float(free.item()) < need0 * 1.2:
            config5 = {"skipped": "needs %.1f GB of HBM on the root, %.1f GB free on the fullest GPU" % (need0 / 1e9, float(free.item()) / 1e9)}
        else:
            r5, g5, _ = sharded_suite(CONFIG5, "vertical", "image", {"host_in_host_out"}, 10, True)
            need1 = 2 * sum(w * h * 4 for w, h in CONFIG5)
            fits = torch.tensor([1 if torch.cuda.mem_get_info(dev)[0] >= need1 * 1.15 else 0], device=ctl, dtype=torch.int32)
            dist.broadcast(fits, 0)                                               # (rank 0 decides for everybody: solo() is collective)
            one5 = solo(lambda: single_suite(CONFIG5, "vertical", {"host_in_host_out"} if not args.no_config5_host else set(), 10, True)) if int(fits.item()) else None
            if rank == 0:
                config5 = {"regions": {"%s/image" % k: v for k, v in r5.items()}, "gather": g5,
                           "one_gpu": one5[0] if one5 else {"skipped": "not enough free HBM on rank 0 for the one-GPU form"},
                           "scaling": scaling_table({"%s/image" % k: v for k, v in r5.items()}, one5[0]) if one5 else {},
                           "what": "BASELINE configs[4]: 64 x 8000x6000 vertical -> 8000x384000 (3072 MP), image i on GPU i mod %d (%d per GPU), inputs synthesised on "
                                   "the devices; resident = bands gathered into GPU 0's canvas (%.2f GB over xGMI per step), host_in_host_out = every GPU uploads its "
                                   "images and delivers its bands over its own PCIe link, no gather" % (world, (64 + world - 1) // world, g5.get("bytes_into_gpu0_per_step", 0) / 1e9)}

    # informational: N independent replicas (every GPU stitches a whole 9 x 12 MP job; no exchange) = the layout a
    # stitching service would use when jobs are independent
    imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in UNIFORM]
    mp = 4032 * 27216 / 1e6
    st = ist.Stitcher(local)
    p_full, job_full = st.compile(imgs, "vertical", {"filter": "bilinear"})
    full_src = [synth(k, w, h, dev) for k, (w, h) in enumerate(UNIFORM)]
    full_out = torch.empty((p_full.canvas_h, p_full.canvas_w, 4), dtype=torch.uint8, device=dev)
    t_rep = timed(lambda: job_full.launch(full_src, full_out), args.steps, 20)
    B_full = int(job_full.info["algorithmic_bytes"])
    if rank == 0:
        if value_sec is None:                              # RCCL unavailable: replicas only
            headline = {"value": round(world * mp / t_rep, 1), "ms_per_step": round(t_rep * 1e3, 5), "scaling": "weak",
                        "workload": "REPLICAS ONLY - RCCL is not usable on this node (%s): every GPU stitches its own whole 9 x 4032x3024 job; the gather regions "
                                    "are unavailable, host_in_host_out (no exchange) is reported in extra.regions" % rccl_error,
                        "timed_region": "one fused launch per GPU per step, inputs resident; aggregate over the GPUs"}
        else:
            headline = {"value": round(mp / value_sec, 1), "ms_per_step": round(value_sec * 1e3, 5), "scaling": "strong",
                        "workload": "BASELINE configs[3]: 9 x 4032x3024 vertical stitch, image i on GPU i mod %d, bands gathered to "
                                    "GPU 0 with one grouped RCCL send/recv batch (%d bands received in place)" % (world, gather["image"]["bands_in_place"]),
                        "timed_region": "per-rank band launches + gather + root launch; inputs resident in each owner's HBM.  " + SCALING_NOTE}
        line = {
            "metric": "stitched megapixels/sec (9x12 MP vertical)", "value": headline["value"], "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": headline["ms_per_step"],
            "higher_is_better": True, "scaling": headline["scaling"], "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": headline["workload"], "timed_region": headline["timed_region"]},
            # the dominant kernel is the one of the N = 1 line; here it is timed on the whole 9 x 12 MP job that every rank
            # launches in the replicas leg (wall time per launch of back-to-back launches, MAX over ranks; no PMC at N > 1)
            "roofline": {"bound": "hbm", "achieved": round(B_full / t_rep / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(B_full / t_rep / 8e12, 4), "traffic": None, "kernel": "ist_stitch_kernel",
                         "kernel_us": round(t_rep * 1e6, 2), "algorithmic_bytes_per_launch": B_full,
                         "what": "per GPU, from the replicas leg (each rank launches the whole job); slowest rank"},
            "cpu_baseline": None,
            "extra": {"ranks_seen": ranks_seen, "world_size": dist.get_world_size(), "collectives": "rccl" if rccl else "gloo (RCCL unavailable: %s)" % rccl_error,
                      "regions": regions, "gather": gather,
                      "one_gpu_same_lease": {"vertical": one[0], "horizontal": one_h[0],
                                             "what": "the ONE-GPU form of every region (one fused launch; everything over GPU 0's one PCIe link), timed by rank 0 "
                                                     "alone in this process while the other ranks block on the rendezvous store"},
                      "scaling": scaling, "config5": config5,
                      "regions_note": "value = resident/image (the contract's region and BASELINE's split).  from_pinned_host and from_jpeg add each rank's own "
                                      "ingest (its PCIe link / its decoder) in front of the same step; /band deals equal output rows to every rank instead of whole images; "
                                      "/rows_horizontal is BASELINE configs[2]'s strip (36288x3024) cut by rows: every rank a full-width band of all nine images.  "
                                      "xGMI is point to point: a step is bounded below by busiest_link_bytes / one link's rate.",
                      "replicas_no_exchange": {"MPs": round(world * mp / t_rep, 1), "scaling": "weak", "note": "every GPU stitches its own whole 9x12 MP job"}},
        }
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        del job_full, full_src, full_out, st
        torch.cuda.empty_cache()
        if not os.environ.get("IST_BENCH_CHILD"):        # under torchrun rank 0 owns the line; under launch_ranks the parent finishes it
            finish_line(line, args, world)
        args.out.emit(json.dumps(line))


# ---------------------------------------------------------------------------------------------------- rank launch + proof
def gather_ranks(dist, rank, local, torch=None):
    """what every rank really is: all-gathered, so rank 0's line shows N distinct processes on N distinct devices"""
    import socket
    me = {"rank": rank, "local_rank": local, "pid": os.getpid(), "host": socket.gethostname()}
    if torch is not None and torch.cuda.is_available():
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        me.update(device=torch.cuda.current_device(), name=pr.name,
                  pci="%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
                  uuid=str(getattr(pr, "uuid", "")))
    seen = [None] * dist.get_world_size()
    dist.all_gather_object(seen, me)
    return seen


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def child_argv(args, extra=()):
    a = ["--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    for flag, on in (("--no-cpu", args.no_cpu), ("--quick", args.quick), ("--kernels-only", args.kernels_only), ("--dry-launch", args.dry_launch),
                     ("--no-config5", args.no_config5), ("--no-config5-host", args.no_config5_host), ("--force-gloo", args.force_gloo)):
        if on:
            a.append(flag)
    return a + list(extra)


def last_json_line(text):
    for ln in reversed(text.splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def launch_ranks(args):
    """`python bench.py --gpus N` invoked plainly: THIS process has not imported torch and has not touched HIP; it starts N
    fresh rank processes (never an exec of a process that initialised the GPU), waits for them, and prints rank 0's line.
    A rank that fails takes the others down (they would wait for it in the collective for ever); exit status = the first
    failure's."""
    import subprocess
    n = args.gpus
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), IST_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + child_argv(args), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    import threading
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + args.launch_timeout
    status, why = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            status, why = bad[0][1], "rank %d exited with status %d" % bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            status, why = 124, "the ranks did not finish within --launch-timeout %d s" % args.launch_timeout
            break
        time.sleep(0.05)
    if why:
        for p in procs:                       # exactly the processes started above, by handle
            if p.poll() is None:
                p.terminate()
        t_kill = time.time() + 10
        for p in procs:
            try:
                p.wait(max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        sys.stderr.write("bench.py: %s\n" % why)
        raise SystemExit(status if status > 0 else 1)
    reader.join(10)
    line = last_json_line(got[0].decode("utf-8", "replace") if got else "")
    if line is None:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        raise SystemExit(1)
    line.setdefault("extra", {})["launcher"] = "bench.py started %d rank processes itself (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*), one per GPU" % n
    if not args.dry_launch:
        finish_line(line, args, n)
    args.out.emit(json.dumps(line))


def finish_line(line, args, n):
    """what the N > 1 line carries besides the ranks' own measurements: the CPU baseline (same leg as N = 1, run once the
    ranks are done so that neither disturbs the other) and the SAME layout driven from ONE process through the C-ABI device
    group (ist_group_*: what the N-API host binds) in a child process with its own time limit."""
    import subprocess
    if not args.no_cpu and not args.kernels_only:
        try:
            line["cpu_baseline"] = cpu_baseline()
        except Exception as ex:
            line["cpu_baseline"] = {"error": repr(ex)}
    if args.kernels_only:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                             "IST_BENCH_CHILD", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    # two children, each with its own time limit: the host-sink legs exchange nothing between GPUs; the device-sink legs gather
    # over xGMI through RCCL - a problem in one must not cost the other's numbers
    merged = {"regions": {}}
    for part in ("host", "resident"):
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--group-leg", str(n), "--group-part", part, "--steps", str(args.steps)] +
                               (["--no-config5"] if args.no_config5 else []), env=env,
                               stdout=subprocess.PIPE, stderr=sys.stderr, timeout=args.group_timeout)
            leg = last_json_line(r.stdout.decode("utf-8", "replace"))
            if leg is None:
                merged["regions"][part + "/error"] = "exit status %d, no JSON" % r.returncode
            else:
                merged["regions"].update(leg.pop("regions", {}))
                c5 = leg.pop("config5", None)
                if c5 is not None:
                    merged.setdefault("config5", {}).update(c5)
                merged.update(leg)
        except subprocess.TimeoutExpired:
            merged["regions"][part + "/error"] = "timed out after %d s" % args.group_timeout
        except Exception as ex:
            merged["regions"][part + "/error"] = repr(ex)
    line.setdefault("extra", {})["single_process_group"] = merged


# ---------------------------------------------------------------------------------------------------- C-ABI device group leg
def group_leg(n, steps, out, part="all", config5=True):
    """ONE process drives GPUs 0..n-1 through the C-ABI device group (ist_group_*; the N-API host's `devices` option).
    Regions, per split:  resident (device pointers per part, canvas on the root, RCCL gather) and host_in_host_out
    (ist_stitch_rgba8_multi: every device uploads only its rows over its own PCIe link and DMAs its finished band straight
    into the pinned result - no gather, no root readback)."""
    import numpy as np
    import torch
    import imagestitching_amd as ist
    have = torch.cuda.device_count()
    if have < n:
        out.emit(json.dumps({"error": "%d devices asked for, %d visible" % (n, have)}))
        return
    devices = list(range(n))
    mp = 4032 * 27216 / 1e6
    imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in UNIFORM]
    res = {"devices": devices, "regions": {}}
    reps = max(5, min(50, steps // 4))
    try:                                       # (the device-sink legs need RCCL between distinct GPUs; the host-sink legs below do not)
        if part == "host":
            raise StopIteration
        g = ist.StitchGroup(devices)
        for split in ("image", "band"):
            job = g.compile(imgs, "vertical", {"filter": "bilinear", "split": split})
            srcs = []
            for p in job.parts:
                a, b = p["rows"]
                w = UNIFORM[p["image"]][0]
                t = torch.empty((b - a + 1, w, 4), dtype=torch.uint8, device="cuda:%d" % p["device"])[:b - a]
                t.random_(0, 256)
                srcs.append((t, a))
            canvas = torch.empty((job.plan.canvas_h, job.plan.canvas_w, 4), dtype=torch.uint8, device="cuda:%d" % devices[0])
            for _ in range(20):
                job.launch(srcs, canvas)
            g.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                job.launch(srcs, canvas)
            g.sync()
            dt = (time.perf_counter() - t0) / reps
            res["regions"]["resident/" + split] = {"ms_per_step": round(dt * 1e3, 4), "MPs": round(mp / dt, 1), "parts": len(job.parts)}
            job.close()
            del srcs, canvas
        g.close()
    except StopIteration:
        pass
    except Exception as ex:
        res["regions"]["resident/error"] = repr(ex)
    if part == "resident":
        out.emit(json.dumps(res))
        return
    px = [synth_np(k, w, h) for k, (w, h) in enumerate(UNIFORM)]
    himgs = [{"width": w, "height": h, "data": a, "opaque": True} for a, (w, h) in zip(px, UNIFORM)]
    for split in ("image", "band"):
        opts = {"filter": "bilinear", "devices": devices, "split": split}
        r = ist.stitch(himgs, "vertical", opts)                 # warm-up: contexts, staging rings, result pool, cached group job
        if split == "image":
            res["checked"] = bool(np.array_equal(r["data"][:3024], px[0]) and np.array_equal(r["data"][-3024:], px[8]))
        del r
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            r = ist.stitch(himgs, "vertical", opts)
            ts.append(time.perf_counter() - t0)
            del r
        t = sorted(ts)[len(ts) // 2]
        res["regions"]["host_in_host_out/" + split] = {"ms_per_stitch": round(t * 1e3, 3), "MPs": round(mp / t, 1),
                                                       "pcie_payload_GBs": round(2 * 438.939648e6 / t / 1e9, 2)}
    # BASELINE configs[2]'s horizontal strip: cut by rows (the default for it) every device owns a full-width band -> host sink
    try:
        from imagestitching_amd import _lib as L
        opts = {"filter": "bilinear", "devices": devices}
        before = L.lib.ist_debug_host_sink_stitches()
        r = ist.stitch(himgs, "horizontal", opts)
        res["checked_horizontal"] = bool(np.array_equal(r["data"][:, :4032], px[0]) and np.array_equal(r["data"][:, -4032:], px[8]))
        del r
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            r = ist.stitch(himgs, "horizontal", opts)
            ts.append(time.perf_counter() - t0)
            del r
        t = sorted(ts)[len(ts) // 2]
        res["regions"]["host_in_host_out/rows_horizontal"] = {"ms_per_stitch": round(t * 1e3, 3), "MPs": round(mp / t, 1), "pcie_payload_GBs": round(2 * 438.939648e6 / t / 1e9, 2),
                                                              "host_sink_used": bool(L.lib.ist_debug_host_sink_stitches() - before == 6)}
    except Exception as ex:
        res["regions"]["host_in_host_out/rows_horizontal"] = {"error": repr(ex)}
    del px, himgs
    if config5:
        try:
            res["config5"] = group_config5(n, devices, part, ist, np, torch)
        except Exception as ex:
            res["config5"] = {"error": repr(ex)}
    res["what"] = ("one process, ist_group_* over devices %s: resident = per-part device pointers, bands + one grouped RCCL batch into the root's canvas; "
                   "host_in_host_out = pageable numpy in, every device uploads only its rows and DMAs its finished band into the pinned result" % devices)
    out.emit(json.dumps(res))


def group_config5(n, devices, part, ist, np, torch):
    """BASELINE configs[4] through the single-process device group: 64 x 8000x6000 by image (8 per device at 8 devices).
    host_in_host_out: pageable numpy in (eight distinct 48 MP noise images, cycled - the bytes do not matter to the clock),
    every device uploads its images over its own link and DMAs its bands into the 12.3 GB pinned result."""
    mp = 8000 * 384000 / 1e6
    res = {}
    free = min(torch.cuda.mem_get_info(d)[0] for d in devices)
    if free < 30e9:
        return {"skipped": "%.1f GB free on the fullest device" % (free / 1e9)}
    imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in CONFIG5]
    if part != "host":
        g = ist.StitchGroup(devices)
        job = g.compile(imgs, "vertical", {"filter": "bilinear", "split": "image"})
        srcs = [device_noise(torch, 6000, 8000, "cuda:%d" % p["device"]) for p in job.parts]
        canvas = torch.empty((job.plan.canvas_h, job.plan.canvas_w, 4), dtype=torch.uint8, device="cuda:%d" % devices[0])
        for _ in range(3):
            job.launch(srcs, canvas)
        g.sync()
        t0 = time.perf_counter()
        for _ in range(8):
            job.launch(srcs, canvas)
        g.sync()
        dt = (time.perf_counter() - t0) / 8
        ok = bool(torch.equal(canvas[:6000], srcs[0].to(canvas.device)) and torch.equal(canvas[-6000:], srcs[63].to(canvas.device)))
        res["resident/image"] = {"ms_per_step": round(dt * 1e3, 3), "MPs": round(mp / dt, 1), "parts": len(job.parts), "checked": ok}
        job.close()
        del srcs, canvas
        g.close()
        torch.cuda.empty_cache()
    if part != "resident":
        base = [np.random.default_rng(5000 + k).integers(0, 256, (6000, 8000, 4), dtype=np.uint8) for k in range(8)]
        for a in base:
            a[..., 3] = 255
        himgs = [{"width": 8000, "height": 6000, "data": base[k % 8], "opaque": True} for k in range(64)]
        opts = {"filter": "bilinear", "devices": devices, "split": "image"}
        r = ist.stitch(himgs, "vertical", opts)                 # warm-up: 12.3 GB pinned result block, staging rings, arenas
        ok = bool(np.array_equal(r["data"][:6000], base[0]) and np.array_equal(r["data"][-6000:], base[63 % 8]))
        del r
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = ist.stitch(himgs, "vertical", opts)
            ts.append(time.perf_counter() - t0)
            del r
        t = sorted(ts)[len(ts) // 2]
        res["host_in_host_out/image"] = {"ms_per_stitch": round(t * 1e3, 2), "MPs": round(mp / t, 1), "pcie_payload_GBs": round(2 * 12.288e9 / t / 1e9, 2), "checked": ok}
        from imagestitching_amd import _lib as L
        L.lib.ist_pool_trim()                                   # the 12.3 GB pinned result block goes back to the system
    return res


# ---------------------------------------------------------------------------------------------------- --dry-launch (CPU)
class StubBackend:
    """--dry-launch only: the sharding + exchange of dist.run_step on CPU tensors with gloo; a band's 'render' writes the id
    of its image, so that the assembled strip can be checked without any raster (the oracle is not involved)."""

    def __init__(self, sh, torch):
        self.sh, self.torch = sh, torch
        self.staging = {p.index: torch.empty(p.shape, dtype=torch.uint8) for p in sh.remote if not p.in_place}

    def new_canvas(self):
        return self.torch.full((self.sh.plan.canvas_h, self.sh.plan.canvas_w, 4), 0xFF, dtype=self.torch.uint8)

    def render_band(self, part, srcs):
        return self.torch.full(part.shape, part.image + 1, dtype=self.torch.uint8)

    def render_root(self, srcs, canvas):
        for p in self.sh.parts:
            if p.slot == 0:
                canvas[p.Y0:p.Y1, p.X0:p.X1] = p.image + 1

    def place(self, part, canvas):
        canvas[part.Y0:part.Y1, part.X0:part.X1] = self.staging[part.index]


def run_dry(args):
    """N ranks, gloo, no GPU: proves the launch path (who starts the ranks, what the line says) on a CPU-only box - the same
    legs as run_sharded at toy sizes with a stub render: configs[3] by image and by band, configs[2]'s horizontal strip by rows,
    configs[4] (64 images) by image, and the one-GPU comparators taken by rank 0 ALONE while the others block on the store"""
    import torch
    import torch.distributed as dist
    from imagestitching_amd import dist as D
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = gather_ranks(dist, rank, int(os.environ.get("LOCAL_RANK", rank)))
    solo = Solo(dist, rank)
    ok, t_all = True, time.perf_counter()
    alone_log = []

    def sharded(sizes, direction, split):
        nonlocal ok
        imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in sizes]
        sh = D.ShardedStitch(imgs, direction, {"filter": "bilinear"}, rank, world, 0, split=split)
        be = StubBackend(sh, torch)
        canvas = be.new_canvas() if rank == 0 else None
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.warmup + args.steps):
            D.run_step(sh, be, [None] * len(sizes), canvas, dist)
        dist.barrier()
        dt = (time.perf_counter() - t0) / max(1, args.warmup + args.steps)
        if rank == 0:
            ok = ok and all(bool((canvas[p.Y0:p.Y1, p.X0:p.X1] == p.image + 1).all()) for p in sh.parts)
        return {"resident": {"ms_per_step": round(dt * 1e3, 4), "split": sh.split, "bands_in_place": sum(1 for p in sh.remote if p.in_place),
                             "bands_staged": sum(1 for p in sh.remote if not p.in_place), "host_sink_available": sh.root_rows() is not None}}

    def single(sizes, direction):
        """the one-'GPU' form: the whole strip painted by rank 0 alone; the store shows that nobody else was in a step meanwhile"""
        imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in sizes]
        sh = D.ShardedStitch(imgs, direction, {"filter": "bilinear"}, 0, 1, 0, split="image")
        be = StubBackend(sh, torch)
        canvas = be.new_canvas()
        t0 = time.perf_counter()
        for _ in range(args.warmup + args.steps):
            be.render_root([None] * len(sizes), canvas)
        dt = (time.perf_counter() - t0) / max(1, args.warmup + args.steps)
        alone_log.append(os.getpid())
        return {"resident": {"ms_per_step": round(max(dt, 1e-6) * 1e3, 4)}}

    small, small5 = [(64, 48)] * 9, [(80, 60)] * 64
    regions = {}
    for split in ("image", "band"):
        for k, v in sharded(small, "vertical", split).items():
            regions["%s/%s" % (k, split)] = v
    for k, v in sharded(small, "horizontal", "auto").items():
        regions["%s/rows_horizontal" % k] = v
    one = solo(lambda: single(small, "vertical"))
    one_h = solo(lambda: single(small, "horizontal"))
    r5 = {"%s/image" % k: v for k, v in sharded(small5, "vertical", "image").items()}
    one5 = solo(lambda: single(small5, "vertical"))
    dt = time.perf_counter() - t_all
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        scaling = scaling_table({k: v for k, v in regions.items() if not k.endswith("/rows_horizontal")}, one)
        scaling.update(scaling_table({k: v for k, v in regions.items() if k.endswith("/rows_horizontal")}, one_h))
        args.out.emit(json.dumps({"metric": "dry launch (no GPU, gloo, stub render)", "value": 0.0, "unit": "MP/s", "n_gpus": world, "steps": args.steps,
                                  "warmup": args.warmup, "ms_per_step": round(dt * 1e3 / max(1, 4 * (args.steps + args.warmup)), 4), "dry_launch": True,
                                  "config": {"timed_region": SCALING_NOTE},
                                  "extra": {"ranks_seen": seen, "world_size": world, "strip_assembled": ok, "regions": regions,
                                            "one_gpu_same_lease": {"vertical": one, "horizontal": one_h, "timed_by_pids": sorted(set(alone_log))},
                                            "scaling": scaling,
                                            "config5": {"regions": r5, "one_gpu": one5, "scaling": scaling_table(r5, one5)}}}))
        if not ok:
            raise SystemExit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--quick", action="store_true", help="headline config only")
    ap.add_argument("--kernels-only", action="store_true", help="only the resident-input kernel configurations (what the rocprofv3 passes run)")
    ap.add_argument("--print-kernel-sha", action="store_true")
    ap.add_argument("--dry-launch", action="store_true", help="N ranks on CPU (gloo, stub render): exercises the rank launch and the line, no GPU")
    ap.add_argument("--force-gloo", action="store_true", help="rehearse the fallback taken when RCCL cannot be initialised (gloo control plane, no gather regions, replicas-only value)")
    ap.add_argument("--no-config5", action="store_true", help="skip the BASELINE configs[4] (64 x 8000x6000) legs")
    ap.add_argument("--no-config5-host", action="store_true", help="configs[4]: skip the one-GPU host_in_host_out comparator (24.6 GB of pinned host memory on rank 0)")
    ap.add_argument("--launch-timeout", type=int, default=1500, help="seconds the self-started ranks may take")
    ap.add_argument("--group-timeout", type=int, default=300, help="seconds each single-process device-group leg may take")
    ap.add_argument("--group-leg", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--group-part", default="all", choices=["all", "host", "resident"], help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.print_kernel_sha:
        print(kernel_source_sha())
        return
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    args.out = StdoutGuard()
    have_ranks = "WORLD_SIZE" in os.environ
    if args.group_leg:
        group_leg(args.group_leg, args.steps, args.out, args.group_part, not args.no_config5)
    elif args.gpus > 1 and not have_ranks:
        launch_ranks(args)                 # nothing above has imported torch or touched HIP
    elif args.dry_launch:
        run_dry(args)
    elif args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("IST_BENCH_FORCE_SHARDED"):
        run_sharded(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
