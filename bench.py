#!/usr/bin/env python3
"""bench.py — stitched output megapixels / second on MI355X, with the HBM roofline of the resample+blit kernel and the
CPU oracle timed beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N = 1  : BASELINE.json configs[1] — 9 x 4032x3024 RGBA vertical stitch, bilinear, inputs resident in HBM, one fused
         launch per step (a "step" = one whole stitch).  Three buffer sets are rotated.
N > 1  : BASELINE.json configs[3] — the same stitch with image i on GPU i mod N, bands gathered to GPU 0 over RCCL
         (strong scaling: total work fixed).  See DESIGN.md section 6 for why this cannot beat one GPU when the
         inputs are already resident in HBM.
Prints ONE JSON line on rank 0.
"""
import argparse
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
MIXED = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]


def synth(k, w, h, device):
    """image k = default_rng(1000+k) uniform bytes, alpha forced to 255."""
    import numpy as np
    import torch
    a = np.random.default_rng(1000 + k).integers(0, 256, (h, w, 4), dtype=np.uint8)
    a[..., 3] = 255
    return torch.from_numpy(a).to(device)


def time_job(job, sets, outs, steps, warmup, torch):
    """steps launches, rotating buffer sets; returns (wall seconds, event milliseconds)."""
    n = len(sets)
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


def cpu_baseline(budget_s=12.0):
    """The CPU oracle (a port: the reference's raster is the closed WeChat client) on the same workload, all host
    cores by output row bands, bounded to ~budget_s of CPU work."""
    import numpy as np
    from oracle import oracle as O
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    px = [np.random.default_rng(1000 + k).integers(0, 256, (3024, 4032, 4), dtype=np.uint8) for k in range(9)]
    for a in px:
        a[..., 3] = 255
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
    # single-thread figure on one repetition, for the record
    t1 = time.perf_counter()
    O.render(pd, rl, descs, px, "bilinear", 1, out=out)
    st = time.perf_counter() - t1
    return {"value": round(mp / dt, 1), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": "%d x the full 9x4032x3024 vertical bilinear stitch (109.7 MP each), oracle/ist_oracle.c, %d threads by row bands" % (reps, threads),
            "single_thread_MPs": round(mp / st, 1)}


def run_single(args):
    import torch
    import imagestitching_amd as ist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    st = ist.Stitcher(0)
    nsets = 3
    results = {}
    for name, sizes, direction in (("uniform_vertical", UNIFORM, "vertical"), ("uniform_horizontal", UNIFORM, "horizontal"),
                                   ("mixed_vertical", MIXED, "vertical"), ("mixed_horizontal", MIXED, "horizontal")):
        if args.quick and name != "uniform_vertical":
            continue
        imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in sizes]
        p, job = st.compile(imgs, direction, {"filter": "bilinear"})
        sets = [[synth(9 * s + k, w, h, dev) for k, (w, h) in enumerate(sizes)] for s in range(nsets)]
        outs = [torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=dev) for _ in range(nsets)]
        steps = args.steps if name == "uniform_vertical" else max(10, args.steps // 2)
        wall, ev_ms = time_job(job, sets, outs, steps, args.warmup, torch)
        mp = p.canvas_w * p.canvas_h / 1e6
        k_us = ev_ms * 1e3 / steps
        results[name] = {"canvas": [p.canvas_w, p.canvas_h], "out_MP": round(mp, 3), "steps": steps,
                         "ms_per_step": wall * 1e3 / steps, "kernel_us": k_us, "MPs": mp / (wall / steps),
                         "algorithmic_bytes": job.info["algorithmic_bytes"],
                         "GBs": job.info["algorithmic_bytes"] / (k_us * 1e-6) / 1e9,
                         "tiles": {k: job.info[k] for k in ("tiles_fill", "tiles_copy", "tiles_sample", "tiles_general")}}
        del sets, outs, job
        torch.cuda.empty_cache()
    head = results["uniform_vertical"]
    # yardstick: the runtime's own device-to-device copy (torch copy_ = hipMemcpyDtoD kernel) of the same number of
    # bytes, on the same box in the same process: what a plain copy reaches next to the stitch kernel
    yard = None
    try:
        nbytes = head["algorithmic_bytes"] // 2
        a = [torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 256) for _ in range(2)]
        b = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        for i in range(5):
            b[i % 2].copy_(a[i % 2])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for i in range(reps):
            b[i % 2].copy_(a[i % 2])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        yard = {"us": round(us, 2), "GBs": round(2 * nbytes / (us * 1e-6) / 1e9, 1), "frac": round(2 * nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "what": "torch Tensor.copy_ device-to-device, %d bytes read + as many written" % nbytes}
        del a, b
        torch.cuda.empty_cache()
    except Exception as ex:       # informational only
        yard = {"error": repr(ex)}
    cpu = None if args.no_cpu else cpu_baseline()
    # HBM traffic per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE
    # doubled per MI355X_MICROARCH.md): collected by profiles/summarize.py, committed as profiles/r01_pmc.json
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc.json")) as f:
            pm = json.load(f)["uniform_vertical (BASELINE configs[1])"]
        traffic = int(pm["FETCH_SIZE_bytes"] + pm["WRITE_SIZE_bytes"])
        traffic_src = "profiles/r01_pmc.json (rocprofv3 --pmc passes of this bench; bytes per launch)"
    except Exception:
        pass
    line = {
        "metric": "stitched megapixels/sec (9x12 MP vertical)", "value": round(head["MPs"], 1), "unit": "MP/s",
        "n_gpus": 1, "steps": head["steps"], "warmup": args.warmup, "ms_per_step": round(head["ms_per_step"], 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 9 x 4032x3024 RGBA8 vertical stitch, bilinear resample to common width, "
                               "caps lifted -> 4032x27216 (109.73 MP); inputs and output resident in HBM, one fused launch per stitch",
                   "buffer_sets_rotated": nsets, "timed_region": "kernel launches only (no H2D/D2H, no PNG)"},
        "roofline": {"bound": "hbm", "achieved": round(head["GBs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(head["GBs"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "ist_stitch_kernel", "kernel_us": round(head["kernel_us"], 2),
                     "algorithmic_bytes_per_launch": head["algorithmic_bytes"]},
        "cpu_baseline": cpu,
        "d2d_copy_yardstick": yard,
        "extra": {k: {"MPs": round(v["MPs"], 1), "kernel_us": round(v["kernel_us"], 2), "GBs": round(v["GBs"], 1),
                      "frac": round(v["GBs"] / HBM_PEAK_GBS, 4), "canvas": v["canvas"], "tiles": v["tiles"]}
                  for k, v in results.items()},
    }
    print(json.dumps(line))


def run_sharded(args):
    import torch
    import torch.distributed as dist
    import imagestitching_amd as ist
    from imagestitching_amd import dist as D
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    imgs = [{"width": w, "height": h, "opaque": True} for (w, h) in UNIFORM]
    sh = D.ShardedStitch(imgs, "vertical", {"filter": "bilinear"}, rank, world, 0)
    be = D.HipBackend(sh, local)
    srcs = [synth(k, w, h, dev) if D.owner_of(k, world) == rank else None for k, (w, h) in enumerate(UNIFORM)]
    canvas = be.new_canvas() if rank == 0 else None
    for _ in range(args.warmup):
        D.run_step(sh, be, srcs, canvas, dist)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        D.run_step(sh, be, srcs, canvas, dist)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)

    # phase breakdown (informational): slowest rank's local launches alone, no exchange
    def local_only():
        if rank == 0:
            be.render_root(srcs, canvas)
        else:
            for i in sh.mine:
                be.render_band(i, srcs)
    for _ in range(3):
        local_only()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        local_only()
    torch.cuda.synchronize()
    loc = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
    dist.all_reduce(loc, op=dist.ReduceOp.MAX)

    # informational: N independent replicas (every GPU stitches a whole 9 x 12 MP job; no exchange) = the layout a
    # stitching service would use when jobs are independent
    st = ist.Stitcher(local)
    p_full, job_full = st.compile(imgs, "vertical", {"filter": "bilinear"})
    full_src = [s_ if s_ is not None else synth(k, w, h, dev) for k, ((w, h), s_) in enumerate(zip(UNIFORM, srcs))]
    full_out = torch.empty((p_full.canvas_h, p_full.canvas_w, 4), dtype=torch.uint8, device=dev)
    for _ in range(3):
        job_full.launch(full_src, full_out)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        job_full.launch(full_src, full_out)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    rep = torch.tensor([time.perf_counter() - t2], device=dev, dtype=torch.float64)
    dist.all_reduce(rep, op=dist.ReduceOp.MAX)
    if rank == 0:
        sec = float(dt.item()) / args.steps
        mp = sh.plan.canvas_w * sh.plan.canvas_h / 1e6
        in_place = sum(1 for i in sh.remote if sh.in_place[i])
        # bytes the gather moves into GPU 0 per step, and what that is per xGMI link (one link per sending GPU)
        gather_bytes = sum((sh.boxes[i][2] - sh.boxes[i][0]) * (sh.boxes[i][3] - sh.boxes[i][1]) * 4 for i in sh.remote)
        senders = len({D.owner_of(i, world) for i in sh.remote})
        busiest = max([sum((sh.boxes[i][2] - sh.boxes[i][0]) * (sh.boxes[i][3] - sh.boxes[i][1]) * 4 for i in sh.remote if D.owner_of(i, world) == r) for r in range(world) if r != 0] or [0])
        line = {
            "metric": "stitched megapixels/sec (9x12 MP vertical)", "value": round(mp / sec, 1), "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec * 1e3, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: 9 x 4032x3024 vertical stitch, image i on GPU i mod %d, bands gathered to "
                                   "GPU 0 with one grouped RCCL send/recv batch (%d bands received in place)" % (world, in_place),
                       "timed_region": "per-rank band launches + gather + root launch; inputs resident in each owner's HBM"},
            "roofline": None, "cpu_baseline": None,
            "extra": {"gather": {"bytes_into_gpu0_per_step": gather_bytes, "sending_gpus": senders, "busiest_link_bytes": busiest,
                                 "GBs_into_gpu0": round(gather_bytes / sec / 1e9, 1), "busiest_link_GBs": round(busiest / sec / 1e9, 1),
                                 "note": "xGMI is point to point (one link per GPU pair, ~153 GB/s peak per direction pair): the step time is bounded below by busiest_link_bytes / link rate"},
                      "local_launches_only_ms_per_step": round(float(loc.item()) / args.steps * 1e3, 5),
                      "exchange_ms_per_step_by_difference": round((float(dt.item()) - float(loc.item())) / args.steps * 1e3, 5),
                      "replicas_no_exchange": {"MPs": round(world * mp / (float(rep.item()) / args.steps), 1), "scaling": "weak",
                                               "note": "every GPU stitches its own whole 9x12 MP job"}},
        }
        print(json.dumps(line))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--quick", action="store_true", help="headline config only")
    args = ap.parse_args()
    if args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("IST_BENCH_FORCE_SHARDED"):
        run_sharded(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
