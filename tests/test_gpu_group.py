"""The single-process device group (ist_group_*, include/imagestitch.h): stitch(images, direction, {devices}) from one
process - what the N-API host binds (`devices`).  A one-GPU box exercises everything but the
cross-device transfer: a device listed several times serves several slots (their bands render straight into the
canvas), and the tuning knob IST_GROUP_SELF_SEND=1 routes those bands through ncclSend/ncclRecv to the same rank, so
the RCCL binding itself (dlopen, communicator, grouped batch, in-place and staged receives) runs here too."""
import os
import subprocess
import sys

import numpy as np
import pytest

import imagestitching_amd as ist
from tests import util as U

pytestmark = pytest.mark.gpu

SIZES = [(403, 302), (302, 403), (400, 300), (192, 108), (640, 480)]


@pytest.mark.parametrize("devices,split", [([0], "image"), ([0, 0], "image"), ([0, 0], "band"), ([0, 0, 0, 0, 0], "band"), ([0] * 8, "image"),
                                           ([0, 0, 0], "rows"), ([0] * 8, "rows"), ([0] * 8, "auto")])
@pytest.mark.parametrize("direction,opts", [
    ("vertical", {"filter": "bilinear", "mode": "max", "gap": 3}),
    ("horizontal", {"filter": "nearest", "mode": "min", "gap": 0}),
    ("vertical", {"filter": "bilinear", "mode": "original", "gap": 5}),
])
def test_host_path_on_a_device_list_matches_the_single_device_result(devices, split, direction, opts):
    px = [U.rand_image(800 + i, h, w, opaque=(i != 1)) for i, (w, h) in enumerate(SIZES)]
    one = ist.stitch(px, direction, opts)
    many = ist.stitch(px, direction, dict(opts, devices=devices, split=split))
    assert (many["width"], many["height"]) == (one["width"], one["height"])
    assert np.array_equal(many["data"], one["data"])           # same kernels, same arithmetic: bit-identical
    ref, _, _ = U.oracle_stitch(px, direction, opts)
    assert U.max_abs_diff(many["data"], ref) <= (0 if opts["filter"] == "nearest" else 1)


def test_device_resident_group_job_with_partial_holdings():
    """ist_group_job_launch: one pointer per part; a slot holds only the source rows its band samples"""
    import torch
    px = [U.rand_image(820 + i, h, w) for i, (w, h) in enumerate(SIZES)]
    g = ist.StitchGroup([0, 0, 0])
    job = g.compile(U.hip_images(px), "vertical", {"filter": "bilinear", "mode": "max", "split": "band"})
    assert {p["slot"] for p in job.parts} == {0, 1, 2} and all(p["device"] == 0 for p in job.parts)
    full = [torch.from_numpy(a).cuda() for a in px]
    srcs = []
    for p in job.parts:
        a, b = p["rows"]
        t = full[p["image"]]
        hold = torch.empty((b - a + 1, t.shape[1], 4), dtype=torch.uint8, device="cuda")[:b - a]
        hold.copy_(t[a:b])
        srcs.append((hold, a))
    out = torch.full((job.plan.canvas_h, job.plan.canvas_w, 4), 0x5A, dtype=torch.uint8, device="cuda")
    job.launch(srcs, out)
    g.sync()
    ref, _, _ = U.oracle_stitch(px, "vertical", {"filter": "bilinear", "mode": "max"})
    assert U.max_abs_diff(out.cpu().numpy(), ref) <= 1


def test_full_size_nine_photos_on_eight_slots_of_one_gpu():
    """BASELINE configs[3] geometry through the C-ABI group: 9 x 4032x3024, devices = [0]*8, both splits, both directions"""
    import torch
    srcs = [torch.empty((3024, 4032, 4), dtype=torch.uint8, device="cuda").random_(0, 256) for _ in range(9)]
    for t in srcs:
        t[..., 3] = 255
    g = ist.StitchGroup([0] * 8)
    imgs = [{"width": 4032, "height": 3024, "opaque": True}] * 9
    for direction, dim in (("vertical", 0), ("horizontal", 1)):
        for split in ("image", "band", "rows"):
            job = g.compile(imgs, direction, {"filter": "bilinear", "split": split})
            assert len(job.parts) == (72 if (split, direction) == ("rows", "horizontal") else 16 if split == "rows" else len(job.parts))
            out = torch.full((job.plan.canvas_h, job.plan.canvas_w, 4), 0x5A, dtype=torch.uint8, device="cuda")
            job.launch([srcs[p["image"]] for p in job.parts], out)
            g.sync()
            assert torch.equal(out, torch.cat(srcs, dim)), (direction, split)
            job.close()


def test_bad_device_lists_are_refused():
    with pytest.raises(ist.StitchError):
        ist.StitchGroup([0, 99])
    with pytest.raises(ist.StitchError):
        ist.StitchGroup([-1])
    px = [U.rand_image(830, 8, 8)]
    with pytest.raises(ist.StitchError):
        ist.stitch(px, "vertical", {"devices": [0, 4096]})


def test_edge_antialiased_and_overlapping_plans_shard_by_rows_only():
    """fractional seams with edge anti-aliasing (the iOS plan) and the reference's orientation-7 placement (overlapping draws,
    utils/canvas.js:187-192) are refused draw by draw (no single draw's owner owns the seam row) and are bit-identical to the
    one-device result by rows - what "auto", the default, picks for them"""
    same = [U.rand_image(840 + i, 48, 64) for i in range(3)]
    uneven = [U.rand_image(843 + i, h, 40) for i, h in enumerate((30, 20, 50))]      # orientation 7: image 2 lands on [0, 50) and covers image 1 at [10, 30)
    for opts, orient in (({"platform": "ios", "edgeAA": True}, None), ({"filter": "nearest"}, [7, 7, 7]), ({"filter": "bilinear", "gap": 3}, [7, 5, 7])):
        imgs = U.hip_images(uneven if orient else same, orientations=orient)
        one = ist.stitch(imgs, "vertical", opts)
        for split in ("image", "band"):
            with pytest.raises(ist.StitchError) as e:
                ist.stitch(imgs, "vertical", dict(opts, devices=[0, 0], split=split))
            assert e.value.code == -7
        for devices in ([0, 0], [0] * 5):
            many = ist.stitch(imgs, "vertical", dict(opts, devices=devices))
            assert np.array_equal(many["data"], one["data"]), (opts, devices)


def test_rccl_binding_through_self_send():
    """IST_TUNING=1 IST_GROUP_SELF_SEND=1: bands of same-device slots are rendered into band buffers and travel through
    ncclSend / ncclRecv (rank 0 -> rank 0) inside one ncclGroupStart/End: in-place receives (vertical) and staged
    receives + placement launches (horizontal).  Run in a child process (knobs are read only by a process started in
    tuning mode)."""
    code = """
import sys
sys.path.insert(0, %r)
import numpy as np
import imagestitching_amd as ist
from tests import util as U
px = [U.rand_image(850 + i, h, w) for i, (w, h) in enumerate(%r)]
for direction in ("vertical", "horizontal"):
    for split in ("image", "band", "rows"):
        opts = {"filter": "bilinear", "mode": "max", "gap": 2}
        one = ist.stitch(px, direction, opts)
        many = ist.stitch(px, direction, dict(opts, devices=[0, 0, 0], split=split))
        assert np.array_equal(many["data"], one["data"]), (direction, split)
print("self send ok")
""" % (U.ROOT, SIZES)
    env = dict(os.environ, IST_TUNING="1", IST_GROUP_SELF_SEND="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "self send ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_host_sink_every_band_is_read_back_by_its_own_device():
    """with a host-destined result (the reference's export, index.js:1577-1581) every slot renders its
    full-width bands compactly and DMAs each one straight into its byte range of the pinned result; the root's launch delivers
    only the rows no band covers (gaps, the last gap after the strip).  BASELINE configs[3] geometry on eight slots of one
    GPU must be bit-identical to the one-device result, both splits; a horizontal strip falls back to gather + readback."""
    from imagestitching_amd import _lib as L
    px = [U.rand_image(870 + i, 302, 403) for i in range(9)]
    for opts in ({"filter": "bilinear"}, {"filter": "bilinear", "gap": 7}, {"filter": "nearest", "mode": "max", "gap": 2}):
        one = ist.stitch(px, "vertical", opts)
        for split in ("image", "band"):
            many = ist.stitch(px, "vertical", dict(opts, devices=[0] * 8, split=split))
            assert np.array_equal(many["data"], one["data"]), (opts, split)
    # a horizontal strip (configs[2]'s geometry; index.js:1540-1553): cut draw by draw it has no full-width band and falls back to
    # gather + readback; cut by rows - the default - every slot's band is a byte range of the result (host sink)
    for opts in ({"filter": "bilinear"}, {"filter": "bilinear", "gap": 5, "mode": "max"}, {"filter": "nearest", "mode": "original", "gap": 3}):
        one = ist.stitch(px[:5] + [U.rand_image(899, 200, 333)], "horizontal", opts)
        for split in ("image", "rows", "auto"):
            before = L.lib.ist_debug_host_sink_stitches()
            many = ist.stitch(px[:5] + [U.rand_image(899, 200, 333)], "horizontal", dict(opts, devices=[0] * 8, split=split))
            assert np.array_equal(many["data"], one["data"]), (opts, split)
            assert L.lib.ist_debug_host_sink_stitches() - before == (0 if split == "image" else 1)      # host_sink_ok for configs[2]'s geometry


def test_without_rccl_the_host_sink_still_works_and_the_gather_fails_cleanly():
    """ist_group_create no longer touches RCCL; a host where librccl cannot be loaded sees the error at the
    first launch that has to gather between DISTINCT devices - never on the host-sink path, which exchanges nothing.  On one
    GPU the only gather is the tuning-mode self-send; IST_RCCL_UNAVAILABLE=1 makes the loader fail as on such a host: the
    stitch fails with IST_E_NO_DEVICE naming RCCL, and the group stays usable."""
    code = """
import sys
sys.path.insert(0, %r)
import numpy as np
import imagestitching_amd as ist
from tests import util as U
px = [U.rand_image(860 + i, h, w) for i, (w, h) in enumerate(%r)]
try:
    ist.stitch(px, "vertical", {"devices": [0, 0, 0]})
    print("no failure")
except ist.StitchError as e:
    assert e.code == -5 and "RCCL" in e.reason, (e.code, e.reason)
    print("failed cleanly")
g = ist.StitchGroup([0, 0])
try:
    job = g.compile(U.hip_images(px), "horizontal", {})
    print("compiled", len(job.parts))
finally:
    g.close()
print("usable")
""" % (U.ROOT, SIZES)
    env = dict(os.environ, IST_TUNING="1", IST_GROUP_SELF_SEND="1", IST_RCCL_UNAVAILABLE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "failed cleanly" in r.stdout and "usable" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    # the same process without the self-send knob: host sink on a device list, no RCCL anywhere
    code2 = code.replace('print("no failure")', 'print("host sink ok")')
    env2 = dict(os.environ, IST_TUNING="1", IST_RCCL_UNAVAILABLE="1")
    r = subprocess.run([sys.executable, "-c", code2], env=env2, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "host sink ok" in r.stdout and "usable" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_no_device_allocation_after_the_first_call():
    """the group keeps its compiled jobs (LRU) and its band / staging arenas; job tables come from the
    context's pool.  20 calls with the same plan, then alternating plans: zero hipMallocs after warm-up."""
    from imagestitching_amd import _lib as L
    px = [U.rand_image(880 + i, h, w) for i, (w, h) in enumerate(SIZES)]
    cases = [("vertical", {"filter": "bilinear", "mode": "max", "devices": [0, 0, 0]}),
             ("horizontal", {"filter": "bilinear", "mode": "min", "devices": [0, 0, 0], "split": "band"})]
    first = [ist.stitch(px, d, o)["data"].copy() for d, o in cases]
    before = L.lib.ist_debug_device_allocs()
    for k in range(20):
        d, o = cases[k % 2]
        r = ist.stitch(px, d, o)
        assert np.array_equal(r["data"], first[k % 2])
        del r
    assert L.lib.ist_debug_device_allocs() == before


def test_destroying_a_job_does_not_wait_for_an_unrelated_stream():
    """ist_job_destroy waits for the streams the job ran on, not for the device."""
    import time
    import torch
    st = ist.Stitcher(0)
    a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    b = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    side, main = torch.cuda.Stream(), torch.cuda.Stream()     # (explicit streams: the legacy default stream has its own implicit synchronisation rules)
    imgs = [{"width": 128, "height": 96, "opaque": True}] * 3
    srcs = [torch.empty((96, 128, 4), dtype=torch.uint8, device="cuda").random_(0, 256) for _ in range(3)]
    p, job = st.compile(imgs, "vertical", {"filter": "bilinear"})
    out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device="cuda")
    job.launch(srcs, out, stream=main)        # (first launch of the process loads the kernel: not part of what is timed below)
    with torch.cuda.stream(side):
        b.copy_(a)
    torch.cuda.synchronize()
    done = torch.cuda.Event()
    with torch.cuda.stream(side):
        for _ in range(300):                  # ~300 x 0.4 ms of copies on the other stream
            b.copy_(a)
        done.record()
    job.launch(srcs, out, stream=main)
    t0 = time.perf_counter()
    job.close()
    dt = time.perf_counter() - t0
    still_running = not done.query()
    side.synchronize()
    assert still_running and dt < 0.05, (still_running, dt)
