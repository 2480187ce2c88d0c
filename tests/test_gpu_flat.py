"""The flat form of a job (ist_compile.cpp compile_flat_twin): a strip whose every op covers whole canvas rows at unit
scale is a copy of contiguous byte ranges when the caller's rows are dense, and the launch then walks the same bytes as rows of 32 KiB.
Same pixels as the oracle and as the row form; never a byte outside the canvas; only when it applies.  Reference anchor of the
workload: the vertical strip of equal-width photos, pages/index/index.js:1251-1581 (BASELINE configs[1], configs[4])."""
import ctypes as C

import numpy as np
import pytest
import torch

import imagestitching_amd as ist
from imagestitching_amd import _lib as L
from tests import util as U

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _flat():
    return L.lib.ist_debug_flat_launches()


def _launch(job, srcs, out):
    before = _flat()
    job.launch(srcs, out)
    torch.cuda.synchronize()
    return _flat() - before


def _dense(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _pitched(a, pad_px):
    h, w = a.shape[:2]
    base = torch.full((h, w + pad_px, 4), 0xEE, dtype=torch.uint8, device=DEV)
    base[:, :w] = torch.from_numpy(a).to(DEV)
    return base[:, :w]


def _guarded_canvas(h, w, pitch_px=None, guard=1 << 16):
    """a canvas view with `guard` bytes of 0xAB before and after its pixels"""
    pitch_px = pitch_px or w
    raw = torch.full((guard + h * pitch_px * 4 + guard,), 0xAB, dtype=torch.uint8, device=DEV)
    view = raw[guard:guard + h * pitch_px * 4].view(h, pitch_px, 4)[:, :w]
    return raw, view, guard


def _guards_intact(raw, guard):
    return bool((raw[:guard] == 0xAB).all()) and bool((raw[-guard:] == 0xAB).all())


@pytest.mark.parametrize("opaque", [True, False])
@pytest.mark.parametrize("gap", [0, 7])
def test_vertical_strip_of_dense_rows_takes_the_flat_form_and_makes_the_oracles_pixels(opaque, gap):
    w = 612                                              # 2448-byte rows: no multiple of anything the memory system likes
    heights = [411, 1, 289, 350, 13]                     # one image shorter than the head row of its byte range
    px = [U.rand_image(70 + i, h, w, opaque=opaque) for i, h in enumerate(heights)]
    opts = {"filter": "bilinear", "gap": gap}
    ref, pd, _ = U.oracle_stitch(px, "vertical", opts)
    st = ist.Stitcher(0)
    imgs = [{"width": w, "height": h, "opaque": opaque} for h in heights]
    p, job = st.compile(imgs, "vertical", opts)
    assert (p.canvas_w, p.canvas_h) == (int(pd["canvas_w"]), int(pd["canvas_h"])) and p.canvas_w * p.canvas_h * 4 >= 64 * 32768
    raw, out, guard = _guarded_canvas(p.canvas_h, p.canvas_w)
    assert _launch(job, [_dense(a) for a in px], out) == 1
    assert np.array_equal(out.cpu().numpy(), ref)
    assert _guards_intact(raw, guard)
    # the same job on rows that are not dense: the row form, the same pixels
    raw2, out2, guard2 = _guarded_canvas(p.canvas_h, p.canvas_w, pitch_px=p.canvas_w + 20)
    assert _launch(job, [_dense(a) for a in px], out2) == 0
    assert np.array_equal(out2.cpu().numpy(), ref) and _guards_intact(raw2, guard2)
    out.zero_()
    assert _launch(job, [_pitched(a, 12) if i == 2 else _dense(a) for i, a in enumerate(px)], out) == 0
    assert np.array_equal(out.cpu().numpy(), ref)


def test_the_host_entry_point_takes_it_too():
    w, heights = 700, [400, 333, 267]
    px = [U.rand_image(90 + i, h, w) for i, h in enumerate(heights)]
    ref, _, _ = U.oracle_stitch(px, "vertical", {"filter": "nearest"})
    before = _flat()
    got = ist.stitch(U.hip_images(px), "vertical", {"filter": "nearest"})
    assert _flat() - before == 1
    assert np.array_equal(got["data"], ref)


def test_jobs_it_does_not_apply_to_keep_the_row_form():
    st = ist.Stitcher(0)
    w, h = 640, 480

    def run(imgs_px, direction, opts):
        imgs = [{"width": a.shape[1], "height": a.shape[0], "opaque": True} for a in imgs_px]
        p, job = st.compile(imgs, direction, opts)
        out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device=DEV)
        n = _launch(job, [_dense(a) for a in imgs_px], out)
        ref, _, _ = U.oracle_stitch(imgs_px, direction, opts)
        assert U.max_abs_diff(out.cpu().numpy(), ref) <= (0 if opts["filter"] == "nearest" else 1)
        return n

    same = [U.rand_image(100 + i, h, w) for i in range(4)]
    assert run(same, "horizontal", {"filter": "nearest"}) == 0                       # a canvas row holds four images
    mixed = [U.rand_image(110, 480, 640), U.rand_image(111, 300, 400), U.rand_image(112, 480, 640)]
    assert run(mixed, "vertical", {"filter": "bilinear", "mode": "max"}) == 0         # one image is resampled
    assert run(mixed, "vertical", {"filter": "nearest", "mode": "original"}) == 0     # one image is narrower than the canvas
    small = [U.rand_image(120 + i, 100, 612) for i in range(3)]
    assert run(small, "vertical", {"filter": "nearest"}) == 0                         # under 64 rows of 32 KiB
    st2 = ist.Stitcher(0)                                                             # a large strip of small images (tools/exp_thin.py)
    p2, job2 = st2.compile([{"width": 2000, "height": 100, "opaque": True}] * 100, "vertical", {"filter": "nearest"})
    thin = [torch.empty((100, 2000, 4), dtype=torch.uint8, device=DEV).random_(0, 256) for _ in range(100)]
    out2 = torch.empty((p2.canvas_h, p2.canvas_w, 4), dtype=torch.uint8, device=DEV)
    assert _launch(job2, thin, out2) == 0 and torch.equal(out2, torch.cat(thin, 0))
    aligned = [U.rand_image(130 + i, 100, 4096) for i in range(3)]
    assert run(aligned, "vertical", {"filter": "nearest"}) == 0                       # 16 KiB rows already
    kib4 = [U.rand_image(133 + i, 300, 1024) for i in range(3)]
    assert run(kib4, "vertical", {"filter": "nearest"}) == 1                          # 4 KiB rows are not in the stores' best class


def test_raw_op_lists_source_crops_holes_and_a_draw_over_a_draw():
    """ist_job_create with a hand-written op list: a draw that takes rows 5.. of its bitmap, the same bitmap drawn twice, a HOLE band
    nobody writes, a fill band - all whole rows, so the flat form applies - against the same list on a pitched canvas (row form)."""
    w, H = 777, 1000
    a = U.rand_image(140, 300, w)
    b = U.rand_image(141, 500, w, opaque=False)
    descs = (L.ImageDesc * 2)()
    for i, im in enumerate((a, b)):
        descs[i].width, descs[i].height, descs[i].opaque = w, im.shape[0], int(i == 0)

    def op(kind, image, dy, dh, sy=0, rgba=(0, 0, 0, 0)):
        o = L.Op()
        o.kind, o.image = kind, image
        o.m[0] = o.m[3] = 1.0
        o.s[0], o.s[1], o.s[2], o.s[3] = 0, sy, w, dh
        o.d[0], o.d[1], o.d[2], o.d[3] = 0, dy, w, dh
        for k in range(4):
            o.rgba[k] = rgba[k]
        return o

    ops = [op(0, -1, 0, H, rgba=(255, 255, 255, 255)), op(1, 0, 0, 295, sy=5), op(1, 1, 295, 500), op(0, -1, 795, 50, rgba=(10, 200, 30, 255)),
           op(1, 0, 845, 100, sy=100), op(2, -1, 945, 55), op(1, 0, 400, 120, sy=17)]         # the last one paints over image 1's middle
    arr = (L.Op * len(ops))(*ops)
    st = ist.Stitcher(0)
    job = st.compile_ops(w, H, arr, len(ops), descs, 2, "nearest", clear=(0, 0, 0, 0))
    srcs = [_dense(a), _dense(b)]
    raw, out, guard = _guarded_canvas(H, w)
    raw2, out2, guard2 = _guarded_canvas(H, w, pitch_px=w + 9)
    assert _launch(job, srcs, out2) == 0
    n = _launch(job, srcs, out)
    got, want = out.cpu().numpy(), out2.cpu().numpy()
    assert np.array_equal(got, want) and _guards_intact(raw, guard) and _guards_intact(raw2, guard2)
    assert (got[945:] == 0xAB).all() and np.array_equal(got[:295], a[5:300]) and np.array_equal(got[400:520], a[17:137])
    assert n in (0, 1)          # a draw over a translucent draw is a paint stack: the compiler may keep such a job on the general path
    # without the overlapping draw the job is fill / copy only and must take the flat form
    arr = (L.Op * (len(ops) - 1))(*ops[:-1])
    job = st.compile_ops(w, H, arr, len(ops) - 1, descs, 2, "nearest", clear=(0, 0, 0, 0))
    raw, out, guard = _guarded_canvas(H, w)
    assert _launch(job, srcs, out2) == 0 and _launch(job, srcs, out) == 1
    assert np.array_equal(out.cpu().numpy(), out2.cpu().numpy()) and _guards_intact(raw, guard)


def test_headline_geometry_in_both_forms():
    """BASELINE configs[1] (9 x 4032x3024 vertical): dense rows take the flat form, a padded canvas the row form; both are the images in order"""
    n, w, h = 9, 4032, 3024
    st = ist.Stitcher(0)
    p, job = st.compile([{"width": w, "height": h, "opaque": True}] * n, "vertical", {"filter": "bilinear"})
    srcs = [torch.empty((h, w, 4), dtype=torch.uint8, device=DEV).random_(0, 256) for _ in range(n)]
    raw, out, guard = _guarded_canvas(p.canvas_h, p.canvas_w)
    assert _launch(job, srcs, out) == 1 and _guards_intact(raw, guard)
    for i in range(n):
        assert torch.equal(out[i * h:(i + 1) * h], srcs[i])
    raw2, out2, guard2 = _guarded_canvas(p.canvas_h, p.canvas_w, pitch_px=w + 32)
    assert _launch(job, srcs, out2) == 0 and _guards_intact(raw2, guard2)
    assert torch.equal(out2, out)


def test_a_clip_of_whole_rows_is_a_shorter_flat_canvas():
    """the band jobs of a device group and of the file pipeline: ist_job_create with a clip that takes whole rows, launched on a compact
    band buffer (dst biased by the clip's origin, ist_job_launch's contract)"""
    w, heights = 612, [500, 450, 450]
    px = [U.rand_image(150 + i, h, w, opaque=(i != 1)) for i, h in enumerate(heights)]
    ref, pd, _ = U.oracle_stitch(px, "vertical", {"filter": "nearest"})
    st = ist.Stitcher(0)
    imgs = [{"width": w, "height": h, "opaque": i != 1} for i, h in enumerate(heights)]
    p = ist.plan(imgs, "vertical", {"filter": "nearest"})
    ops, n_ops = p.ops()
    srcs = [_dense(a) for a in px]
    for (y0, y1, flat) in ((200, 1150, 1), (0, 900, 1), (501, 1400, 1), (600, 700, 0)):
        job = st.compile_ops(p.canvas_w, p.canvas_h, ops, n_ops, p._descs, 3, "nearest", clear=(0, 0, 0, 0), clip=(0, y0, w, y1 - y0))
        raw, band, guard = _guarded_canvas(y1 - y0, w)
        before = _flat()
        job.launch_ptrs([t.data_ptr() for t in srcs], [t.stride(0) for t in srcs], band.data_ptr() - y0 * w * 4, w * 4, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert _flat() - before == flat, (y0, y1)
        assert np.array_equal(band.cpu().numpy(), ref[y0:y1]) and _guards_intact(raw, guard), (y0, y1)
    # a clip that does not take whole rows keeps the row form
    job = st.compile_ops(p.canvas_w, p.canvas_h, ops, n_ops, p._descs, 3, "nearest", clear=(0, 0, 0, 0), clip=(10, 0, w - 10, 1400))
    out = torch.zeros((1400, w, 4), dtype=torch.uint8, device=DEV)
    assert _launch(job, srcs, out) == 0
    assert np.array_equal(out.cpu().numpy()[:, 10:], ref[:, 10:])


def test_the_pitch_a_job_prefers():
    """ist_job_preferred_dst_pitch: dense rows where the flat form applies, rows padded to 4 KiB where it does not; a canvas made with
    it renders the oracle's pixels either way"""
    st = ist.Stitcher(0)
    same = [U.rand_image(160 + i, 400, 700) for i in range(3)]
    p, job = st.compile([{"width": 700, "height": 400, "opaque": True}] * 3, "vertical", {"filter": "nearest"})
    assert job.preferred_pitch == 700 * 4
    out = job.empty_canvas()
    assert out.stride(0) == 700 * 4 and _launch(job, [_dense(a) for a in same], out) == 1
    assert np.array_equal(out.cpu().numpy(), U.oracle_stitch(same, "vertical", {"filter": "nearest"})[0])
    p, job = st.compile([{"width": 700, "height": 400, "opaque": True}] * 3, "horizontal", {"filter": "nearest"})
    assert job.preferred_pitch == 12288 and p.canvas_w == 2100      # 8400 bytes -> the next multiple of 4096
    out = job.empty_canvas()
    assert out.stride(0) == 3072 * 4 and out.shape == (400, 2100, 4) and out.data_ptr() % 4096 == 0
    assert _launch(job, [_dense(a) for a in same], out) == 0
    assert np.array_equal(out.cpu().numpy(), U.oracle_stitch(same, "horizontal", {"filter": "nearest"})[0])
    assert L.lib.ist_job_preferred_dst_pitch(None) == 0
