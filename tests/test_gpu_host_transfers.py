"""The host <-> device layer of the host-buffer entry points (imagestitching_amd/csrc/ist_host.cpp).

Round 1 saw an intermittent SIGABRT inside ist_png_encode_rgba8 on a 70000 x 8 image: the call
page-locked the caller's heap block (hipHostRegister), issued a 70000-row pitched hipMemcpy2DAsync from it and walked a
65535-row slab loop.  The library now never registers caller memory and never issues a pitched runtime copy: caller
rows are packed through pinned chunks the library owns, and results come from a pinned pool.  These tests cover what
remains reachable: tall / pitched / tiny / chunk-straddling sources, region readbacks, and the pool's reuse."""
import io

import numpy as np
import pytest

import imagestitching_amd as ist
from imagestitching_amd import _lib as L
from imagestitching_amd import stitch as S_  # noqa: F401  (the function; module access below)
from tests import util as U

pytestmark = pytest.mark.gpu


def test_tall_pitched_source_through_the_one_call_entry_point():
    """>= 70000 rows of 32 bytes, source pitch != row bytes: the exact shape of the copy that aborted in round 1,
    now through ist_stitch_rgba8 (plan + staged upload + launch + pooled readback)."""
    wide = U.rand_image(500, 70001, 11, opaque=False)
    view = wide[:, 2:10]                                     # 8 px of an 11-px buffer: pitch 44, row 32
    other = U.rand_image(501, 9, 8)
    got = ist.stitch([view, other], "vertical", {"filter": "nearest"})
    ref, _, _ = U.oracle_stitch([np.ascontiguousarray(view), other], "vertical", {"filter": "nearest"})
    assert np.array_equal(got["data"], ref)


def test_tall_narrow_png_encode_from_a_heap_block():
    from PIL import Image
    a = U.rand_image(412, 70000, 8, opaque=False)
    for level in (0, 1):
        png = ist.encode_png(a, level=level)
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(png)).convert("RGBA")), a)
    b = U.rand_image(413, 70000, 12, opaque=False)[:, 1:9]   # pitched view
    png = ist.encode_png(b)
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(png)).convert("RGBA")), b)


def test_rows_longer_than_one_staging_chunk_and_sizes_around_the_chunk_boundaries():
    """a staging chunk is 4 MiB: totals of chunk +- one row exercise the piece arithmetic on both sides; a row of exactly
    one chunk (1048576 px, the widest canvas the lifted limits plan) is one piece per row; rows LONGER than a chunk are
    carried in column segments (only reachable through the PNG encoder: the planner caps canvases at 1048576 px)"""
    from PIL import Image
    for w, h in ((1048576, 2), (1024, 1023), (1024, 1024), (1024, 1025), (1024, 4097)):
        px = [U.rand_image(520 + h % 7, h, w), U.rand_image(530, 1, w)]
        got = ist.stitch(px, "vertical", {"filter": "nearest"})
        assert np.array_equal(got["data"], np.concatenate(px, 0)), (w, h)
    a = U.rand_image(531, 2, 1048577 + 300, opaque=False)
    Image.MAX_IMAGE_PIXELS = None
    for src in (a, a[:, 150:150 + 1048577]):                         # contiguous and pitched
        png = ist.encode_png(src, level=0)
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(png)).convert("RGBA")), src)


def test_region_readback_is_compact_and_exact():
    """getImageData-style regions (index.js:1564) come back through a scratch that holds only the region"""
    from imagestitching_amd.stitch import render_ops
    px = [U.rand_image(540, 300, 500), U.rand_image(541, 200, 350)]
    p = ist.plan(U.hip_images(px), "vertical", {"mode": "max"})
    ops, n_ops = p.ops()
    ref, _, _ = U.oracle_stitch(px, "vertical", {"filter": "nearest", "mode": "max"})
    for region in [(0, 0, 1, 1), (17, 33, 141, 270), (499, 0, 1, p.canvas_h), (0, p.canvas_h - 1, 500, 1), (-5, -7, 20, 20)]:
        got = render_ops(p.canvas_w, p.canvas_h, ops, n_ops, p._descs, px, "nearest", region=region)
        x, y, w, h = region
        x0, y0 = max(0, x), max(0, y)
        assert np.array_equal(got, ref[y0:min(p.canvas_h, y + h), x0:min(p.canvas_w, x + w)]), region


def test_result_blocks_return_to_the_pool_and_are_reused():
    px = [U.rand_image(550, 400, 600), U.rand_image(551, 300, 600)]
    a = ist.stitch(px, "vertical", {"filter": "nearest"})
    addr = a["data"].ctypes.data
    want = a["data"].copy()
    del a                                                    # the last view dies: ist_free -> pool
    b = ist.stitch(px, "vertical", {"filter": "nearest"})
    assert b["data"].ctypes.data == addr, "a same-sized result should reuse the pooled pinned block"
    assert np.array_equal(b["data"], want)
    c = ist.stitch(px, "vertical", {"filter": "nearest"})    # b is still alive: a different block
    assert c["data"].ctypes.data != addr and np.array_equal(c["data"], want)
    del b, c
    L.lib.ist_pool_trim()
    d = ist.stitch(px, "vertical", {"filter": "nearest"})
    assert np.array_equal(d["data"], want)


def test_many_small_calls_do_not_disturb_each_other():
    """single-lane path (< 2 MiB): chunks are reused across calls while earlier DMAs may still be in flight"""
    rng = np.random.default_rng(9)
    for k in range(40):
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 120))
        px = [U.rand_image(600 + k, h, w, opaque=False), U.rand_image(700 + k, int(rng.integers(1, 50)), w, opaque=False)]
        got = ist.stitch(px, "vertical", {"filter": "nearest"})
        ref, _, _ = U.oracle_stitch(px, "vertical", {"filter": "nearest"})
        assert np.array_equal(got["data"], ref), k


def test_two_threads_on_two_contexts():
    """C-ABI promise: re-entrant across handles.  Two threads, each with its own context, compile and run
    different jobs concurrently; compiles read no mutable global (ist_compile.cpp CompileKnobs)."""
    import ctypes as C
    import threading
    from imagestitching_amd.stitch import _descs, _limits, _merge, _filter_of, _take_pixels
    errors, results = [], {}

    def worker(tid):
        try:
            ctx = L.lib.ist_ctx_create(0)
            assert ctx
            rng = np.random.default_rng(100 + tid)
            for it in range(12):
                sizes = [(int(rng.integers(20, 400)), int(rng.integers(20, 300))) for _ in range(3)]
                px = [U.rand_image(1000 * tid + 10 * it + i, h, w) for i, (w, h) in enumerate(sizes)]
                direction = ("vertical", "horizontal")[(it + tid) & 1]
                opts = _merge({"filter": ("nearest", "bilinear")[it & 1], "mode": ("min", "max", "original")[it % 3], "gap": it % 4})
                descs = _descs(px)
                n = len(px)
                ptrs, pitches = (C.c_void_p * n)(), (C.c_size_t * n)()
                for i, a in enumerate(px):
                    ptrs[i], pitches[i] = a.ctypes.data, a.strides[0]
                cplan, lim, out = L.Plan(), _limits(opts), C.POINTER(C.c_uint8)()
                rc = L.lib.ist_stitch_rgba8(ctx, descs, ptrs, pitches, n, {"vertical": 0, "horizontal": 1}[direction],
                                            {"min": 0, "max": 1, "original": 2}[opts["mode"]], float(opts["gap"]), C.byref(lim),
                                            _filter_of(opts), C.byref(cplan), C.byref(out))
                assert rc == 0, L.last_error()
                w, h = int(cplan.canvas_w), int(cplan.canvas_h)
                L.lib.ist_plan_free(C.byref(cplan))
                results[(tid, it)] = (px, direction, opts, _take_pixels(out, w, h).copy())
            L.lib.ist_ctx_destroy(ctx)
        except Exception as e:      # noqa: BLE001
            errors.append((tid, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert len(results) == 24
    for (tid, it), (px, direction, opts, got) in results.items():
        ref, _, _ = U.oracle_stitch(px, direction, opts)
        tol = 0 if opts["filter"] == "nearest" else 1
        assert got.shape == ref.shape and U.max_abs_diff(got, ref) <= tol, (tid, it)


def test_a_large_stitch_goes_up_and_comes_down_at_the_same_time():
    """ist_stitch_rgba8, round 4: a stitch of 32 MB or more is cut into row bands; the source rows band b + 1 samples go up in 32 MiB pieces while
    band b comes down (ist_debug_duplex_stitches counts such calls); small stitches take the one-shot path.  Same pixels either way: vertical
    and horizontal strips, gaps, mixed sizes (resampled bands with tap rows across band cuts), a pitched source, translucent pixels over the
    white fill, quarter-turned and mirrored images (whose bands need other rows than their own)."""
    from imagestitching_amd import _lib as L
    sizes = [(2000, 1500), (1800, 1400), (2000, 1600), (1500, 1100)]
    px = [U.rand_image(300 + i, h, w, opaque=(i != 2)) for i, (w, h) in enumerate(sizes)]
    for direction in ("vertical", "horizontal"):
        for opts in ({"filter": "nearest", "gap": 9, "mode": "max"}, {"filter": "bilinear", "gap": 0, "mode": "min"}, {"filter": "nearest", "gap": 3, "mode": "original"}):
            ref, pd, _ = U.oracle_stitch(px, direction, opts)
            before = L.lib.ist_debug_duplex_stitches()
            got = ist.stitch(U.hip_images(px), direction, opts)
            assert L.lib.ist_debug_duplex_stitches() - before == (1 if ref.size >= (32 << 20) else 0), (direction, opts, ref.size)
            assert got["data"].shape == ref.shape and U.max_abs_diff(got["data"], ref) <= (0 if opts["filter"] == "nearest" else 1), (direction, opts)
    # orientations: the rows a band needs are not the band's own rows
    ori = [6, 3, 2, 8]
    for direction in ("vertical", "horizontal"):
        ref, _, _ = U.oracle_stitch(px, direction, {"filter": "nearest", "mode": "max"}, orientations=ori)
        before = L.lib.ist_debug_duplex_stitches()
        got = ist.stitch(U.hip_images(px, ori), direction, {"filter": "nearest", "mode": "max"})
        assert L.lib.ist_debug_duplex_stitches() - before == (1 if ref.size >= (32 << 20) else 0)
        assert np.array_equal(got["data"], ref), direction
    small = [U.rand_image(320 + i, 300, 400) for i in range(3)]
    before = L.lib.ist_debug_duplex_stitches()
    got = ist.stitch(U.hip_images(small), "vertical", {"filter": "nearest"})
    assert L.lib.ist_debug_duplex_stitches() == before and np.array_equal(got["data"], U.oracle_stitch(small, "vertical", {"filter": "nearest"})[0])
    # a pitched source (a view into a wider buffer) through the big pieces
    wide = np.zeros((1500, 2100, 4), np.uint8)
    wide[:, :2000] = px[0]
    imgs = U.hip_images(px)
    imgs[0]["data"] = wide[:, :2000]
    ref, _, _ = U.oracle_stitch(px, "vertical", {"filter": "nearest", "mode": "max"})
    got = ist.stitch(imgs, "vertical", {"filter": "nearest", "mode": "max"})
    assert np.array_equal(got["data"], ref)


def test_page_locked_caller_memory_goes_up_without_staging():
    """a host that keeps its images in pinned blocks (here: torch pinned tensors seen as numpy): the band uploads copy straight from them"""
    import torch
    sizes = [(2000, 1500), (2000, 1400), (2000, 1600), (2000, 1100)]
    px = [U.rand_image(340 + i, h, w) for i, (w, h) in enumerate(sizes)]
    pinned = [torch.from_numpy(a).pin_memory() for a in px]
    imgs = [{"width": a.shape[1], "height": a.shape[0], "data": t.numpy()} for a, t in zip(px, pinned)]
    for direction in ("vertical", "horizontal"):
        ref, _, _ = U.oracle_stitch(px, direction, {"filter": "nearest", "mode": "max"})
        got = ist.stitch(imgs, direction, {"filter": "nearest", "mode": "max"})
        assert np.array_equal(got["data"], ref), direction
