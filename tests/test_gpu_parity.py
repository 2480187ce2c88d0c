"""HIP path vs CPU oracle, through the C-ABI.  Bit-exact for nearest; within +-1 LSB per channel for bilinear
(BASELINE.json north_star).  Needs a real MI355X: run with -m gpu."""
import ctypes as C
import os

import numpy as np
import pytest

import imagestitching_amd as ist
from imagestitching_amd import _lib as L
from oracle import oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu

BILINEAR_TOL = 1   # LSB per channel, stated by BASELINE.json north_star


def _check(pixels, direction, opts=None, orientations=None):
    opts = dict(opts or {})
    ref, pd, rl = U.oracle_stitch(pixels, direction, opts, orientations)
    got = ist.stitch(U.hip_images(pixels, orientations), direction, opts)
    assert (got["width"], got["height"]) == (int(pd["canvas_w"]), int(pd["canvas_h"]))
    out = got["data"]
    assert out.shape == ref.shape
    if opts.get("filter", "bilinear") == "nearest":
        assert np.array_equal(out, ref), "nearest must be bit-exact (%d px differ)" % int((out != ref).any(-1).sum())
    else:
        d = U.max_abs_diff(out, ref)
        assert d <= BILINEAR_TOL, "bilinear max |diff| = %d" % d
    return out, ref


def test_config1_3x640x480_vertical_nearest_exact():
    """BASELINE configs[0]: 3 x 640x480 vertical, nearest, superSample 1 -> 640x1440, bit-exact."""
    px = [O.synth_image(k, 480, 640) for k in range(3)]
    out, _ = _check(px, "vertical", {"filter": "nearest"})
    assert out.shape == (1440, 640, 4)
    for k in range(3):   # identity plan: the strip is the concatenation
        assert np.array_equal(out[480 * k:480 * (k + 1)], px[k])


@pytest.mark.parametrize("filt", ["nearest", "bilinear"])
@pytest.mark.parametrize("direction", ["vertical", "horizontal"])
@pytest.mark.parametrize("mode", ["min", "max", "original"])
def test_mixed_sizes(filt, direction, mode):
    sizes = [(403, 302), (302, 403), (400, 300), (192, 108), (37, 211)]
    px = [U.smooth_image(10 + i, h, w) if i % 2 else U.rand_image(10 + i, h, w) for i, (w, h) in enumerate(sizes)]
    _check(px, direction, {"filter": filt, "mode": mode, "gap": 7})


@pytest.mark.parametrize("filt", ["nearest", "bilinear"])
@pytest.mark.parametrize("orientation", [1, 2, 3, 4, 5, 6, 7, 8])
def test_exif_orientations(filt, orientation):
    """utils/canvas.js:160-200, including the reference's orientation-7 placement (drawn one rect-height up)."""
    px = [U.rand_image(20, 45, 61), U.smooth_image(21, 83, 50), U.rand_image(22, 64, 64)]
    for direction in ("vertical", "horizontal"):
        _check(px, direction, {"filter": filt, "gap": 3}, orientations=[orientation] * 3)


@pytest.mark.parametrize("filt", ["nearest", "bilinear"])
def test_translucent_sources_source_over_white(filt):
    px = [U.rand_image(30 + i, h, w, opaque=False) for i, (w, h) in enumerate([(120, 90), (90, 120), (64, 33)])]
    _check(px, "vertical", {"filter": filt, "mode": "max", "gap": 5})
    _check(px, "horizontal", {"filter": filt, "mode": "original", "gap": 0})


@pytest.mark.parametrize("filt", ["nearest", "bilinear"])
@pytest.mark.parametrize("platform", ["devtools", "ios", "android"])
def test_reference_default_plans_supersample_and_caps(filt, platform):
    """n<7 small images -> the reference super-samples (2.6 / 2.2); big ones -> scaleDown<1 with fractional cursors."""
    px = [U.smooth_image(40 + i, 48, 64) for i in range(3)]
    _check(px, "vertical", {"filter": filt, "platform": platform})
    px = [U.rand_image(50 + i, h, w) for i, (w, h) in enumerate([(403, 302), (108, 192), (400, 300), (192, 108), (403, 302), (108, 192), (400, 300)])]
    _check(px, "vertical", {"filter": filt, "platform": platform, "gap": 10, "maxSide": 512, "maxPixels": 512 * 300})
    _check(px, "horizontal", {"filter": filt, "platform": platform, "gap": 10, "maxSide": 640, "maxPixels": 640 * 200, "mode": "original"})


@pytest.mark.parametrize("filt", ["nearest", "bilinear"])
def test_ragged_and_tiny(filt):
    _check([U.rand_image(60, 1, 1)], "vertical", {"filter": filt})
    _check([U.rand_image(61, 1, 1), U.rand_image(62, 3, 500), U.rand_image(63, 900, 7)], "vertical", {"filter": filt, "gap": 3})
    _check([U.rand_image(64, 5, 3), U.rand_image(65, 9, 13), U.rand_image(66, 2, 2)], "horizontal", {"filter": filt, "mode": "max", "gap": 1})
    _check([U.rand_image(67, 33, 257), U.rand_image(68, 65, 259)], "vertical", {"filter": filt, "mode": "max"})
    _check([U.rand_image(69, 31, 1021)], "horizontal", {"filter": filt})


@pytest.mark.parametrize("filt", ["nearest", "bilinear"])
def test_heavy_downscale(filt):
    """the phone-capped plans shrink 12 MP photos 6.6x: nearest gathers directly, bilinear streams row pairs through LDS."""
    px = [U.smooth_image(120, 300, 2000), U.rand_image(121, 260, 1900), U.rand_image(122, 90, 300)]
    _check(px, "vertical", {"filter": filt, "mode": "min"})
    _check(px, "horizontal", {"filter": filt, "mode": "min"})
    _check(px, "vertical", {"filter": filt, "platform": "android", "maxSide": 256, "maxPixels": 256 * 256})


def test_direct_gather_path_matches_when_lds_staging_is_disabled():
    """Compile knobs are only read by a process STARTED in tuning mode (IST_TUNING=1; production compiles touch no
    environment), so the knob run is a child process; it checks itself against the oracle."""
    import subprocess
    import sys
    code = """
import sys
sys.path.insert(0, %r)
from tests import test_gpu_parity as T, util as U
sizes = [(403, 302), (302, 403), (400, 300), (192, 108)]
px = [U.rand_image(130 + i, h, w, opaque=(i != 2)) for i, (w, h) in enumerate(sizes)]
for direction in ("vertical", "horizontal"):
    T._check(px, direction, {"filter": "bilinear", "mode": "max", "gap": 2})
T._check(px, "vertical", {"filter": "bilinear"}, orientations=[2, 3, 4, 1])
import imagestitching_amd as ist
p, job = ist.Stitcher(0).compile(U.hip_images(px), "vertical", {"filter": "bilinear", "mode": "max"})
assert job.info["tiles_sample"] > 0
print("direct path ok")
""" % (U.ROOT,)
    env = dict(os.environ, IST_TUNING="1", IST_NO_LDS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "direct path ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_lds_staged_path_edges():
    """upscale (tiny footprints), flips (negative scale), 2-pixel sources, ragged widths, translucent taps."""
    px = [U.rand_image(140, 7, 9), U.rand_image(141, 2, 2, opaque=False), U.smooth_image(142, 40, 1000), U.rand_image(143, 33, 517)]
    _check(px, "vertical", {"filter": "bilinear", "mode": "max"})
    _check(px, "vertical", {"filter": "bilinear", "mode": "max"}, orientations=[2, 3, 4, 2])
    _check(px, "horizontal", {"filter": "bilinear", "mode": "max", "gap": 1})
    _check([U.rand_image(144, 300, 700), U.rand_image(145, 450, 1050)], "vertical", {"filter": "bilinear", "mode": "min"})   # 1.5x down
    _check([U.rand_image(146, 300, 700), U.rand_image(147, 1200, 2450)], "vertical", {"filter": "bilinear", "mode": "min"})  # 3.5x down


@pytest.mark.parametrize("direction", ["vertical", "horizontal"])
def test_streamed_path_scales_and_edges(direction):
    """|ky| >= 2 (tile_sample_stream: per-wave LDS rings, counted waits): the threshold itself, 256 / 128 / 64-pixel-wide
    tiles (kx 2..2.9 / ..6 / ..16), beyond it (direct path), flips, translucent taps, ragged right edges (output widths
    that are no multiple of 64, so the store count per row changes), tiles shorter than 8 rows, the bottom-right tile whose
    last source row goes through registers."""
    base = (301, 77) if direction == "vertical" else (77, 301)        # the smallest image fixes the strip's cross size
    for n, k in enumerate([2.0, 2.2, 2.9, 3.5, 6.65, 12.0, 17.0]):
        w, h = int(round(base[0] * k)), int(round(base[1] * k))
        px = [U.rand_image(300 + n, base[1], base[0]), U.rand_image(310 + n, h, w, opaque=(n % 2 == 0)), U.smooth_image(320 + n, h + 3, w + 5)]
        _check(px, direction, {"filter": "bilinear", "mode": "min", "gap": n % 3})
    px = [U.rand_image(330, 129, 517), U.rand_image(331, 400, 1300), U.rand_image(332, 1033, 2068, opaque=False)]
    for o in (2, 3, 4):
        _check(px, direction, {"filter": "bilinear", "mode": "min"}, orientations=[o] * 3)
    p, job = ist.Stitcher(0).compile(U.hip_images(px), direction, {"filter": "bilinear", "mode": "min"})
    assert job.info["tiles_sample"] > 0


def test_empty_input_returns_none():
    assert ist.stitch([], "vertical") is None


def test_missing_bitmap_is_decode_error():
    with pytest.raises(ist.StitchError) as e:
        ist.stitch([{"width": 4, "height": 4, "data": None}], "vertical")
    assert e.value.code == -6 and "解码异常" in str(e.value)


def test_strided_host_buffers():
    """src pitch != width*4 (a view into a wider buffer)."""
    big = U.rand_image(70, 50, 100)
    view = big[:, 10:70]           # 60 px wide, pitch 400
    other = U.rand_image(71, 40, 60)
    ref, _, _ = U.oracle_stitch([np.ascontiguousarray(view), other], "vertical", {"filter": "nearest"})
    got = ist.stitch([view, other], "vertical", {"filter": "nearest"})
    assert np.array_equal(got["data"], ref)


def _device_stitch(px, direction, opts, orientations=None):
    import torch
    st = ist.Stitcher(0)
    p, job = st.compile(U.hip_images(px, orientations), direction, opts)
    srcs = [torch.from_numpy(a).cuda() for a in px]
    out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device="cuda")
    out.fill_(0x5A)                      # poison: every pixel must be overwritten by the single launch
    job.launch(srcs, out)
    torch.cuda.synchronize()
    return p, job, srcs, out


def test_device_path_full_size_vertical_identity():
    """BASELINE configs[1] at full size: 9 x 4032x3024 vertical bilinear. Size-independent property: with equal
    widths the strip is exactly the concatenation of the inputs (bilinear at 1:1 is the identity)."""
    import torch
    px = [O.synth_image(k, 3024, 4032) for k in range(9)]
    p, job, srcs, out = _device_stitch(px, "vertical", {"filter": "bilinear"})
    assert (p.canvas_w, p.canvas_h) == (4032, 27216)
    assert job.info["tiles_copy"] > 0 and job.info["tiles_sample"] == 0 and job.info["tiles_general"] == 0
    assert job.info["algorithmic_bytes"] == 8 * 4032 * 27216
    for k in range(9):
        assert torch.equal(out[3024 * k:3024 * (k + 1)], srcs[k])


def test_device_path_full_size_horizontal_identity():
    """BASELINE configs[2]: 9 x 4032x3024 horizontal -> 36288x3024, column bands."""
    import torch
    px = [O.synth_image(k, 3024, 4032) for k in range(9)]
    p, job, srcs, out = _device_stitch(px, "horizontal", {"filter": "bilinear"})
    assert (p.canvas_w, p.canvas_h) == (36288, 3024)
    for k in range(9):
        assert torch.equal(out[:, 4032 * k:4032 * (k + 1)], srcs[k])


@pytest.mark.parametrize("direction", ["vertical", "horizontal"])
def test_device_path_full_size_mixed_bilinear_vs_oracle(direction):
    """Supplementary mixed-size variant of configs 2/3 at full size, against the oracle."""
    sizes = [(4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024), (3024, 4032), (4000, 3000), (3840, 2160), (4032, 3024)]
    px = [O.synth_image(k, h, w) for k, (w, h) in enumerate(sizes)]
    ref, pd, _ = U.oracle_stitch(px, direction, {"filter": "bilinear"}, threads=16)
    p, job, srcs, out = _device_stitch(px, direction, {"filter": "bilinear"})
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    step = 1 << 22     # compare in slabs to bound host memory
    flat_g, flat_r = got.reshape(-1), ref.reshape(-1)
    worst = 0
    for s in range(0, flat_g.size, step * 4):
        worst = max(worst, int(np.abs(flat_g[s:s + step * 4].astype(np.int16) - flat_r[s:s + step * 4].astype(np.int16)).max()))
    assert worst <= BILINEAR_TOL


def test_clip_region_like_getImageData():
    """index.js:1564 getImageData(0,0,1,1): render only a region of the canvas."""
    import torch
    px = [U.rand_image(80, 60, 80), U.rand_image(81, 30, 40)]
    ref, pd, _ = U.oracle_stitch(px, "vertical", {"filter": "bilinear", "mode": "max"})
    st = ist.Stitcher(0)
    p = ist.plan(U.hip_images(px), "vertical", {"mode": "max"})
    ops, n_ops = p.ops()
    for clip in [(0, 0, 1, 1), (17, 33, 41, 70), (79, 0, 1, 120)]:
        job = st.compile_ops(p.canvas_w, p.canvas_h, ops, n_ops, p._descs, 2, "bilinear", clip=clip)
        out = torch.full((p.canvas_h, p.canvas_w, 4), 0x5A, dtype=torch.uint8, device="cuda")
        job.launch([torch.from_numpy(a).cuda() for a in px], out)
        got = out.cpu().numpy()
        x, y, w, h = clip
        assert U.max_abs_diff(got[y:y + h, x:x + w], ref[y:y + h, x:x + w]) <= 1
        mask = np.ones(got.shape[:2], bool)
        mask[y:y + h, x:x + w] = False
        assert (got[mask] == 0x5A).all(), "pixels outside the clip must not be written"


def test_pitched_device_buffers():
    import torch
    px = [U.rand_image(90, 50, 64), U.rand_image(91, 20, 64)]
    st = ist.Stitcher(0)
    p, job = st.compile(U.hip_images(px), "vertical", {"filter": "nearest"})
    wide = [torch.zeros((a.shape[0], a.shape[1] + 12, 4), dtype=torch.uint8, device="cuda") for a in px]
    for t, a in zip(wide, px):
        t[:, 4:4 + a.shape[1]] = torch.from_numpy(a).cuda()
    srcs = [t[:, 4:4 + a.shape[1]] for t, a in zip(wide, px)]
    canvas = torch.zeros((p.canvas_h, p.canvas_w + 8, 4), dtype=torch.uint8, device="cuda")
    out = canvas[:, 8:]
    job.launch(srcs, out)
    ref, _, _ = U.oracle_stitch(px, "vertical", {"filter": "nearest"})
    assert np.array_equal(out.cpu().numpy(), ref)
    assert int(canvas[:, :8].max()) == 0


def test_randomised_plans_against_oracle():
    """80 seeded random stitches: sizes 1..400, every mode / direction / filter / platform, gaps, EXIF orientations,
    translucent pixels, pitched buffers.  nearest bit-exact, bilinear within 1 LSB."""
    rng = np.random.default_rng(20261004)
    for case in range(80):
        n = int(rng.integers(1, 8))
        sizes = [(int(rng.integers(1, 400)), int(rng.integers(1, 400))) for _ in range(n)]
        if case % 7 == 0:       # near-identical widths: exercises the copy path next to resampled cells
            w0 = int(rng.integers(30, 300))
            sizes = [(w0, int(rng.integers(1, 300))) for _ in range(n)]
        px = [U.rand_image(int(rng.integers(1 << 30)), h, w, opaque=bool(rng.integers(0, 2))) for (w, h) in sizes]
        ori = [int(rng.integers(0, 9)) for _ in range(n)] if case % 3 == 0 else None
        opts = {"filter": ["nearest", "bilinear"][int(rng.integers(0, 2))], "mode": ["min", "max", "original"][int(rng.integers(0, 3))],
                "gap": int(rng.integers(0, 21))}
        plat = [None, "ios", "android", "devtools"][int(rng.integers(0, 4))]
        if plat:
            opts["platform"] = plat
            if rng.integers(0, 2):
                opts["maxSide"] = int(rng.integers(64, 600))
                opts["maxPixels"] = int(opts["maxSide"] * rng.integers(32, 400))
        direction = ["vertical", "horizontal"][int(rng.integers(0, 2))]
        try:
            _check(px, direction, opts, ori)
        except AssertionError as e:
            raise AssertionError("case %d: sizes=%s ori=%s opts=%s dir=%s: %s" % (case, sizes, ori, opts, direction, e))


def test_config5_shape_64_images_single_gpu():
    """BASELINE configs[4] geometry on ONE GPU at 1/5 linear scale per image (64 x 1600x1200 -> 1600x76800): 64 cells,
    the kernarg source table at half capacity.  Property: the strip is the concatenation (uniform sizes)."""
    import torch
    px = [torch.randint(0, 256, (1200, 1600, 4), dtype=torch.uint8, device="cuda") for _ in range(64)]
    st = ist.Stitcher(0)
    p, job = st.compile([{"width": 1600, "height": 1200}] * 64, "vertical", {"filter": "bilinear"})
    assert (p.canvas_w, p.canvas_h) == (1600, 76800) and job.info["n_cells"] == 64
    out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device="cuda")
    job.launch(px, out)
    torch.cuda.synchronize()
    # source-over white: the random alpha makes this the blend path of COPY; check against the integer formula
    src = torch.cat(px, 0).to(torch.int32)
    a = src[..., 3:4]
    want = (src[..., :3] * a + 255 * (255 - a) + 127) // 255
    assert torch.equal(out[..., :3].to(torch.int32), want)
    assert int(out[..., 3].min()) == 255


def test_baseline_config4_full_size_64_x_8000x6000_on_one_gpu():
    """BASELINE configs[4] at its own size on ONE GPU: 64 x 8000x6000 RGBA8 vertical -> 8000x384000 (3072 MP; 12.288 GB in
    + 12.288 GB out, both resident in HBM), one launch.  Size-independent property: with equal widths the strip is the
    concatenation of the inputs (bilinear at 1:1 is the identity); checked on the device, image by image."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 30 * (1 << 30):
        pytest.skip("needs ~25 GB of free HBM")
    st = ist.Stitcher(0)
    p, job = st.compile([{"width": 8000, "height": 6000, "opaque": True}] * 64, "vertical", {"filter": "bilinear"})
    assert (p.canvas_w, p.canvas_h) == (8000, 384000)
    info = job.info
    assert info["n_cells"] == 64 and info["tiles_sample"] == 0 and info["tiles_general"] == 0 and info["tiles_fill"] == 0
    assert info["n_tiles"] == info["tiles_copy"] == 64 * ((8000 + 255) // 256) * (6000 // 8)      # 256 x 8 tiles per image
    assert info["algorithmic_bytes"] == 8 * 8000 * 384000
    srcs = []
    for k in range(64):
        t = torch.empty((6000, 8000, 4), dtype=torch.uint8, device="cuda")
        t.view(torch.int32).random_(-2 ** 31, 2 ** 31 - 1)          # 4 random bytes per pixel in one pass
        t[..., 3] = 255
        srcs.append(t)
    out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device="cuda")
    out[::1000].fill_(0x5A)                                         # poison a sample of rows
    job.launch(srcs, out)
    torch.cuda.synchronize()
    for k in range(64):
        assert torch.equal(out[6000 * k:6000 * (k + 1)], srcs[k]), "image %d" % k
    del srcs, out
    torch.cuda.empty_cache()


def test_one_call_c_entry_point_equals_two_step_path():
    from imagestitching_amd.stitch import stitch_via_c_abi
    px = [U.rand_image(150 + i, h, w, opaque=False) for i, (w, h) in enumerate([(120, 90), (90, 120), (64, 33)])]
    a = ist.stitch(U.hip_images(px), "vertical", {"mode": "max", "gap": 3})
    b = stitch_via_c_abi(U.hip_images(px), "vertical", {"mode": "max", "gap": 3})
    assert (a["width"], a["height"]) == (b["width"], b["height"]) and np.array_equal(a["data"], b["data"])


@pytest.mark.parametrize("filt", ["nearest", "bilinear"])
def test_orientations_at_photo_scale(filt):
    """EXIF 2-8 on larger bitmaps: 1:1 flips (mirrored COPY path), quarter turns at 1:1 and scaled (transposed LDS
    path), translucent pixels, odd sizes that leave ragged tiles."""
    a = U.rand_image(160, 301, 403, opaque=False)       # landscape bitmap
    b = U.smooth_image(161, 403, 301)                   # portrait bitmap
    for o in (2, 3, 4):
        _check([a, a], "vertical", {"filter": filt}, orientations=[o, o])            # 1:1 mirrored copies
    for o in (5, 6, 7, 8):
        # natural size = oriented size (portrait), bitmap stored landscape: pure quarter turn at 1:1
        imgs = [{"width": 301, "height": 403, "bmpWidth": 403, "bmpHeight": 301, "orientation": o, "data": a}] * 2
        descs = [{"width": 301, "height": 403, "bmp_w": 403, "bmp_h": 301, "orientation": o}] * 2
        rc, pd, rl = O.plan(descs, "vertical", "min", 5, U.oracle_limits({}))
        ref = O.render(pd, rl, descs, [a, a], filt, 4)
        got = ist.stitch(imgs, "vertical", {"filter": filt, "gap": 5})["data"]
        assert got.shape == ref.shape
        assert U.max_abs_diff(got, ref) <= (0 if filt == "nearest" else 1), o
        # scaled quarter turns next to a plain image
        _check([a, b, a], "horizontal", {"filter": filt, "mode": "max", "gap": 2}, orientations=[o, 1, o])


@pytest.mark.parametrize("filt", ["nearest", "bilinear"])
def test_edge_antialiasing_of_fractional_rectangles(filt):
    """IST_FILTER_EDGE_AA: fractional edges (gap*scaleDown cursors, ctx.scale(superSample)) blended by area coverage.
    nearest stays bit-exact (the blend runs in fp64 on both sides), bilinear within 1 LSB."""
    px = [U.rand_image(170 + i, h, w, opaque=(i % 2 == 0)) for i, (w, h) in enumerate([(40, 30), (40, 20), (37, 25), (64, 48)])]
    caps = {"platform": "android", "maxSide": 48, "maxPixels": 48 * 40, "gap": 3}
    for direction in ("vertical", "horizontal"):
        for mode in ("min", "original"):
            out, ref = _check(px, direction, dict(caps, filter=filt, mode=mode, edgeAA=True))
            plain = ist.stitch(U.hip_images(px), direction, dict(caps, filter=filt, mode=mode, edgeAA=False))["data"]
            assert not np.array_equal(plain, out), "the plan has fractional edges: AA must change some pixels"
    # superSample 2.6 / 2.2 on small inputs (reference default for n < 7), with EXIF turns on top
    small = [U.smooth_image(180 + i, 21, 33) for i in range(3)]
    for plat in ("devtools", "ios"):
        _check(small, "vertical", {"filter": filt, "platform": plat, "edgeAA": True})
        _check(small, "horizontal", {"filter": filt, "platform": plat, "edgeAA": True, "gap": 2}, orientations=[6, 1, 3])
    # integer-edged plans are unaffected
    a = ist.stitch(U.hip_images(px), "vertical", {"filter": filt, "gap": 4, "edgeAA": True})["data"]
    b = ist.stitch(U.hip_images(px), "vertical", {"filter": filt, "gap": 4})["data"]
    assert np.array_equal(a, b)


def test_edge_antialiasing_known_answer():
    """one black 2x2 image drawn at y = 0.25 .. 2.25 on white (scaleDown makes the cursor fractional): rows get
    coverage 0.75, 1, 0.25 -> 255*(1-cov)."""
    import ctypes as C
    from imagestitching_amd import _lib as L
    img = np.zeros((2, 2, 4), np.uint8)
    img[..., 3] = 255
    ops = (L.Op * 2)()
    ops[0].kind = 0; ops[0].m[:] = [1, 0, 0, 1, 0, 0]; ops[0].d[:] = [0, 0, 2, 3]; ops[0].rgba[:] = [255, 255, 255, 255]
    ops[1].kind = 1; ops[1].image = 0; ops[1].m[:] = [1, 0, 0, 1, 0, 0]; ops[1].s[:] = [0, 0, 2, 2]; ops[1].d[:] = [0, 0.25, 2, 2]
    descs = (L.ImageDesc * 1)(L.ImageDesc(2, 2, 1, 0, 0, 0, 0))
    out = np.zeros((3, 2, 4), np.uint8)
    ptrs = (C.c_void_p * 1)(img.ctypes.data)
    pit = (C.c_size_t * 1)(8)
    clear = (C.c_uint8 * 4)(0, 0, 0, 0)
    from imagestitching_amd.stitch import _ctx
    for filt in (0, 1):
        L.check(L.lib.ist_render_rgba8(_ctx(0), 2, 3, clear, ops, 2, descs, ptrs, pit, 1, filt | 0x100, None, out.ctypes.data, 8))
        assert out[:, 0, 0].tolist() == [64, 0, 191], out[:, 0, 0].tolist()      # floor(255*0.25+0.5), 0, floor(255*0.75+0.5)
        assert (out[..., 3] == 255).all()


@pytest.mark.parametrize("direction", ["vertical", "horizontal"])
def test_area_filter_matches_the_oracle(direction):
    """IST_FILTER_AREA: minified draws are box-averaged (fp32 sums on the GPU, fp64 in the oracle: +-1 LSB); draws that do not
    shrink take the bilinear fast paths and are identical to filter 'bilinear'."""
    px = [U.rand_image(900, 300, 400), U.smooth_image(901, 130, 120), U.rand_image(902, 90, 700, opaque=False), U.rand_image(903, 66, 120)]
    for opts, ori in (({"filter": "area", "mode": "min"}, None), ({"filter": "area", "mode": "min", "gap": 3}, [6, 1, 3, 8]),
                      ({"filter": "area", "platform": "android", "maxSide": 256, "maxPixels": 65536}, None),          # ~6x shrink, edge AA on
                      ({"filter": "area", "platform": "ios", "superSample": 1, "maxSide": 200}, [2, 5, 1, 7])):
        out, ref = _check(px, direction, opts, orientations=ori)
    up = [U.rand_image(904, 40, 60), U.rand_image(905, 90, 120)]
    a = ist.stitch(U.hip_images(up), direction, {"filter": "area", "mode": "max"})["data"]
    b = ist.stitch(U.hip_images(up), direction, {"filter": "bilinear", "mode": "max"})["data"]
    assert np.array_equal(a, b)
    big = [U.smooth_image(906, 3024 // 4, 4032 // 4) for _ in range(3)]                 # integer 3x shrink: the block mean
    got = ist.stitch(U.hip_images(big), direction, {"filter": "area", "maxSide": 1008 // 3 if direction == "vertical" else 1008, "platform": "ios", "superSample": 1, "edgeAA": False})
    ref, _, _ = U.oracle_stitch(big, direction, {"filter": "area", "maxSide": 1008 // 3 if direction == "vertical" else 1008, "platform": "ios", "superSample": 1, "edgeAA": False})
    assert U.max_abs_diff(got["data"], ref) <= 1


def test_random_op_lists_against_oracle():
    """Differential test of the op-list surface (what the Canvas shim records): fills, draws under all eight
    axis-aligned transforms at random scales, source rectangles that leave the bitmap, overlapping and translucent
    draws, opaque or transparent canvases, all three coverage rules.  Nearest must be exact, bilinear within 1 LSB."""
    import ctypes as C
    from imagestitching_amd import _lib as L
    from imagestitching_amd.stitch import _ctx
    rng = np.random.default_rng(777)
    worst = 0
    for case in range(250):
        cw, ch = int(rng.integers(8, 300)), int(rng.integers(8, 300))
        n_img = int(rng.integers(1, 4))
        px = [U.rand_image(900 + 7 * case + k, int(rng.integers(2, 120)), int(rng.integers(2, 120)), opaque=bool(rng.integers(0, 2))) for k in range(n_img)]
        descs_o = [{"width": a.shape[1], "height": a.shape[0]} for a in px]
        ops_o = []
        if rng.integers(0, 2):
            ops_o.append({"kind": "fill", "m": [1, 0, 0, 1, 0, 0], "rect": [0, 0, cw, ch], "rgba": tuple(int(v) for v in rng.integers(0, 256, 3)) + (255,)})
        for _ in range(int(rng.integers(1, 6))):
            k = int(rng.integers(0, n_img))
            h, w = px[k].shape[:2]
            sc = 1.0 if rng.integers(0, 3) == 0 else float(rng.uniform(0.3, 3.0))
            t = int(rng.integers(0, 8))
            sx, sy = (-sc if t & 1 else sc), (-sc if t & 2 else sc)
            e, f = float(rng.integers(0, cw)), float(rng.integers(0, ch))
            if rng.integers(0, 2):
                e += float(rng.uniform(0, 1)); f += float(rng.uniform(0, 1))
            m = [0, sx, sy, 0, e, f] if t & 4 else [sx, 0, 0, sy, e, f]
            if rng.integers(0, 2):
                s = [0, 0, w, h]
            else:
                s = [float(rng.uniform(-5, w / 2)), float(rng.uniform(-5, h / 2)), float(rng.uniform(1, w)), float(rng.uniform(1, h))]
            d = [float(rng.uniform(-20, 20)), float(rng.uniform(-20, 20)), float(rng.uniform(4, 150)), float(rng.uniform(4, 150))]
            if rng.integers(0, 3) == 0:
                d = [round(v) for v in d]
            ops_o.append({"kind": "draw", "image": k, "m": m, "s": s, "d": d})
        clear = (0, 0, 0, 0) if rng.integers(0, 2) else tuple(int(v) for v in rng.integers(0, 256, 3)) + (255,)
        filt = "nearest" if rng.integers(0, 2) else "bilinear"
        aa = bool(rng.integers(0, 3) == 0)
        ref = O.render_ops(cw, ch, ops_o, descs_o, px, filt, clear=clear, edge_aa=aa)
        ops = (L.Op * len(ops_o))()
        for i, o in enumerate(ops_o):
            ops[i].m[:] = o["m"]
            if o["kind"] == "fill":
                ops[i].kind = 0; ops[i].image = -1; ops[i].d[:] = o["rect"]; ops[i].rgba[:] = o["rgba"]
            else:
                ops[i].kind = 1; ops[i].image = o["image"]; ops[i].s[:] = o["s"]; ops[i].d[:] = o["d"]
        descs = (L.ImageDesc * n_img)(*[L.ImageDesc(a.shape[1], a.shape[0], 1, 0, 0, 0, 0) for a in px])
        ptrs = (C.c_void_p * n_img)(*[a.ctypes.data for a in px])
        pit = (C.c_size_t * n_img)(*[a.strides[0] for a in px])
        out = np.zeros((ch, cw, 4), np.uint8)
        f = {"nearest": 0, "bilinear": 1}[filt] | (0x100 if aa else 0)
        L.check(L.lib.ist_render_rgba8(_ctx(0), cw, ch, (C.c_uint8 * 4)(*clear), ops, len(ops_o), descs, ptrs, pit, n_img, f, None, out.ctypes.data, out.strides[0]))
        d = np.abs(out.astype(np.int16) - ref.astype(np.int16))
        tol = 0 if filt == "nearest" and not aa else 1
        solid = ref[..., 3] == 255                     # what the stitch path produces (the canvas is filled white first)
        assert d[..., 3].max() <= tol, (case, filt, aa, ops_o)
        if solid.any():
            worst = max(worst, int(d[solid].max()))
            assert d[solid].max() <= tol, (case, filt, aa, int(d[solid].max()), ops_o)
        # a translucent pixel of a transparent canvas reads back un-premultiplied (c * 255 / a): one LSB of the
        # premultiplied value becomes up to 255 / a LSBs of the colour
        soft = ~solid & (ref[..., 3] > 0)
        if soft.any() and tol:
            lim = tol + np.ceil(255.0 / ref[..., 3][soft].astype(np.float64))
            assert (d[..., :3][soft].max(axis=-1) <= lim).all(), (case, filt, aa, ops_o)
        elif soft.any():
            assert d[soft].max() == 0, (case, filt, aa, ops_o)
    assert worst <= 1


@pytest.mark.parametrize("opts", [{"platform": "ios", "superSample": 1}, {"platform": "android", "superSample": 1}, {"maxSide": 6804}])
def test_area_filter_fast_path_on_the_phone_capped_plans_at_full_size(opts):
    """filter 'area' on the plans the reference really renders (9 x 12 MP shrunk 2.2x on iOS, 6.6x on
    Android, 4x): the streamed box filter (PATH_AREA_STREAM: no per-pixel general tiles), within 1 LSB of the oracle.  Opaque
    photos and - one image - translucent pixels (the premultiplied sums)."""
    px = [U.rand_image(950 + i, 3024, 4032, opaque=(i != 4)) for i in range(9)]
    o = dict(opts, filter="area", edgeAA=False)
    imgs = [{"width": 4032, "height": 3024, "data": a, "opaque": i != 4} for i, a in enumerate(px)]
    st = ist.Stitcher(0)
    p, job = st.compile(imgs, "vertical", o)
    assert job.info["tiles_general"] == 0 and job.info["tiles_sample"] > 0, job.info
    got = ist.stitch(imgs, "vertical", o)
    ref, _, _ = U.oracle_stitch(px, "vertical", o, threads=16)
    assert got["data"].shape == ref.shape
    assert U.max_abs_diff(got["data"], ref) <= 1
