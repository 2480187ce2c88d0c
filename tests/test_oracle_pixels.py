"""Pins the CPU oracle's pixel arithmetic (CPU only): committed fixtures minted by oracle/witness_check.py (inputs,
oracle outputs, and outputs of three independent witnesses: cairo/pixman, torch interpolate, PIL affine), the probe
row quoted in SURVEY.md section 8c, and domain properties (identity, flips, compositing known answers)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = np.load(os.path.join(ROOT, "tests", "golden", "pixel_fixtures.npz"))
KEYS = sorted({k.split("__")[0] for k in FIX.files})


def _draw(img, dw, dh, filt):
    sh, sw = img.shape[:2]
    ops = [{"kind": "fill", "m": [1, 0, 0, 1, 0, 0], "rect": [0, 0, dw, dh], "rgba": (255, 255, 255, 255)},
           {"kind": "draw", "image": 0, "m": [1, 0, 0, 1, 0, 0], "s": [0, 0, sw, sh], "d": [0, 0, dw, dh]}]
    return O.render_ops(dw, dh, ops, [{"width": sw, "height": sh}], [img], filt)


@pytest.mark.parametrize("key", KEYS)
def test_oracle_reproduces_committed_fixture(key):
    img, want = FIX[key + "__in"], FIX[key + "__oracle"]
    filt = "bilinear" if key.endswith("bilinear") else "nearest"
    got = _draw(img, want.shape[1], want.shape[0], filt)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("key", KEYS)
def test_witnesses_agree_with_oracle(key):
    """bilinear: torch (fp32 exact-weight) and PIL within 1 LSB; cairo/pixman within 3 (its weights are 7-bit).
    nearest: all three identical except where the sample point lies exactly on a source pixel edge (a tie)."""
    ora = FIX[key + "__oracle"][..., :3].astype(int)
    if key.endswith("bilinear"):
        for w, tol in (("torch", 1), ("pil", 1), ("cairo", 3)):
            assert np.abs(FIX[key + "__" + w][..., :3].astype(int) - ora).max() <= tol, w
    else:
        nt = FIX[key + "__not_tie"]
        assert np.array_equal(FIX[key + "__torch"][..., :3], FIX[key + "__oracle"][..., :3])   # nearest-exact: identical everywhere
        for w in ("pil", "cairo"):
            assert np.array_equal(FIX[key + "__" + w][..., :3][nt], FIX[key + "__oracle"][..., :3][nt]), w


def test_survey_probe_row():
    """row [0,100,200,50] -> 8 px."""
    row = np.zeros((1, 4, 4), np.uint8)
    row[0, :, :3] = np.array([0, 100, 200, 50])[:, None]
    row[..., 3] = 255
    assert _draw(row, 8, 1, "nearest")[0, :, 0].tolist() == [0, 0, 100, 100, 200, 200, 50, 50]
    assert _draw(row, 8, 1, "bilinear")[0, :, 0].tolist() == [0, 25, 75, 125, 175, 163, 88, 50]


@pytest.mark.parametrize("filt", ["nearest", "bilinear"])
def test_uniform_sizes_reduce_to_concatenation(filt):
    px = [U.rand_image(k, 48, 64) for k in range(3)]
    out, pd, _ = O.stitch(px, "vertical", filter=filt)
    assert np.array_equal(out, np.concatenate(px, 0))
    out, pd, _ = O.stitch(px, "horizontal", filter=filt)
    assert np.array_equal(out, np.concatenate(px, 1))


def test_threads_do_not_change_the_result():
    px = [U.rand_image(k, h, w, opaque=False) for k, (w, h) in enumerate([(90, 60), (61, 77), (40, 40)])]
    a, _, _ = O.stitch(px, "vertical", "max", 5, filter="bilinear", threads=1)
    b, _, _ = O.stitch(px, "vertical", "max", 5, filter="bilinear", threads=7)
    assert np.array_equal(a, b)


def test_orientations_are_index_remaps_at_unit_scale():
    """EXIF 2/3/4 at 1:1 are flips of the plain draw (utils/canvas.js:161-175); 5/6/8 transpose-type remaps of the
    bitmap into the SAME rect (stretching when the rect is not the transposed size)."""
    a = U.rand_image(5, 40, 40)
    base, _, _ = O.stitch([a], "vertical", filter="nearest", orientations=[1])
    for o, f in ((2, lambda x: x[:, ::-1]), (3, lambda x: x[::-1, ::-1]), (4, lambda x: x[::-1]),
                 (5, lambda x: x.transpose(1, 0, 2)), (6, lambda x: x.transpose(1, 0, 2)[:, ::-1]), (8, lambda x: x.transpose(1, 0, 2)[::-1])):
        got, _, _ = O.stitch([a], "vertical", filter="nearest", orientations=[o])
        assert np.array_equal(got, f(base)), o
    # orientation 7: the reference translates to (dx+dw, dy) where (dx+dw, dy+dh) would be needed, so the image lands
    # one rect-height above its rect: a single image is drawn entirely off-canvas and the canvas stays white
    got, _, _ = O.stitch([a], "vertical", filter="nearest", orientations=[7])
    assert (got == 255).all()


def test_source_over_white_known_answers():
    a = np.zeros((1, 3, 4), np.uint8)
    a[0, 0] = (10, 20, 30, 0)        # fully transparent -> white
    a[0, 1] = (10, 20, 30, 255)      # opaque -> itself
    a[0, 2] = (10, 20, 30, 128)      # (c*128 + 255*127 + 127) // 255
    for filt in ("nearest", "bilinear"):
        out, _, _ = O.stitch([a], "vertical", filter=filt)
        assert out[0, 0].tolist() == [255, 255, 255, 255]
        assert out[0, 1].tolist() == [10, 20, 30, 255]
        assert out[0, 2].tolist() == [(10 * 128 + 255 * 127 + 127) // 255, (20 * 128 + 255 * 127 + 127) // 255, (30 * 128 + 255 * 127 + 127) // 255, 255]


def test_gap_and_rounding_slack_stay_white():
    """iOS-capped 9 x 12 MP plan leaves 3 white rows; here a small analogue + gaps."""
    px = [np.zeros((30, 40, 4), np.uint8) for _ in range(3)]
    for p in px:
        p[..., 3] = 255
    out, pd, rl = O.stitch(px, "vertical", "min", 6, filter="nearest")
    assert out.shape == (102, 40, 4)
    assert (out[30:36] == 255).all() and (out[66:72] == 255).all() and (out[0:30, :, :3] == 0).all()


def test_area_filter_is_the_exact_block_mean_at_integer_ratios_and_bilinear_when_nothing_shrinks():
    """IST_FILTER_AREA (opt-in reading of imageSmoothingQuality 'high', index.js:1419): per minified axis a box of width |k|;
    at integer shrink factors that is the mean of k x k blocks (an independent closed form), PIL's BOX resize agrees within 1."""
    from PIL import Image
    rng = np.random.default_rng(0)
    for k in (2, 3, 5):
        a = rng.integers(0, 256, (60 * k, 80 * k, 4), dtype=np.uint8)
        a[..., 3] = 255
        b = rng.integers(0, 256, (10, 80, 4), dtype=np.uint8)
        b[..., 3] = 255
        out, pd, rl = O.stitch([a, b], "vertical", "min", 0, O.lifted_limits(1.0), "area")
        ref = np.floor(a.reshape(60, k, 80, k, 4).astype(np.float64).mean((1, 3)) + 0.5)
        assert np.array_equal(out[:60], ref.astype(np.uint8))
        assert np.array_equal(out[60:], b)                                   # the 1:1 image is untouched
        box = np.asarray(Image.fromarray(a).resize((80, 60), Image.BOX))
        assert np.abs(out[:60].astype(int) - box.astype(int)).max() <= 1
    a = rng.integers(0, 256, (30, 40, 4), dtype=np.uint8)
    b = rng.integers(0, 256, (45, 60, 4), dtype=np.uint8)
    up_area, _, _ = O.stitch([a, b], "vertical", "max", 3, O.lifted_limits(1.0), "area")
    up_bil, _, _ = O.stitch([a, b], "vertical", "max", 3, O.lifted_limits(1.0), "bilinear")
    assert np.array_equal(up_area, up_bil)


def test_hosts_turn_edge_antialiasing_on_for_reference_platform_plans():
    from imagestitching_amd.stitch import _filter_of, _merge, edge_aa_of
    assert not edge_aa_of(_merge(None)) and not edge_aa_of(_merge({"gap": 3}))
    assert edge_aa_of(_merge({"platform": "ios"})) and edge_aa_of(_merge({"platform": "devtools"}))
    assert not edge_aa_of(_merge({"platform": "ios", "edgeAA": False})) and edge_aa_of(_merge({"edgeAA": True}))
    assert _filter_of(_merge({"platform": "android", "filter": "area"})) == (2 | 0x100)
    assert _filter_of(_merge({"filter": "nearest"})) == 0
