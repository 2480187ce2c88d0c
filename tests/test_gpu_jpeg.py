"""JPEG decode: Huffman decoding on the host + IDCT / fancy upsampling / colour conversion
on the GPU, against PIL (libjpeg-turbo, slow integer IDCT, fancy upsampling: the defaults).  The arithmetic is meant to
be the same, so the bar is bit-exact; a tolerance would hide a wrong rounding constant."""
import io

import numpy as np
import pytest
from PIL import Image

import imagestitching_amd as ist
from oracle import oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu


def _jpeg(a, **kw):
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", **kw)
    return b.getvalue()


def _pil(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"))


def _photo(seed, h, w):
    """smooth content + some texture: what JPEG is for (pure noise at quality 75 is mostly clamping)."""
    a = U.smooth_image(seed, h, w)[..., :3].astype(np.int32)
    a += np.random.default_rng(seed).integers(-20, 21, a.shape)
    return np.clip(a, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("subsampling", [0, 1, 2])
@pytest.mark.parametrize("size", [(64, 48), (67, 45), (1, 1), (8, 8), (17, 33), (250, 131)])
def test_jpeg_matches_pil(subsampling, size):
    w, h = size
    for q, seed in ((90, 1), (50, 2)):
        data = _jpeg(_photo(seed, h, w), quality=q, subsampling=subsampling)
        got = ist.decode_image(data)
        ref = _pil(data)
        assert got.shape == ref.shape
        d = np.abs(got.astype(int) - ref.astype(int))
        assert d.max() == 0, "subsampling %d size %s q%d: max diff %d, %d px differ" % (subsampling, size, q, d.max(), (d.max(-1) > 0).sum())


def test_jpeg_greyscale_noise_and_optimised_tables():
    g = np.random.default_rng(3).integers(0, 256, (37, 53), dtype=np.uint8)
    for kw in ({"quality": 95}, {"quality": 30, "optimize": True}):
        data = _jpeg(g, **kw)
        assert np.array_equal(ist.decode_image(data), _pil(data))
    noise = np.random.default_rng(4).integers(0, 256, (40, 56, 3), dtype=np.uint8)
    data = _jpeg(noise, quality=100, subsampling=2, optimize=True)
    assert np.array_equal(ist.decode_image(data), _pil(data))


def test_jpeg_restart_intervals():
    a = _photo(5, 120, 200)
    for kw in ({"restart_marker_blocks": 3}, {"restart_marker_rows": 1}):
        data = _jpeg(a, quality=80, subsampling=2, **kw)
        assert b"\xff\xdd" in data
        assert np.array_equal(ist.decode_image(data), _pil(data))


def test_jpeg_exif_orientation_feeds_the_planner(tmp_path):
    a = _photo(6, 48, 64)
    exif = Image.Exif()
    exif[0x0112] = 6
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", quality=92, exif=exif)
    assert ist.image_info(b.getvalue()) == (64, 48, 6)
    p = tmp_path / "rot.jpg"
    p.write_bytes(b.getvalue())
    plain = tmp_path / "plain.jpg"
    plain.write_bytes(_jpeg(_photo(7, 64, 48), quality=92))
    res = ist.stitch_files([str(p), str(plain)], "vertical", {"filter": "nearest"})
    bitmaps = [_pil(b.getvalue()), _pil(plain.read_bytes())]
    ref, pd, _ = U.oracle_stitch(bitmaps, "vertical", {"filter": "nearest"}, orientations=[6, 1])
    assert np.array_equal(ist.decode_png(res["png"]), ref)


def test_full_size_photo_decode():
    """a 12 MP 4:2:0 JPEG (what a phone camera writes)."""
    a = _photo(8, 3024, 4032)
    data = _jpeg(a, quality=88, subsampling=2)
    got = ist.decode_image(data)
    assert np.array_equal(got, _pil(data))


@pytest.mark.parametrize("subsampling", [0, 1, 2])
def test_progressive_jpeg_matches_pil(subsampling):
    """SOF2: spectral selection + successive approximation scans (host), same GPU reconstruction."""
    a = _photo(20 + subsampling, 203, 317)
    data = _jpeg(a, quality=83, subsampling=subsampling, progressive=True)
    assert b"\xff\xc2" in data
    assert np.array_equal(ist.decode_image(data), _pil(data))
    g = io.BytesIO()
    Image.fromarray(a).convert("L").save(g, "JPEG", quality=70, progressive=True)
    assert np.array_equal(ist.decode_image(g.getvalue()), _pil(g.getvalue()))


def test_jpeg_unsupported_and_damaged_files():
    a = _photo(9, 40, 40)
    # luma sampling factors beyond 2 (true 4:1:1): patch the frame header of a 4:4:4 file (Y: 0x11 -> 0x41)
    raw = bytearray(_jpeg(a, quality=80, subsampling=0))
    sof = raw.find(b"\xff\xc0")
    assert raw[sof + 11] == 0x11
    raw[sof + 11] = 0x41
    with pytest.raises(ist.StitchError) as e:
        ist.decode_image(bytes(raw))
    assert e.value.code == -7
    good = _jpeg(a, quality=80)
    with pytest.raises(ist.StitchError) as e:
        ist.decode_image(good[:len(good) // 3])
    assert e.value.code == -6


@pytest.mark.parametrize("subsampling,quality,optimize", [(0, 92, False), (1, 85, True), (2, 75, False), (2, 96, True), (2, 30, False)])
def test_gpu_entropy_decoder_in_the_file_pipeline(tmp_path, subsampling, quality, optimize):
    """stitch_files decodes baseline JPEGs with the GPU Huffman decoder (ist_jpeg_gpu.hip): same-width images stitched
    vertically with the nearest filter are a pure concatenation, so the PNG must hold PIL's pixels bit for bit."""
    paths, want = [], []
    for k, (h, w) in enumerate([(203, 317), (64, 317), (411, 317), (8, 317)]):
        a = _photo(40 + 5 * k + subsampling, h, w)
        data = _jpeg(a, quality=quality, subsampling=subsampling, optimize=optimize)
        p = tmp_path / ("g%d.jpg" % k)
        p.write_bytes(data)
        paths.append(str(p))
        want.append(_pil(data))
    grey = io.BytesIO()
    Image.fromarray(_photo(77, 100, 317)).convert("L").save(grey, "JPEG", quality=80)
    (tmp_path / "grey.jpg").write_bytes(grey.getvalue())
    paths.append(str(tmp_path / "grey.jpg"))
    want.append(_pil(grey.getvalue()))
    res = ist.stitch_files(paths, "vertical", {"filter": "nearest"})
    assert np.array_equal(ist.decode_png(res["png"]), np.concatenate(want, 0))


def test_gpu_entropy_decoder_full_size_and_mixed_inputs(tmp_path):
    """a 12 MP 4:2:0 photo (15 000 subsequences), a progressive file and a PNG in one call: GPU Huffman, host Huffman and
    host inflate side by side"""
    a = _photo(91, 3024, 4032)
    base = _jpeg(a, quality=88, subsampling=2)
    prog = _jpeg(_photo(92, 240, 4032), quality=80, subsampling=2, progressive=True)
    (tmp_path / "a.jpg").write_bytes(base)
    (tmp_path / "b.jpg").write_bytes(prog)
    Image.fromarray(_photo(93, 100, 4032)).save(tmp_path / "c.png")
    rst = _jpeg(_photo(94, 64, 4032), quality=85, subsampling=2, restart_marker_blocks=7)      # restart intervals: each one its own stream on the GPU
    assert b"\xff\xdd" in rst
    (tmp_path / "d.jpg").write_bytes(rst)
    res = ist.stitch_files([str(tmp_path / "a.jpg"), str(tmp_path / "b.jpg"), str(tmp_path / "c.png"), str(tmp_path / "d.jpg")], "vertical", {"filter": "nearest"})
    want = np.concatenate([_pil(base), _pil(prog), np.asarray(Image.open(tmp_path / "c.png").convert("RGBA")), _pil(rst)], 0)
    assert np.array_equal(ist.decode_png(res["png"]), want)


def _gpu_files():
    from imagestitching_amd import _lib as L
    return L.lib.ist_debug_gpu_entropy_files()


@pytest.mark.parametrize("subsampling", [0, 1, 2])
@pytest.mark.parametrize("restart", [{"restart_marker_rows": 1}, {"restart_marker_blocks": 2}, {"restart_marker_blocks": 5}, {"restart_marker_rows": 3}])
def test_gpu_entropy_decoder_takes_restart_intervals(tmp_path, subsampling, restart):
    """Files with DRI: every restart interval is an independent stream (T.81 E.1.4: byte aligned, DC predictors reset), so
    the GPU decoder takes each as a unit of its batch writing into the same planes at the interval's first MCU.  The
    counter proves the GPU path decoded them (the host fall-back would give the same pixels)."""
    paths, want = [], []
    for k, (h, w) in enumerate([(203, 317), (64, 317), (16, 317), (411, 317)]):
        data = _jpeg(_photo(140 + 5 * k + subsampling, h, w), quality=85, subsampling=subsampling, **restart)
        assert b"\xff\xdd" in data and (h <= 16 or b"\xff\xd0" in data)
        p = tmp_path / ("r%d.jpg" % k)
        p.write_bytes(data)
        paths.append(str(p))
        want.append(_pil(data))
    grey = io.BytesIO()
    Image.fromarray(_photo(177, 100, 317)).convert("L").save(grey, "JPEG", quality=80, **restart)
    (tmp_path / "grey.jpg").write_bytes(grey.getvalue())
    paths.append(str(tmp_path / "grey.jpg"))
    want.append(_pil(grey.getvalue()))
    before = _gpu_files()
    res = ist.stitch_files(paths, "vertical", {"filter": "nearest"})
    assert np.array_equal(ist.decode_png(res["png"]), np.concatenate(want, 0))
    assert _gpu_files() - before == len(paths)


def test_gpu_entropy_decoder_restart_intervals_full_size_and_too_many(tmp_path):
    """a 12 MP photo with one interval per MCU row (189 units, 15 000 subsequences) on the GPU; the same photo with an
    interval per MCU (47 628 of them) is left to the host decoder, and both give PIL's pixels"""
    a = _photo(191, 3024, 4032)
    rows = _jpeg(a, quality=88, subsampling=2, restart_marker_rows=1)
    every = _jpeg(a[:1024], quality=88, subsampling=2, restart_marker_blocks=1)
    (tmp_path / "rows.jpg").write_bytes(rows)
    (tmp_path / "every.jpg").write_bytes(every)
    before = _gpu_files()
    res = ist.stitch_files([str(tmp_path / "rows.jpg"), str(tmp_path / "every.jpg")], "vertical", {"filter": "nearest"})
    assert np.array_equal(ist.decode_png(res["png"]), np.concatenate([_pil(rows), _pil(every)], 0))
    assert _gpu_files() - before == 1


def test_stitch_files_names_the_file_it_cannot_open(tmp_path):
    """ist_stitch_paths_png reads the files itself: a missing or empty one fails the call like a file that does not decode -
    the reference's message with the image's index (index.js:1512-1514)."""
    good = tmp_path / "good.jpg"
    good.write_bytes(_jpeg(_photo(1, 40, 56), quality=80))
    (tmp_path / "empty.jpg").write_bytes(b"")
    for bad, what in ((tmp_path / "missing.jpg", "cannot open"), (tmp_path / "empty.jpg", "empty")):
        with pytest.raises(ist.StitchError) as e:
            ist.stitch_files([str(good), str(bad)], "vertical")
        assert "图片1解码异常" in str(e.value) and what in str(e.value)
    res = ist.stitch_files([str(good), str(good)], "vertical", {"filter": "nearest"})
    assert res["height"] == 80 and res["width"] == 56


def test_paths_are_read_not_mapped_and_a_rewritten_file_is_the_new_file(tmp_path):
    """ist_stitch_paths_png reads every file into a block its context keeps from call to call.  A path
    rewritten between calls (larger, smaller, other sampling, another type) gives the new file's pixels - nothing of the
    block's earlier contents; a directory is refused like a file that cannot be read."""
    p, q = tmp_path / "a.jpg", tmp_path / "b.jpg"
    q.write_bytes(_jpeg(_photo(31, 64, 96), quality=85, subsampling=2))
    versions = [_jpeg(_photo(32, 200, 96), quality=95, subsampling=0), _jpeg(_photo(33, 24, 96), quality=40, subsampling=2),
                _jpeg(_photo(34, 120, 96), quality=80, subsampling=1, restart_marker_rows=1)]
    b = io.BytesIO()
    Image.fromarray(_photo(35, 50, 96)).save(b, "PNG")
    versions.append(b.getvalue())
    for data in versions:
        p.write_bytes(data)
        res = ist.stitch_files([str(p), str(q)], "vertical", {"filter": "nearest"})
        want = np.concatenate([_pil(data), _pil(q.read_bytes())], 0)
        assert np.array_equal(ist.decode_png(res["png"]), want)
    with pytest.raises(ist.StitchError) as e:
        ist.stitch_files([str(q), str(tmp_path)], "vertical")
    assert "图片1解码异常" in str(e.value)


def test_a_frame_layout_that_differs_on_the_second_read_is_refused():
    """the device arena is sized from the header-only parse; the worker's second parse must see the same
    sampling factors and block counts, or the Huffman write kernel / IDCT would run past the planes (4:2:0 -> 4:4:4 doubles
    the chroma blocks).  IST_TUNING=1 IST_JPEG_SECOND_READ_444=1 shows the second read other luma sampling factors."""
    import os
    import subprocess
    import sys
    code = """
import sys, io
sys.path.insert(0, %r)
import numpy as np
from PIL import Image
import imagestitching_amd as ist
a = (np.add.outer(np.arange(160), np.arange(240))[..., None] * np.array([1, 2, 3]) %% 256).astype(np.uint8)
b = io.BytesIO(); Image.fromarray(a).save(b, "JPEG", quality=85, subsampling=2)
open(sys.argv[1], "wb").write(b.getvalue())
try:
    ist.stitch_files([sys.argv[1]], "vertical")
    print("accepted")
except ist.StitchError as e:
    print("refused:", e)
b = io.BytesIO(); Image.fromarray(a).save(b, "JPEG", quality=85, subsampling=0)      # already 4:4:4: the flip changes nothing
open(sys.argv[1], "wb").write(b.getvalue())
res = ist.stitch_files([sys.argv[1]], "vertical")
assert np.array_equal(ist.decode_png(res["png"]), np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGBA")))
print("444 ok")
""" % (U.ROOT,)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, IST_TUNING="1", IST_JPEG_SECOND_READ_444="1")
        r = subprocess.run([sys.executable, "-c", code, os.path.join(d, "x.jpg")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "refused:" in r.stdout and "changed between two reads" in r.stdout and "444 ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_more_restart_intervals_than_a_batch_holds(tmp_path):
    """The Huffman batch addresses its units with 16 bits.  33 files of 2016 intervals each + 2 plain ones are 66 530 units: the
    batch leaves as many of the interval-heavy files to the host decoder as it takes to fit (one), and every file still decodes
    to PIL's pixels."""
    paths, want = [], []
    base = _photo(400, 1008, 512)
    dri = _jpeg(base, quality=70, subsampling=2, restart_marker_blocks=1)           # 32 x 63 MCUs, one interval each
    plain = _jpeg(base, quality=70, subsampling=2)
    for k in range(35):
        data = plain if k in (0, 20) else dri
        p = tmp_path / ("b%d.jpg" % k)
        p.write_bytes(data)
        paths.append(str(p))
    before = _gpu_files()
    res = ist.stitch_files(paths, "horizontal", {"filter": "nearest"})
    assert _gpu_files() - before == 34
    got = ist.decode_png(res["png"])
    a, b = _pil(plain), _pil(dri)
    for k in range(35):
        assert np.array_equal(got[:, 512 * k:512 * (k + 1)], a if k in (0, 20) else b), k


def test_gpu_entropy_decoder_damaged_restart_markers_go_to_the_host(tmp_path):
    """a marker missing, a marker out of sequence, one too many, and a truncated interval: never the GPU path's pixels
    unless they are the host decoder's too"""
    good = _jpeg(_photo(195, 120, 160), quality=85, subsampling=2, restart_marker_rows=1)
    marks = [i for i in range(good.find(b"\xff\xda"), len(good) - 1) if good[i] == 0xFF and 0xD0 <= good[i + 1] <= 0xD7]
    assert len(marks) == 7
    cases = {}
    cases["missing"] = good[:marks[2]] + good[marks[2] + 2:]
    swapped = bytearray(good); swapped[marks[1] + 1], swapped[marks[2] + 1] = swapped[marks[2] + 1], swapped[marks[1] + 1]
    cases["out_of_sequence"] = bytes(swapped)
    cases["one_too_many"] = good[:-2] + b"\xff\xd7" + good[marks[0] + 2:marks[1]] + good[-2:]
    cases["short_interval"] = good[:marks[3] + 2] + good[marks[3] + 12:]
    cases["empty_interval"] = good[:marks[2] + 2] + good[marks[3]:]                      # two markers back to back
    cases["marker_at_the_end"] = good[:-2] + b"\xff\xd7" + good[-2:]                     # one more RSTn than intervals, no data behind it
    for name, data in cases.items():
        p = tmp_path / (name + ".jpg")
        p.write_bytes(data)
        try:
            host = ist.decode_image(data)
        except ist.StitchError:
            host = None
        try:
            res = ist.stitch_files([str(p)], "vertical", {"filter": "nearest"})
        except ist.StitchError as e:
            assert host is None, name
            assert "解码异常" in str(e)
        else:
            assert host is not None, name
            assert np.array_equal(ist.decode_png(res["png"]), host), name


def test_a_scan_that_ends_after_too_few_restart_intervals_never_shows_an_earlier_calls_image(tmp_path):
    """a DRI scan cut at an RSTn boundary (k < n intervals, EOI appended) passes a PER-INTERVAL block count;
    the coefficient planes live in the context's grow-only arena, so the rows of the missing intervals would be reconstructed
    from what the previous call left there.  The parser now hands such a file to the host decoder: the result is the host
    decoder's (an error, or its pixels), never the GPU path's, and it does not depend on what was decoded before."""
    a, b = _photo(401, 160, 240), 255 - _photo(402, 160, 240)
    first = tmp_path / "first.jpg"
    first.write_bytes(_jpeg(b, quality=90, subsampling=2, restart_marker_rows=1))
    good = _jpeg(a, quality=90, subsampling=2, restart_marker_rows=1)
    marks = [i for i in range(good.find(b"\xff\xda"), len(good) - 1) if good[i] == 0xFF and 0xD0 <= good[i + 1] <= 0xD7]
    assert len(marks) == 9                                                # ten MCU rows of 16
    for cut in (marks[3], marks[3] + 2, marks[8]):                        # in front of an RSTn, right behind it, the last one
        data = good[:cut] + b"\xff\xd9"
        p = tmp_path / ("cut%d.jpg" % cut)
        p.write_bytes(data)
        try:
            host = ist.decode_image(data)
        except ist.StitchError:
            host = None
        outcomes = []
        for prime in (True, False, True):
            if prime:                                                     # same size, same arena offsets: the stale planes are the other photo's
                before = _gpu_files()
                ist.stitch_files([str(first)], "vertical", {"filter": "nearest"})
                assert _gpu_files() - before == 1
            before = _gpu_files()
            try:
                res = ist.stitch_files([str(p)], "vertical", {"filter": "nearest"})
            except ist.StitchError as e:
                assert host is None and "解码异常" in str(e)
                outcomes.append(None)
            else:
                assert host is not None
                got = ist.decode_png(res["png"])
                assert np.array_equal(got, host)
                outcomes.append(got)
            assert _gpu_files() == before                                 # the GPU entropy decoder did not take the file
        assert all((o is None) == (outcomes[0] is None) for o in outcomes)


@pytest.mark.parametrize("direction", ["vertical", "horizontal"])
def test_images_placed_without_scaling_are_reconstructed_straight_into_the_canvas(tmp_path, direction):
    """mode 'original' keeps every image at its own size: each draw only moves its image (odd offsets, gaps, narrower
    images beside the canvas background), so the file pipeline reconstructs the JPEGs straight into their boxes of the canvas
    (no bitmap, no band launch) - except the turned one, which keeps its bitmap.  Same pixels as the oracle's stitch of PIL's
    bitmaps, and the phase-timed run (which renders every band through the stitch kernel) makes the same file."""
    sizes = [(203, 317), (64, 251), (120, 317), (33, 90), (77, 318)]
    paths, bitmaps, orients = [], [], []
    for k, (h, w) in enumerate(sizes):
        kw = {}
        if k == 3:
            ex = Image.Exif()
            ex[0x0112] = 6
            kw["exif"] = ex
        data = _jpeg(_photo(300 + k, h, w), quality=88, subsampling=[2, 0, 1, 2, 2][k], **kw)
        p = tmp_path / ("o%d.jpg" % k)
        p.write_bytes(data)
        paths.append(str(p))
        bitmaps.append(_pil(data))
        orients.append(6 if k == 3 else 1)
    opts = {"filter": "nearest", "mode": "original", "gap": 7}
    from imagestitching_amd import _lib as L
    before = L.lib.ist_debug_direct_images()
    res = ist.stitch_files(paths, direction, opts)
    assert L.lib.ist_debug_direct_images() - before == 4          # every image but the turned one
    ref, _, _ = U.oracle_stitch(bitmaps, direction, opts, orientations=orients)
    got = ist.decode_png(res["png"])
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)
    ist.set_phase_timing(True)
    try:
        timed = ist.stitch_files(paths, direction, opts)
    finally:
        ist.set_phase_timing(False)
    assert np.array_equal(ist.decode_png(timed["png"]), got)


def test_gpu_entropy_decoder_hands_damaged_scans_to_the_host_decoder(tmp_path):
    a = _photo(95, 120, 160)
    good = _jpeg(a, quality=85, subsampling=2)
    sos = good.find(b"\xff\xda")
    bad = bytearray(good)
    for i in range(sos + 40, len(bad) - 2, 97):           # flip bits all over the scan: block count / codes go wrong
        bad[i] ^= 0x55
        if bad[i] == 0xFF:
            bad[i] = 0x7F
    (tmp_path / "good.jpg").write_bytes(good)
    (tmp_path / "bad.jpg").write_bytes(bytes(bad))
    res = ist.stitch_files([str(tmp_path / "good.jpg")], "vertical", {"filter": "nearest"})
    assert np.array_equal(ist.decode_png(res["png"]), _pil(good))
    try:                                                     # either the host decoder rejects it, or both decoders agree
        res = ist.stitch_files([str(tmp_path / "bad.jpg")], "vertical", {"filter": "nearest"})
    except ist.StitchError as e:
        assert "解码异常" in str(e)
    else:
        assert np.array_equal(ist.decode_png(res["png"]), ist.decode_image(bytes(bad)))


def test_gpu_entropy_decoder_agrees_with_the_host_decoder_on_mutated_scans(tmp_path):
    """Mutation fuzzing on the GPU box: whatever a damaged scan looks like, the file pipeline (GPU Huffman with its
    validation + host fall-back) must behave like the host decoder alone: the same pixels, or an error from both."""
    rng = np.random.default_rng(4242)
    seeds = [_jpeg(_photo(60, 120, 160), quality=85, subsampling=2), _jpeg(_photo(61, 97, 131), quality=60, subsampling=0, optimize=True)]
    outcomes = {"same": 0, "both_fail": 0}
    for case in range(80):
        good = seeds[case % 2]
        sos = good.find(b"\xff\xda")
        start = sos + 14
        bad = bytearray(good)
        kind = case % 4
        if kind == 0:                                   # a few bit flips inside the scan
            for _ in range(int(rng.integers(1, 4))):
                i = int(rng.integers(start, len(bad) - 2)); bad[i] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:                                 # random bytes
            for _ in range(int(rng.integers(1, 6))):
                bad[int(rng.integers(start, len(bad) - 2))] = int(rng.integers(0, 255))
        elif kind == 2:                                 # truncate the scan (keep EOI)
            cut = int(rng.integers(start + 4, len(bad) - 2)); bad = bad[:cut] + b"\xff\xd9"
        else:                                           # garbage inserted in the middle
            at = int(rng.integers(start, len(bad) - 2)); bad[at:at] = bytes(int(v) for v in rng.integers(0, 255, int(rng.integers(1, 40))))
        for i in range(start, len(bad) - 2):            # keep the mutation inside the entropy data: no accidental markers
            if bad[i] == 0xFF and bad[i + 1] != 0x00:
                bad[i] = 0xFE
        p = tmp_path / ("m%d.jpg" % case)
        p.write_bytes(bytes(bad))
        try:
            host = ist.decode_image(bytes(bad))         # host Huffman + GPU reconstruction
        except ist.StitchError:
            host = None
        try:
            pipe = ist.decode_png(ist.stitch_files([str(p)], "vertical", {"filter": "nearest"})["png"])
        except ist.StitchError:
            pipe = None
        if host is None or pipe is None:
            assert host is None and pipe is None, case
            outcomes["both_fail"] += 1
        else:
            assert np.array_equal(pipe, host), case
            outcomes["same"] += 1
    assert outcomes["same"] > 20


def test_random_file_sets_through_the_whole_pipeline(tmp_path):
    """End-to-end differential test: random sets of files (baseline / progressive JPEG with EXIF orientations, PNG with
    and without alpha, BMP, GIF, lossless and lossy WebP with EXIF orientations), random direction / mode / gap / filter, through ist_stitch_files_png (GPU Huffman,
    reconstruction, stitch, compressed PNG) against PIL's decoders + the CPU oracle."""
    rng = np.random.default_rng(31337)
    for case in range(80):
        n = int(rng.integers(1, 6))
        paths, bitmaps, orients = [], [], []
        for k in range(n):
            h, w = int(rng.integers(8, 200)), int(rng.integers(8, 200))
            a = _photo(1000 + 10 * case + k, h, w)
            kind = int(rng.integers(0, 8))
            p = tmp_path / ("c%d_%d" % (case, k))
            o = 1
            if kind <= 1:
                o = int(rng.integers(1, 9))
                ex = Image.Exif()
                ex[0x0112] = o
                kw = {"quality": int(rng.integers(40, 96)), "subsampling": int(rng.integers(0, 3)), "exif": ex}
                if kind == 1:
                    kw["progressive"] = True
                p = p.with_suffix(".jpg"); Image.fromarray(a).save(p, "JPEG", **kw)
            elif kind == 2:
                p = p.with_suffix(".png"); Image.fromarray(a).save(p, "PNG")
            elif kind == 3:
                rgba = np.concatenate([a, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], -1)
                p = p.with_suffix(".png"); Image.fromarray(rgba, "RGBA").save(p, "PNG")
            elif kind == 4:
                p = p.with_suffix(".bmp"); Image.fromarray(a).save(p, "BMP")
            elif kind == 5:
                p = p.with_suffix(".gif"); Image.fromarray(a).convert("P", palette=Image.ADAPTIVE).save(p, "GIF")
            else:                                   # WebP: lossless RGBA or lossy RGB, orientation in the container's EXIF chunk
                o = int(rng.integers(1, 9))
                ex = Image.Exif()
                ex[0x0112] = o
                p = p.with_suffix(".webp")
                if kind == 6:
                    rgba = np.concatenate([a, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], -1)
                    Image.fromarray(rgba, "RGBA").save(p, "WEBP", lossless=True, exact=True, exif=ex.tobytes())
                else:
                    Image.fromarray(a).save(p, "WEBP", quality=int(rng.integers(30, 96)), exif=ex.tobytes())
            bm = np.array(Image.open(p).convert("RGBA"))
            bm[bm[..., 3] == 0] = 0
            paths.append(str(p)); bitmaps.append(bm); orients.append(o)
        direction = "vertical" if rng.integers(0, 2) else "horizontal"
        opts = {"mode": ["min", "max", "original"][int(rng.integers(0, 3))], "gap": int(rng.integers(0, 2)) * 5,
                "filter": "nearest" if rng.integers(0, 2) else "bilinear"}
        # natural sizes follow the orientation, as getImageInfo reports them (the C side reads the EXIF tag itself)
        descs = []
        for bm, o in zip(bitmaps, orients):
            hh, ww = bm.shape[:2]
            descs.append({"width": ww, "height": hh, "orientation": o})
        rc, pd, rl = O.plan(descs, direction, opts["mode"], opts["gap"], U.oracle_limits(opts))
        assert rc == 0
        ref = O.render(pd, rl, descs, bitmaps, opts["filter"], 4)
        res = ist.stitch_files(paths, direction, opts)
        got = ist.decode_png(res["png"])
        assert got.shape == ref.shape, (case, got.shape, ref.shape)
        assert U.max_abs_diff(got, ref) <= (0 if opts["filter"] == "nearest" else 1), (case, opts, orients)


def test_sixty_four_files_in_one_call(tmp_path):
    """BASELINE configs[4] has 64 inputs: 64 JPEGs (one GPU Huffman batch, 64 decode threads) stitched in one call"""
    paths, bitmaps = [], []
    for k in range(64):
        a = _photo(3000 + k, 40 + (k % 7), 96)
        p = tmp_path / ("f%02d.jpg" % k)
        Image.fromarray(a).save(p, "JPEG", quality=60 + (k % 35), subsampling=k % 3)
        paths.append(str(p)); bitmaps.append(_pil(p.read_bytes()))
    res = ist.stitch_files(paths, "vertical", {"filter": "nearest"})
    assert np.array_equal(ist.decode_png(res["png"]), np.concatenate(bitmaps, 0))


def test_decode_files_into_device_memory_matches_pil_and_reports_phases(tmp_path):
    """ist_decode_files_device: files -> bitmaps in caller-owned HBM (baseline JPEG on the GPU, progressive JPEG / PNG / BMP
    on host threads), bit-exact vs PIL; the context reports the phase times of the call"""
    import torch
    rng = np.random.default_rng(31)
    blobs, want = [], []
    for k, (fmt, kw, size) in enumerate([("JPEG", {"quality": 88, "subsampling": 2}, (203, 317)), ("JPEG", {"quality": 75, "subsampling": 0}, (64, 64)),
                                         ("JPEG", {"progressive": True, "subsampling": 1}, (131, 97)), ("PNG", {}, (50, 77)), ("BMP", {}, (33, 18))]):
        h, w = size
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([(xx * 3 + yy + 10 * k) % 256, (yy * 2 + 5 * k) % 256, (xx + yy * 2) % 256], -1) + rng.integers(-15, 15, (h, w, 3))
        im = Image.fromarray(a.clip(0, 255).astype(np.uint8), "RGB")
        b = io.BytesIO()
        im.save(b, fmt, **kw)
        blobs.append(b.getvalue())
        want.append(np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGBA")))
    ist.set_phase_timing(True)
    try:
        out, imgs = ist.decode_files_device(blobs)
        times = ist.last_phase_times()
    finally:
        ist.set_phase_timing(False)
    assert [(i["width"], i["height"]) for i in imgs] == [(a.shape[1], a.shape[0]) for a in want]
    assert [i["opaque"] for i in imgs] == [True, True, True, False, False]
    for t, a in zip(out, want):
        assert np.array_equal(t.cpu().numpy(), a)
    assert times["host_decode"] > 0 and times["entropy_gpu"] > 0 and times["reconstruct"] > 0 and times["png"] == 0
    # into pitched, caller-allocated buffers
    wide = [torch.zeros((a.shape[0] + 2, a.shape[1] + 5, 4), dtype=torch.uint8, device="cuda") for a in want]
    views = [t[1:1 + a.shape[0], 2:2 + a.shape[1]] for t, a in zip(wide, want)]
    ist.decode_files_device(blobs, out=views)
    for t, v, a in zip(wide, views, want):
        assert np.array_equal(v.cpu().numpy(), a)
        assert int(t[0].max()) == 0 and int(t[-1].max()) == 0 and int(t[:, :2].max()) == 0
    # a buffer that is too small for what the file's header announces is refused
    small = [torch.zeros((a.shape[0] - 1, a.shape[1], 4), dtype=torch.uint8, device="cuda") for a in want]
    with pytest.raises(ist.StitchError):
        ist.decode_files_device(blobs, out=small)


def test_pipelined_and_phase_timed_runs_make_the_same_file(tmp_path):
    """The file pipeline runs per image (thread + stream each; band k of the canvas is rendered and exported while later
    images still decode; index.js:1441-1520 / :1559-1571).  With phase timing on the same steps run with barriers in
    between.  Same file either way, and the same pixels as PIL + the oracle; vertical (bands final one by one), horizontal
    (every band needed by the first slab), 'original' mode (centred, not full-width) and a mixed set of formats."""
    rng = np.random.default_rng(77)
    paths, bitmaps = [], []
    for k in range(7):
        h, w = int(rng.integers(300, 700)), int(rng.integers(300, 700))
        a = _photo(5000 + k, h, w)
        p = tmp_path / ("p%d" % k)
        if k == 2:
            p = p.with_suffix(".png"); Image.fromarray(a).save(p, "PNG")
        elif k == 4:
            p = p.with_suffix(".jpg"); Image.fromarray(a).save(p, "JPEG", quality=80, progressive=True)
        else:
            p = p.with_suffix(".jpg"); Image.fromarray(a).save(p, "JPEG", quality=85, subsampling=k % 3)
        paths.append(str(p)); bitmaps.append(np.array(Image.open(p).convert("RGBA")))
    for direction, opts in (("vertical", {"filter": "bilinear"}), ("horizontal", {"filter": "bilinear", "gap": 3}),
                            ("vertical", {"filter": "nearest", "mode": "original", "gap": 2})):
        fast = ist.stitch_files(paths, direction, opts)
        ist.set_phase_timing(True)
        try:
            slow = ist.stitch_files(paths, direction, opts)
            times = ist.last_phase_times()
        finally:
            ist.set_phase_timing(False)
        assert bytes(fast["png"]) == bytes(slow["png"]), (direction, opts)
        assert times["entropy_gpu"] > 0 and times["png"] > 0
        ref, _, _ = U.oracle_stitch(bitmaps, direction, opts)
        assert U.max_abs_diff(ist.decode_png(fast["png"]), ref) <= (0 if opts["filter"] == "nearest" else 1)
