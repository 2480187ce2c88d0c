"""PNG export on the GPU: the file must be a valid PNG that decodes, with two independent
decoders (PIL; and a by-hand chunk walk + zlib.decompress), to exactly the canvas bytes."""
import io
import os
import struct
import zlib

import numpy as np
import pytest

import imagestitching_amd as ist
from tests import util as U

pytestmark = pytest.mark.gpu


def _decode_by_hand(png):
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr, n_idat = 8, b"", None, 0
    while pos < len(png):
        ln, typ = struct.unpack(">I4s", png[pos:pos + 8])
        data = png[pos + 8:pos + 8 + ln]
        crc, = struct.unpack(">I", png[pos + 8 + ln:pos + 12 + ln])
        assert zlib.crc32(typ + data) == crc, "bad CRC in %r chunk" % typ
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", data)
        elif typ == b"IDAT":
            idat += data
            n_idat += 1
        elif typ == b"IEND":
            assert ln == 0 and pos + 12 == len(png)
        pos += 12 + ln
    w, h, depth, ctype, comp, filt, inter = ihdr
    assert (depth, ctype, comp, filt, inter) == (8, 6, 0, 0, 0)
    raw = zlib.decompress(idat)                      # also verifies the Adler-32
    assert len(raw) == h * (4 * w + 1)
    rows = np.frombuffer(raw, np.uint8).reshape(h, 4 * w + 1)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 4), n_idat


def _check(a):
    from PIL import Image
    png = ist.encode_png(a, level=0)
    by_hand, _ = _decode_by_hand(png)
    assert np.array_equal(by_hand, a)
    pil = np.asarray(Image.open(io.BytesIO(png)).convert("RGBA"))
    assert np.array_equal(pil, a)
    return png


@pytest.mark.parametrize("w,h", [(1, 1), (3, 5), (4, 4), (5, 3), (64, 33), (257, 19), (1000, 700), (4032, 50)])
def test_png_roundtrip_shapes(w, h):
    _check(U.rand_image(400 + w, h, w, opaque=False))


def test_png_rows_longer_than_one_stored_block():
    """36288-px rows (BASELINE configs[2]) are 145 KB: three 63-KiB stored blocks per row."""
    _check(U.rand_image(410, 7, 36288))
    _check(U.rand_image(411, 3, 16129))       # 64516 bytes: one byte over a block


def test_png_more_rows_than_65535():
    _check(U.rand_image(412, 70000, 8))


def test_png_multiple_idat_chunks(monkeypatch):
    monkeypatch.setenv("IST_PNG_IDAT_LIMIT", "65536")
    a = U.rand_image(413, 300, 257)
    png = ist.encode_png(a, level=0)
    got, n_idat = _decode_by_hand(png)
    assert n_idat > 3 and np.array_equal(got, a)


def test_stitch_png_equals_stitch_pixels():
    from PIL import Image
    px = [U.rand_image(420 + i, h, w) for i, (w, h) in enumerate([(403, 302), (302, 403), (400, 300)])]
    for direction in ("vertical", "horizontal"):
        ref = ist.stitch(U.hip_images(px), direction, {"mode": "max", "gap": 4})
        got = ist.stitch_png(U.hip_images(px), direction, {"mode": "max", "gap": 4})
        assert (got["width"], got["height"]) == (ref["width"], ref["height"])
        dec = np.asarray(Image.open(io.BytesIO(got["png"])).convert("RGBA"))
        assert np.array_equal(dec, ref["data"])


def test_png_device_resident_full_size():
    """BASELINE configs[1] canvas (4032x27216, 439 MB) encoded without leaving HBM; decoded by zlib on the host."""
    import torch
    canvas = torch.randint(0, 256, (27216, 4032, 4), dtype=torch.uint8, device="cuda")
    out, n = ist.encode_png_device(canvas, level=0)
    torch.cuda.synchronize()
    png = out.cpu().numpy().tobytes()
    assert len(png) == n and n < ist._lib.lib.ist_png_bound(4032, 27216)
    got, n_idat = _decode_by_hand(png)
    assert n_idat == 1 and np.array_equal(got, canvas.cpu().numpy())


def test_compressed_png_device_resident_full_size():
    """the same canvas size with photo-like content, compressed form: 439 MB -> about 0.43 of it, decoded by zlib + PIL-free un-filter (own decoder)"""
    import torch
    H, W = 27216, 4032
    yy = torch.arange(H, device="cuda", dtype=torch.float32)[:, None]
    xx = torch.arange(W, device="cuda", dtype=torch.float32)[None, :]
    c = torch.stack([128 + 90 * torch.sin(xx / 37 + yy / 91), 128 + 80 * torch.cos(xx / 53 - yy / 29), 100 + 0.03 * xx + 0.002 * yy, torch.full((H, W), 255.0, device="cuda")], -1)
    c[..., :3] += torch.randn((H, W, 3), device="cuda") * 2.0
    canvas = c.clamp(0, 255).to(torch.uint8).contiguous()
    del c
    out, n = ist.encode_png_device(canvas, level=1)
    torch.cuda.synchronize()
    png = out.cpu().numpy().tobytes()
    assert len(png) == n and n < 0.5 * canvas.numel()
    ihdr, raw, n_idat = _walk(png)
    assert ihdr == (W, H, 8, 6, 0, 0, 0) and len(raw) == H * (4 * W + 1) and n_idat == 1
    assert np.array_equal(ist.decode_png(png), canvas.cpu().numpy())


def test_stitch_files_png_in_png_out(tmp_path):
    from PIL import Image
    px = [U.rand_image(430 + i, h, w, opaque=(i != 1)) for i, (w, h) in enumerate([(120, 90), (90, 120), (64, 33)])]
    paths = []
    for i, a in enumerate(px):
        Image.fromarray(a, "RGBA").save(tmp_path / ("in%d.png" % i))
        paths.append(str(tmp_path / ("in%d.png" % i)))
    res = ist.stitch_files(paths, "horizontal", {"mode": "min", "gap": 2}, out_path=str(tmp_path / "out.png"))
    ref, pd, _ = U.oracle_stitch(px, "horizontal", {"mode": "min", "gap": 2})
    got = np.asarray(Image.open(tmp_path / "out.png").convert("RGBA"))
    assert (res["width"], res["height"]) == (ref.shape[1], ref.shape[0])
    assert U.max_abs_diff(got, ref) <= 1
    assert np.array_equal(ist.decode_png(res["png"]), got)          # own decoder reads own encoder's files
    view = ist.stitch_files(paths, "horizontal", {"mode": "min", "gap": 2}, copy=False)["png"]      # zero-copy hand-over
    assert isinstance(view, memoryview) and bytes(view) == res["png"]
    del view
    with pytest.raises(ist.StitchError) as e:
        (tmp_path / "bad.png").write_bytes(b"not a png at all, definitely not, no no no no no no no no no no no")
        ist.stitch_files([paths[0], str(tmp_path / "bad.png")], "vertical")
    assert "图片1解码异常" in str(e.value)


# ---- the compressing form (ist_ctx_set_png_level 1): Paeth + run-length matches + a dynamic Huffman code per 16 KiB ----
def _walk(png):
    """chunk walk: every CRC, the Adler (zlib.decompress), the filter bytes; returns (ihdr, raw stream, IDAT count)"""
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr, n_idat, seen_end = 8, b"", None, 0, False
    while pos < len(png):
        ln, typ = struct.unpack(">I4s", png[pos:pos + 8])
        data = png[pos + 8:pos + 8 + ln]
        crc, = struct.unpack(">I", png[pos + 8 + ln:pos + 12 + ln])
        assert zlib.crc32(typ + data) == crc, "bad CRC in %r chunk" % typ
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", data)
        elif typ == b"IDAT":
            idat += data
            n_idat += 1
        elif typ == b"IEND":
            assert ln == 0 and pos + 12 == len(png)
            seen_end = True
        pos += 12 + ln
    assert seen_end
    return ihdr, zlib.decompress(idat), n_idat


def _check_compressed(a, max_ratio=None):
    from PIL import Image
    png = ist.encode_png(a, level=1)
    h, w = a.shape[:2]
    ihdr, raw, n_idat = _walk(png)
    assert ihdr == (w, h, 8, 6, 0, 0, 0)
    assert len(raw) == h * (4 * w + 1)
    assert (np.frombuffer(raw, np.uint8).reshape(h, 4 * w + 1)[:, 0] == 4).all()      # Paeth on every row
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(png)).convert("RGBA")), a)
    assert np.array_equal(ist.decode_png(png), a)
    if max_ratio is not None:
        assert len(png) <= max_ratio * a.nbytes + 400, (len(png), a.nbytes)
    return png, n_idat


def _photo_like(seed, h, w):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 90 * np.sin(xx / 37.0 + yy / 91.0), 128 + 80 * np.cos(xx / 53.0 - yy / 29.0), 100 + 0.03 * xx + 0.05 * yy, np.full((h, w), 255.0)], -1)
    base[..., :3] += rng.normal(0, 2.0, (h, w, 3))
    return base.clip(0, 255).astype(np.uint8)


def _screenshot_like(seed, h, w):
    rng = np.random.default_rng(seed)
    a = np.full((h, w, 4), 255, np.uint8)
    for _ in range(max(1, h // 12)):                      # "text lines": short dark dashes on white
        y = int(rng.integers(0, h)); x = int(rng.integers(0, max(1, w - 40)))
        a[y:y + 2, x:x + int(rng.integers(5, 40)), :3] = rng.integers(0, 80, 3, dtype=np.uint8)
    a[h // 3:h // 3 + max(1, h // 10), :, :3] = (40, 120, 200)      # a flat coloured banner
    return a


@pytest.mark.parametrize("w,h", [(1, 1), (3, 5), (4, 4), (64, 33), (257, 19), (1000, 300), (4032, 50), (4095, 3), (4096, 3), (5000, 5), (36288, 2)])
def test_compressed_png_roundtrip_random(w, h):
    """random bytes: every chunk falls back to a stored block; whole rows per chunk, several rows per chunk, and
    pieces of one row (rows longer than 16 KiB) all round-trip"""
    png, _ = _check_compressed(U.rand_image(500 + w, h, w, opaque=False))
    assert len(png) <= 1.02 * 4 * w * h + 400


@pytest.mark.parametrize("w,h", [(7, 9), (640, 480), (4032, 64), (5000, 40), (300, 2000)])
def test_compressed_png_photo_like(w, h):
    a = _photo_like(w + h, h, w)
    png, _ = _check_compressed(a, max_ratio=0.62)
    raw = np.frombuffer(_walk(png)[1], np.uint8)
    yard = len(zlib.compress(raw.tobytes(), 6))
    print("photo-like %dx%d: raw %d, ours %d (%.3f), zlib-6 on the same Paeth stream %d (%.3f)" % (w, h, a.nbytes, len(png), len(png) / a.nbytes, yard, yard / a.nbytes))
    assert len(png) <= 1.25 * yard + 400


@pytest.mark.parametrize("w,h", [(750, 1334), (4032, 200)])
def test_compressed_png_flat_content(w, h):
    a = _screenshot_like(w, h, w)
    png, _ = _check_compressed(a, max_ratio=0.04)
    print("screenshot-like %dx%d: raw %d, ours %d (%.4f)" % (w, h, a.nbytes, len(png), len(png) / a.nbytes))


def test_compressed_png_translucent_and_extremes():
    a = _photo_like(3, 120, 333)
    a[..., 3] = (np.arange(333) % 256).astype(np.uint8)[None, :]
    _check_compressed(a)
    _check_compressed(np.zeros((50, 70, 4), np.uint8), max_ratio=0.02)
    _check_compressed(np.full((50, 70, 4), 255, np.uint8), max_ratio=0.02)
    sk = np.zeros((64, 256, 4), np.uint8)                      # a skewed histogram: long Huffman codes
    sk[..., 0] = (np.arange(256) ** 2 // 256).astype(np.uint8)[None, :]
    sk[::7, ::5, 1] = 200
    _check_compressed(sk)


def test_compressed_png_splits_idat(monkeypatch):
    monkeypatch.setenv("IST_PNG_IDAT_LIMIT", "65536")
    a = U.rand_image(77, 200, 300, opaque=False)
    png, n_idat = _check_compressed(a)
    assert n_idat >= 3
    b = _photo_like(5, 300, 800)
    png, n_idat = _check_compressed(b)
    assert n_idat >= 2


def test_stitch_png_with_compression_matches_stored_pixels():
    imgs = [{"width": 320, "height": 240, "data": _photo_like(k, 240, 320)} for k in range(3)]
    a = ist.stitch_png(imgs, "vertical", {"gap": 6, "pngLevel": 0})
    b = ist.stitch_png(imgs, "vertical", {"gap": 6, "pngLevel": 1})
    assert np.array_equal(ist.decode_png(a["png"]), ist.decode_png(b["png"]))
    assert len(b["png"]) < 0.7 * len(a["png"])


def test_compressed_png_code_longer_than_15_bits_is_limited():
    """Fibonacci symbol counts make the unrestricted Huffman code ~19 bits deep: the length limiter must produce a
    valid, complete 15-bit code (zlib refuses both over-subscribed and incomplete literal/length codes)."""
    fib = [1, 1]
    while len(fib) < 19:
        fib.append(fib[-1] + fib[-2])
    vals = np.arange(19) * 7 + 20                                   # 19 distinct residual values
    res = np.repeat(vals, fib).astype(np.uint8)
    rng = np.random.default_rng(11)
    rng.shuffle(res)
    res = np.concatenate([res, np.full((-len(res)) % 4, 20, np.uint8)])
    a = np.ascontiguousarray(np.cumsum(res.reshape(-1, 4).astype(np.int64), axis=0).astype(np.uint8)[None, :, :]).copy()      # h = 1: Paeth = Sub
    png, _ = _check_compressed(a)
    assert len(png) < 0.5 * a.nbytes


def test_compressed_png_codes_complete_on_adversarial_histograms():
    """The encoder's code lengths are Shannon lengths with the Kraft slack handed back until the code is COMPLETE
    (ist_png_deflate.hip, phase C; tools/sim_code_lengths.py restates the rule).  Histograms that stress it: two symbols
    (one very rare), powers of two, one dominant symbol + many singletons, every literal once, uniform over a few values.
    zlib (through PIL) refuses an over-subscribed or incomplete code, so decoding IS the check."""
    rng = np.random.default_rng(77)

    def row_image(res):
        res = np.asarray(res, np.uint8)
        res = np.concatenate([res, np.full((-len(res)) % 4, res[0], np.uint8)])
        return np.ascontiguousarray(np.cumsum(res.reshape(-1, 4).astype(np.int64), axis=0).astype(np.uint8)[None, :, :]).copy()      # h = 1: Paeth = Sub

    cases = []
    cases.append(np.where(np.arange(16000) == 7777, 9, 200))                                   # 15999 : 1
    cases.append(np.repeat(np.arange(14) * 3 + 1, 2 ** np.arange(14))[:16000])                 # counts 1, 2, 4, ... 8192
    cases.append(np.concatenate([np.full(15000, 33), np.arange(250)]))                         # one dominant + 250 singletons
    cases.append(np.arange(256).repeat(2))                                                     # every literal twice
    cases.append(rng.choice(np.array([5, 6, 7], np.uint8), 12000))                             # three values, equal
    cases.append(np.concatenate([np.full(3, 1), np.full(3, 2), np.full(16000, 3)]))           # two rare, one common
    for k in range(6):                                                                          # random power-law shapes
        ns = int(rng.integers(2, 256))
        c = np.maximum(1, (2.0 ** rng.uniform(0, 12, ns)).astype(int))
        c = (c * min(1.0, 15000 / c.sum())).astype(int).clip(1)
        cases.append(np.repeat(rng.permutation(256)[:ns], c))
    for res in cases:
        res = np.asarray(res).copy()
        rng.shuffle(res)
        _check_compressed(row_image(res))


def test_compressed_png_many_random_shapes_and_contents():
    rng = np.random.default_rng(2024)
    for case in range(40):
        h, w = int(rng.integers(1, 200)), int(rng.integers(1, 400))
        kind = case % 5
        if kind == 0:
            a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        elif kind == 1:
            a = rng.choice(np.array([0, 1, 2, 255], np.uint8), (h, w, 4))
        elif kind == 2:
            a = np.repeat(rng.integers(0, 256, (h, (w + 15) // 16, 4), dtype=np.uint8), 16, axis=1)[:, :w]      # runs
        elif kind == 3:
            a = _photo_like(case, h, w)
        else:
            a = _screenshot_like(case, h, w)
        _check_compressed(np.ascontiguousarray(a))


def test_compressed_png_of_several_slabs_reaches_the_host_whole(monkeypatch):
    """A canvas of more than 4096 chunks (64 MiB of filtered stream) is compressed slab by slab, one ahead of the host, and
    each slab's bytes travel to the pinned result on a second stream while the next one compresses (ist_png_deflate.hip):
    three slabs here, each its own IDAT, the last one ragged; then the same with a small IDAT limit, so that slabs also
    split inside.  Both files must decode (PIL, the product's decoder, and a by-hand chunk walk that checks every CRC
    and the Adler-32) to the canvas."""
    h, w = 9000, 4032                                             # 145 MB; a chunk is one row here: 9000 chunks = 4096 + 4096 + 808
    a = np.empty((h, w, 4), np.uint8)
    a[:3000] = _photo_like(11, 3000, w)
    a[3000:6000] = U.rand_image(12, 3000, w, opaque=False)        # incompressible: stored chunks
    a[6000:] = 255
    a[6000::37, 100:900, :3] = 30                                 # flat with a few lines
    for limit, least in ((None, 3), ("1048576", 40)):
        if limit:
            monkeypatch.setenv("IST_PNG_IDAT_LIMIT", limit)
        png, n_idat = _check_compressed(a)
        assert n_idat >= least, n_idat


def test_a_failed_encode_drains_its_streams_before_the_pinned_blocks_go_back_to_the_pool():
    """every error return of the compressing encoder first waits for both streams - kernels in flight
    write their per-chunk results into pooled pinned blocks.  IST_TUNING=1 IST_PNG_FAIL_AT=1 fails the layout of slab 1 while
    slab 2 is compressing; the next encodes of the same process (which re-use those blocks) must still be right."""
    import subprocess
    import sys
    code = """
import sys, io
sys.path.insert(0, %r)
import numpy as np
from PIL import Image
import imagestitching_amd as ist
Image.MAX_IMAGE_PIXELS = None
big = np.random.default_rng(5).integers(0, 256, (9000, 4096, 4), dtype=np.uint8)     # 147 MB: three slabs of 64 MiB
big[..., 3] = 255
try:
    ist.encode_png(big)
    print("no failure")
except ist.StitchError as e:
    print("failed as asked:", e.reason)
for k in range(4):
    small = np.random.default_rng(10 + k).integers(0, 256, (700 + 13 * k, 900, 4), dtype=np.uint8)
    png = ist.encode_png(small)
    back = np.asarray(Image.open(io.BytesIO(png)).convert("RGBA"))
    assert np.array_equal(back, small), k
print("pool ok")
""" % (U.ROOT,)
    env = dict(os.environ, IST_TUNING="1", IST_PNG_FAIL_AT="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "failed as asked" in r.stdout and "pool ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
