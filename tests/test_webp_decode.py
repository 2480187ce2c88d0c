"""WebP decode ('webp' is in SUPPORTED_IMAGE_TYPES, pages/index/index.js:4, and among the
chooser's extensions, index.js:1030).  Host decoders, so this runs without a GPU.  Lossless (VP8L) is pinned bit for bit
by libwebp through PIL; lossy (VP8 key frames) is compared with PIL's own decode of the same file."""
import ctypes as C
import io
import struct

import numpy as np
import pytest
from PIL import Image

from imagestitching_amd import _lib as L


def _webp(a, mode, **kw):
    b = io.BytesIO()
    Image.fromarray(a, mode).save(b, "WEBP", **kw)
    return b.getvalue()


def _decode(buf):
    w, h, o = C.c_int32(), C.c_int32(), C.c_int32()
    rc = L.lib.ist_image_info(buf, len(buf), C.byref(w), C.byref(h), C.byref(o))
    if rc:
        return rc, None, 0
    out = np.full((h.value + 1, w.value, 4), 0xA5, np.uint8)
    rc = L.lib.ist_image_decode_rgba8(None, buf, len(buf), out.ctypes.data, w.value * 4, h.value)
    assert (out[h.value] == 0xA5).all(), "decoder wrote past the declared capacity"
    return rc, out[:h.value], o.value


def _pil(buf):
    return np.asarray(Image.open(io.BytesIO(buf)).convert("RGBA"))


def _pictures():
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:120, 0:173]
    smooth = np.stack([(xx * 2 + yy) % 256, (yy * 3) % 256, (xx + yy * 2) % 256, 255 - (xx % 256)], -1).astype(np.uint8)
    noise = rng.integers(0, 256, (64, 97, 4), dtype=np.uint8)
    pal = (rng.integers(0, 5, (50, 70)) * 50).astype(np.uint8)
    photo = (128 + 90 * np.sin(xx / 17.0 + yy / 31.0))[..., None] * np.ones(3) + rng.normal(0, 4, (120, 173, 3))
    return {"smooth": smooth, "noise": noise, "palette5": np.stack([pal, pal // 2, 255 - pal], -1), "two": (np.stack([pal > 100] * 3, -1) * 255).astype(np.uint8),
            "photo": photo.clip(0, 255).astype(np.uint8), "one": np.zeros((1, 1, 3), np.uint8), "wide": np.full((3, 5000, 3), 7, np.uint8),
            "tall": rng.integers(0, 3, (3000, 2, 3), dtype=np.uint8) * 100}


@pytest.mark.parametrize("name", ["smooth", "noise", "palette5", "two", "photo", "one", "wide", "tall"])
@pytest.mark.parametrize("quality,method", [(0, 0), (50, 3), (100, 6)])
def test_lossless_webp_is_bit_exact(name, quality, method):
    a = _pictures()[name]
    mode = "RGBA" if a.shape[2] == 4 else "RGB"
    buf = _webp(a, mode, lossless=True, quality=quality, method=method)
    assert buf[12:16] in (b"VP8L", b"VP8X")
    rc, got, o = _decode(buf)
    assert rc == 0, L.last_error()
    assert np.array_equal(got, _pil(buf))
    if mode == "RGB":
        assert np.array_equal(got[..., :3], a) and (got[..., 3] == 255).all()


def test_webp_exif_orientation_is_read_from_the_container():
    a = _pictures()["photo"]
    ex = Image.Exif()
    ex[0x0112] = 6
    buf = _webp(a, "RGB", lossless=True, exif=ex.tobytes())
    assert buf[12:16] == b"VP8X" and b"EXIF" in buf
    rc, got, o = _decode(buf)
    assert rc == 0 and o == 6 and np.array_equal(got[..., :3], a)       # the bitmap is returned as stored; the stitch turns it


def _riff(chunks):
    body = b"WEBP" + b"".join(t + struct.pack("<I", len(d)) + d + (b"\x00" if len(d) & 1 else b"") for t, d in chunks)
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_webp_container_errors():
    a = _pictures()["smooth"]
    buf = _webp(a, "RGBA", lossless=True)
    payload = buf[20:20 + struct.unpack("<I", buf[16:20])[0]]
    # animated without any frame: damaged, named
    anim = _riff([(b"VP8X", bytes([0x02, 0, 0, 0]) + struct.pack("<I", a.shape[1] - 1)[:3] + struct.pack("<I", a.shape[0] - 1)[:3]), (b"ANIM", bytes(6)), (b"VP8L", payload)])
    rc, _, _ = _decode(anim)
    assert rc == -6 and "animated" in L.last_error()
    # canvas size in VP8X disagrees with the frame
    lie = _riff([(b"VP8X", bytes([0, 0, 0, 0]) + struct.pack("<I", 9)[:3] + struct.pack("<I", 9)[:3]), (b"VP8L", payload)])
    assert _decode(lie)[0] == -6
    # no image chunk; truncated chunk; truncated bitstream at every 97th byte; bad signature
    assert _decode(_riff([(b"EXIF", b"II*\x00")]))[0] == -6
    assert _decode(buf[:30])[0] == -6
    for cut in range(21, len(payload), 97):
        rc, _, _ = _decode(_riff([(b"VP8L", payload[:cut])]))
        assert rc == -6, cut
    assert _decode(_riff([(b"VP8L", b"\x2e" + payload[1:])]))[0] == -6
    assert _decode(_riff([(b"VP8L", payload[:4] + bytes([payload[4] | 0x20]) + payload[5:])]))[0] == -6       # version != 0


def test_webp_capacity_is_checked_against_the_files_own_header():
    a = _pictures()["noise"]
    buf = _webp(a, "RGBA", lossless=True)
    out = np.zeros((10, a.shape[1], 4), np.uint8)
    assert L.lib.ist_image_decode_rgba8(None, buf, len(buf), out.ctypes.data, a.shape[1] * 4, 10) == -1


def test_lossless_webp_survives_bit_flips():
    """every single-byte corruption of a small file comes back as pixels or as an error code (the library stays up)"""
    a = _pictures()["smooth"][:40, :60]
    buf = bytearray(_webp(a, "RGBA", lossless=True))
    rng = np.random.default_rng(5)
    for _ in range(400):
        b = bytearray(buf)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(20, len(b)))] ^= 1 << int(rng.integers(0, 8))
        rc, got, _ = _decode(bytes(b))
        assert rc in (0, -6, -7, -1)


@pytest.mark.parametrize("size", [(1, 1), (2, 3), (15, 17), (16, 16), (17, 15), (31, 33), (100, 1), (1, 100), (255, 257)])
@pytest.mark.parametrize("quality,method", [(5, 0), (40, 2), (80, 4), (99, 6)])
def test_lossy_webp_equals_libwebp(size, quality, method):
    """VP8 key frames: boolean-coded modes and tokens, intra prediction, inverse WHT/DCT, the loop filter (simple and normal,
    with segments at method >= 4), then libwebp's fancy chroma upsampling and fixed-point YUV -> RGB: the bytes PIL returns"""
    h, w = size
    rng = np.random.default_rng(h * 1000 + w)
    yy, xx = np.mgrid[0:h, 0:w]
    a = ((128 + 90 * np.sin(xx / 37.0 + yy / 91.0))[..., None] * np.array([1, .8, .6]) + 40 * np.sin(xx * yy / 900.)[..., None] + rng.normal(0, 8, (h, w, 3))).clip(0, 255).astype(np.uint8)
    buf = _webp(a, "RGB", quality=quality, method=method)
    assert buf[12:16] == b"VP8 "
    rc, got, o = _decode(buf)
    assert rc == 0, L.last_error()
    assert np.array_equal(got, _pil(buf))


def test_lossy_webp_with_alpha_sharp_edges_and_a_photo_sized_frame():
    rng = np.random.default_rng(21)
    yy, xx = np.mgrid[0:200, 0:300]
    base = ((128 + 90 * np.sin(xx / 17.0 + yy / 31.0))[..., None] * np.ones(3) + rng.normal(0, 6, (200, 300, 3))).clip(0, 255).astype(np.uint8)
    for aq in (0, 50, 100):                                        # ALPH chunk: raw, or a VP8L stream, with its predictive filters
        a = np.dstack([base, ((xx * 3) % 256).astype(np.uint8)])
        buf = _webp(a, "RGBA", quality=70, alpha_quality=aq, method=4)
        assert buf[12:16] == b"VP8X" and b"ALPH" in buf
        rc, got, _ = _decode(buf)
        assert rc == 0 and np.array_equal(got, _pil(buf)), aq
    hard = np.dstack([base, np.where(xx % 50 < 25, 0, 255).astype(np.uint8)])
    for kw in ({}, {"exact": True}):
        buf = _webp(hard, "RGBA", quality=80, **kw)
        rc, got, _ = _decode(buf)
        assert rc == 0 and np.array_equal(got, _pil(buf))
    sharp = np.zeros((128, 160, 3), np.uint8)
    sharp[::7] = 255
    sharp[:, ::5, 0] = 255
    sharp[40:80, 60:100] = (255, 0, 0)
    for q in (20, 75, 100):
        buf = _webp(sharp, "RGB", quality=q)
        rc, got, _ = _decode(buf)
        assert rc == 0 and np.array_equal(got, _pil(buf)), q
    big = np.asarray(Image.fromarray(base).resize((1008, 756)))     # 63 x 48 macroblocks (ragged right column)
    buf = _webp(big, "RGB", quality=85)
    rc, got, _ = _decode(buf)
    assert rc == 0 and np.array_equal(got, _pil(buf))


def test_lossy_webp_truncations_and_bit_flips_are_survived():
    rng = np.random.default_rng(8)
    yy, xx = np.mgrid[0:48, 0:64]
    a = np.stack([(xx * 4) % 256, (yy * 5) % 256, ((xx + yy) * 3) % 256], -1).astype(np.uint8)
    buf = bytearray(_webp(a, "RGB", quality=60))
    for cut in range(12, len(buf), 23):
        rc, _, _ = _decode(bytes(buf[:cut]))
        assert rc in (0, -6)                                    # a truncated token partition decodes zeros: pixels or an error, never a fault
    for _ in range(300):
        b = bytearray(buf)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(20, len(b)))] ^= 1 << int(rng.integers(0, 8))
        assert _decode(bytes(b))[0] in (0, -6, -7, -1)


def test_animated_webp_shows_its_first_frame():
    """An animated file given to Image.src shows one frame (utils/canvas.js:27-121; 'webp' in index.js:4): the first ANMF frame on
    its transparent canvas.  PIL (libwebp's animation decoder) is the witness for whole-canvas frames; a frame placed at an
    offset is checked against the placement the container prescribes."""
    pics = _pictures()
    frames = [pics["smooth"], np.roll(pics["smooth"], 7, axis=1), np.roll(pics["smooth"], 9, axis=0)]
    for kw in ({"lossless": True}, {"quality": 80}):
        b = io.BytesIO()
        Image.fromarray(frames[0], "RGBA").save(b, "WEBP", save_all=True, append_images=[Image.fromarray(f, "RGBA") for f in frames[1:]], duration=50, **kw)
        buf = b.getvalue()
        assert b"ANMF" in buf
        im = Image.open(io.BytesIO(buf))
        im.seek(0)
        ref = np.asarray(im.convert("RGBA"))
        rc, got, _ = _decode(buf)
        assert rc == 0, L.last_error()
        assert got.shape == ref.shape
        if kw.get("lossless"):
            assert np.array_equal(got, ref)
        else:                                   # (libwebp's animation decoder blends the frame onto a canvas; alpha 0 pixels lose their colour there)
            vis = ref[..., 3] > 0
            assert np.array_equal(got[vis], ref[vis])
    # a 20 x 10 frame at (6, 4) of a 40 x 30 canvas
    small = np.ascontiguousarray(pics["noise"][:10, :20])
    one = _webp(small, "RGBA", lossless=True, exact=True)
    payload = one[20:20 + struct.unpack("<I", one[16:20])[0]] if one[12:16] == b"VP8L" else one[one.index(b"VP8L") + 8:]
    p24 = lambda v: struct.pack("<I", v)[:3]      # noqa: E731
    anmf = p24(3) + p24(2) + p24(19) + p24(9) + p24(100) + b"\x00" + b"VP8L" + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")
    buf = _riff([(b"VP8X", bytes([0x12, 0, 0, 0]) + p24(39) + p24(29)), (b"ANIM", bytes(6)), (b"ANMF", anmf)])
    rc, got, _ = _decode(buf)
    assert rc == 0, L.last_error()
    want = np.zeros((30, 40, 4), np.uint8)
    want[4:14, 6:26] = small
    assert np.array_equal(got, want)
    # a frame that leaves its canvas is refused
    bad = _riff([(b"VP8X", bytes([0x12, 0, 0, 0]) + p24(20) + p24(29)), (b"ANIM", bytes(6)), (b"ANMF", anmf)])
    assert _decode(bad)[0] == -6
