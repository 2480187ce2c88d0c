"""Crafted files that once drove the host-side parsers out of bounds (round-1 advisor findings): every one must come
back as an error code from the PRODUCTION library (CRC gate on), never as a write past the caller's buffer.  No GPU:
PNG / BMP / GIF decode on the host, and the JPEG cases die in the container parser before any device work."""
import ctypes as C
import io
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

from imagestitching_amd import _lib as L


def _chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def _png(chunks):
    return b"\x89PNG\r\n\x1a\n" + b"".join(chunks)


def _ihdr(w, h):
    return _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))


def _idat(w, h):
    raw = b"".join(b"\x00" + bytes([(x + y) & 255 for x in range(w) for _ in range(4)]) for y in range(h))
    return _chunk(b"IDAT", zlib.compress(raw))


def _png_decode(buf, w, h, guard=4096):
    """decode into a buffer of exactly h rows followed by a guard area that must stay untouched"""
    out = np.full(w * h * 4 + guard, 0xA5, np.uint8)
    rc = L.lib.ist_png_decode_rgba8(buf, len(buf), out.ctypes.data, w * 4, h)
    assert (out[w * h * 4:] == 0xA5).all(), "decoder wrote past the declared capacity"
    return rc, out[:w * h * 4].reshape(h, w, 4)


def test_png_second_ihdr_is_rejected_and_cannot_overrun_the_callers_buffer():
    bad = _png([_ihdr(10, 10), _ihdr(10, 1000), _idat(10, 1000), _chunk(b"IEND", b"")])
    w, h = C.c_int32(), C.c_int32()
    # info and decode agree on ONE header: info reads the first IHDR, decode refuses the second
    assert L.lib.ist_png_info(bad, len(bad), C.byref(w), C.byref(h)) == 0 and (w.value, h.value) == (10, 10)
    rc, _ = _png_decode(bad, 10, 10)
    assert rc == -6 and "IHDR" in L.last_error()


def test_png_must_start_with_ihdr():
    bad = _png([_idat(4, 4), _ihdr(4, 4), _idat(4, 4), _chunk(b"IEND", b"")])
    rc, _ = _png_decode(bad, 4, 4)
    assert rc == -6


def test_png_capacity_is_checked_against_the_files_own_header():
    ok = _png([_ihdr(10, 12), _idat(10, 12), _chunk(b"IEND", b"")])
    rc, px = _png_decode(ok, 10, 12)
    assert rc == 0 and px[3, 2, 0] == 5
    out = np.zeros(10 * 4 * 4, np.uint8)
    assert L.lib.ist_png_decode_rgba8(ok, len(ok), out.ctypes.data, 40, 4) == -1       # 4 rows offered, 12 needed


def _jpeg(w=16, h=16, **kw):
    b = io.BytesIO()
    yy, xx = np.mgrid[0:h, 0:w]
    Image.fromarray(np.stack([xx * 9 % 256, yy * 7 % 256, (xx + yy) * 5 % 256], -1).astype(np.uint8), "RGB").save(b, "JPEG", **kw)
    return b.getvalue()


def _segments(f):
    pos, out = 2, []
    while f[pos] == 0xFF:
        m = f[pos + 1]
        n = struct.unpack(">H", f[pos + 2:pos + 4])[0]
        out.append((m, pos, 2 + n))
        pos += 2 + n
        if m == 0xDA:
            break
    return out, pos


def _jpeg_parse(buf):
    w, h, o = C.c_int32(), C.c_int32(), C.c_int32()
    rc = L.lib.ist_jpeg_info(buf, len(buf), C.byref(w), C.byref(h), C.byref(o))
    return rc, w.value, h.value


def test_jpeg_second_frame_header_is_rejected():
    """SOF(8x8)+SOS+SOF(256x256)+SOS: the second SOF used to resize the frame under the planes of the first"""
    small, big = _jpeg(8, 8), _jpeg(256, 256)
    segs_b, data_b = _segments(big)
    sof_b = next(big[p:p + n] for m, p, n in segs_b if m == 0xC0)
    sos_b = next(big[p:p + n] for m, p, n in segs_b if m == 0xDA)
    bad = small[:-2] + sof_b + sos_b + big[data_b:]
    assert _jpeg_parse(bad)[0] == 0                                  # header-only: the first frame
    # the full parse (host entropy decode: no GPU involved) must refuse the file at the second SOF
    out = np.zeros((8, 8, 4), np.uint8)
    rc = L.lib.ist_image_decode_rgba8(None, bad, len(bad), out.ctypes.data, 32, 8)
    assert rc in (-4, -6)                                            # no context (parse never ran) or decode error
    if rc == -6:
        assert "frame header" in L.last_error()


def _run_exact_files(tmp_path, files):
    """exact files through the ASan/UBSan build of the parsers (tools/run_fuzz.sh with 0 mutations)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = []
    for name, blob in files.items():
        (tmp_path / name).write_bytes(blob)
        paths.append(str(tmp_path / name))
    env = dict(os.environ, IST_FUZZ_BIN=str(tmp_path / "fuzz1"))
    r = subprocess.run([os.path.join(root, "tools", "run_fuzz.sh"), "0"] + paths, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    return r.stdout


def test_crafted_files_are_clean_under_the_sanitizers(tmp_path):
    small, big = _jpeg(8, 8), _jpeg(256, 256)
    segs_b, data_b = _segments(big)
    sof_b = next(big[p:p + n] for m, p, n in segs_b if m == 0xC0)
    sos_b = next(big[p:p + n] for m, p, n in segs_b if m == 0xDA)
    dup_sof = small[:-2] + sof_b + sos_b + big[data_b:]
    # a 4:2:0 scan that lists Y three times: 12 blocks per MCU would overrun the 10 slot entries of the GPU scan record
    j = bytearray(_jpeg(32, 32, subsampling=2))
    segs, _ = _segments(bytes(j))
    m, p, n = next(s for s in segs if s[0] == 0xDA)
    y_id = j[p + 5]
    j[p + 7] = y_id
    j[p + 9] = y_id
    # SOS with an empty payload at the very end of the file (the parser read d[0] with dl == 0)
    empty_sos = small[:2] + b"".join(small[p:p + n] for m, p, n in _segments(small)[0] if m != 0xDA) + b"\xff\xda\x00\x02"
    files = {"dup_sof.jpg": dup_sof, "dup_comp.jpg": bytes(j), "empty_sos.jpg": empty_sos,
             "dup_ihdr.png": _png([_ihdr(10, 10), _ihdr(10, 1000), _idat(10, 1000), _chunk(b"IEND", b"")])}
    out = _run_exact_files(tmp_path, files)
    assert "0 decoded, %d rejected" % len(files) in out, out


def _bmp32(w, h, masks, px):
    """BITMAPV4HEADER, BI_BITFIELDS, 32 bpp"""
    dib = struct.pack("<IiiHHIIiiII", 108, w, -h, 1, 32, 3, w * h * 4, 2835, 2835, 0, 0) + struct.pack("<IIII", *masks) + b"\x00" * (108 - 56)
    off = 14 + len(dib)
    return b"BM" + struct.pack("<IHHI", off + len(px), 0, 0, off) + dib + px


def _misc_decode(buf, w, h):
    out = np.zeros((h, w, 4), np.uint8)
    rc = L.lib.ist_image_decode_rgba8(None, buf, len(buf), out.ctypes.data, w * 4, h)
    return rc, out


def test_bmp_v4_bgra_with_the_standard_alpha_mask():
    """mask 0xFF000000 reaches bit 31: the per-pixel bit scan shifted a 32-bit value by 32 (UB) here"""
    px = np.array([[[10, 20, 30, 40], [50, 60, 70, 255]], [[1, 2, 3, 4], [250, 251, 252, 0]]], np.uint8)     # B, G, R, A
    f = _bmp32(2, 2, (0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000), px.tobytes())
    rc, out = _misc_decode(f, 2, 2)
    assert rc == 0
    assert (out == px[..., [2, 1, 0, 3]]).all()
    ref = np.asarray(Image.open(io.BytesIO(f)).convert("RGBA"))
    assert (out == ref).all()


@pytest.mark.timeout(20)
def test_bmp_all_ones_mask_terminates():
    """mask 0xFFFFFFFF made the -O3 build of the bit scan loop forever (one hostile BMP hung a decode thread)"""
    f = _bmp32(2, 2, (0xFFFFFFFF, 0x0000FF00, 0x000000FF, 0), bytes(range(16)))
    rc, out = _misc_decode(f, 2, 2)
    assert rc == 0 and out[0, 0, 0] == 3          # top byte of the 32-bit field
    g = _bmp32(2, 2, (0x00FF00FF, 0x0000FF00, 0x000000FF, 0), bytes(16))
    assert _misc_decode(g, 2, 2)[0] == -6         # a mask with a hole is not a bit field


def test_bmp_height_int_min_is_rejected():
    f = bytearray(_bmp32(2, 2, (0xFF0000, 0xFF00, 0xFF, 0), bytes(16)))
    f[22:26] = struct.pack("<i", -2 ** 31)
    w, h, o = C.c_int32(), C.c_int32(), C.c_int32()
    assert L.lib.ist_image_info(bytes(f), len(f), C.byref(w), C.byref(h), C.byref(o)) == -6


def test_gif_descriptor_larger_than_the_screen_does_not_reserve_gigabytes():
    b = io.BytesIO()
    Image.fromarray((np.arange(64).reshape(8, 8) * 3).astype(np.uint8), "P").save(b, "GIF")
    f = bytearray(b.getvalue())
    at = f.index(b"\x2c")
    f[at + 5:at + 9] = struct.pack("<HH", 65535, 65535)           # image descriptor: 4 G pixels on an 8x8 screen
    rc, _ = _misc_decode(bytes(f), 8, 8)
    assert rc in (0, -6)
