"""PNG decode (PNG inputs): pinned bit for bit by PIL on files PIL wrote — every colour
type, bit depth, filter heuristic and tRNS flavour the decoder claims.  CPU only (the decoder is host code)."""
import io
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

import imagestitching_amd as ist

RNG = np.random.default_rng(77)


def _png(img, **kw):
    b = io.BytesIO()
    img.save(b, "PNG", **kw)
    return b.getvalue()


def _smooth(h, w, c):
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([(xx * 3 + yy * 5 + 40 * k) % 256 for k in range(c)], -1).astype(np.uint8)
    return a


@pytest.mark.parametrize("mode,shape", [("RGBA", (37, 53, 4)), ("RGB", (21, 64, 3)), ("L", (19, 33)), ("LA", (16, 17, 2))])
@pytest.mark.parametrize("content", ["noise", "smooth"])
def test_8bit_colour_types(mode, shape, content):
    a = RNG.integers(0, 256, shape, dtype=np.uint8) if content == "noise" else _smooth(shape[0], shape[1], shape[2] if len(shape) == 3 else 1).reshape(shape)
    img = Image.fromarray(a, mode)
    for kw in ({}, {"compress_level": 9}, {"optimize": True}):
        got = ist.decode_png(_png(img, **kw))
        assert np.array_equal(got, np.asarray(img.convert("RGBA")))


@pytest.mark.parametrize("bits", [1, 2, 4, 8])
def test_palette_and_low_bit_depths(bits):
    n = 1 << bits
    idx = RNG.integers(0, n, (23, 45), dtype=np.uint8)
    img = Image.fromarray(idx, "P")
    pal = RNG.integers(0, 256, 3 * n, dtype=np.uint8).tolist()
    img.putpalette(pal + [0] * (768 - len(pal)))
    data = _png(img, bits=bits)
    assert np.array_equal(ist.decode_png(data), np.asarray(img.convert("RGBA")))
    # palette transparency
    data = _png(img, bits=bits, transparency=bytes(RNG.integers(0, 256, n, dtype=np.uint8).tolist()))
    assert np.array_equal(ist.decode_png(data), np.asarray(Image.open(io.BytesIO(data)).convert("RGBA")))


def test_grey_1bit_and_trns_colour_key():
    a = (RNG.integers(0, 2, (17, 29)) * 255).astype(np.uint8)
    img = Image.fromarray(a, "L").convert("1")
    assert np.array_equal(ist.decode_png(_png(img)), np.asarray(img.convert("RGBA")))
    rgb = Image.fromarray(RNG.integers(0, 4, (9, 9, 3), dtype=np.uint8) * 60, "RGB")
    data = _png(rgb, transparency=(60, 120, 180))
    assert np.array_equal(ist.decode_png(data), np.asarray(Image.open(io.BytesIO(data)).convert("RGBA")))


def test_16bit_keeps_high_byte():
    a = RNG.integers(0, 65536, (11, 13), dtype=np.uint16)
    img = Image.fromarray(a, "I;16")
    got = ist.decode_png(_png(img))
    assert np.array_equal(got[..., 0], (a >> 8).astype(np.uint8)) and (got[..., 3] == 255).all()


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))


def test_every_filter_type_by_hand():
    """Rows filtered by hand with each of the five predictors (PIL picks filters adaptively; this forces them)."""
    h, w = 10, 14
    px = RNG.integers(0, 256, (h, w, 4), dtype=np.uint8)
    raw, prev = b"", np.zeros(w * 4, np.int32)
    for y in range(h):
        cur = px[y].reshape(-1).astype(np.int32)
        ft = y % 5
        left = np.concatenate([np.zeros(4, np.int32), cur[:-4]])
        upleft = np.concatenate([np.zeros(4, np.int32), prev[:-4]])
        if ft == 0: f = cur
        elif ft == 1: f = cur - left
        elif ft == 2: f = cur - prev
        elif ft == 3: f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            f = cur - pred
        raw += bytes([ft]) + (f & 255).astype(np.uint8).tobytes()
        prev = cur
    comp = zlib.compress(raw)
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + _chunk(b"IDAT", comp[:20]) + _chunk(b"IDAT", comp[20:]) + _chunk(b"IEND", b"")
    assert np.array_equal(ist.decode_png(png), px)
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(png)).convert("RGBA")), px)      # PIL agrees the file is valid


def test_errors_are_reported_like_decode_failures():
    good = _png(Image.fromarray(RNG.integers(0, 256, (4, 4, 4), dtype=np.uint8), "RGBA"))
    with pytest.raises(ist.StitchError) as e:
        ist.decode_png(b"\xff\xd8\xff\xe0" + b"0" * 100)
    assert e.value.code == -7 and "JPEG" in str(e.value)
    with pytest.raises(ist.StitchError) as e:
        ist.decode_png(good[:40])
    assert e.value.code == -6
    bad = bytearray(good)
    bad[50] ^= 0xFF
    with pytest.raises(ist.StitchError) as e:
        ist.decode_png(bytes(bad))
    assert e.value.code == -6 and "CRC" in str(e.value)
    inter = _png(Image.fromarray(RNG.integers(0, 256, (8, 8, 3), dtype=np.uint8), "RGB"))
    inter = bytearray(inter); inter[28] = 1
    inter[29:33] = struct.pack(">I", zlib.crc32(bytes(inter[12:29])))
    with pytest.raises(ist.StitchError) as e:          # flagged interlaced, but the data is one plain image
        ist.decode_png(bytes(inter))
    assert e.value.code == -6


def _adam7_png(a, ctype, filters=(0, 1, 2, 3, 4)):
    """An Adam7-interlaced 8-bit PNG of array a (H x W x C) written by hand (PIL cannot write one): seven sub-images,
    every scanline with one of the five filter types."""
    h, w, c = a.shape
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    raw = bytearray()
    k = 0
    for x0, y0, dx, dy in passes:
        sub = a[y0::dy, x0::dx]
        if sub.size == 0:
            continue
        prev = np.zeros(sub.shape[1] * c, np.int32)
        for row in sub.reshape(sub.shape[0], -1).astype(np.int32):
            ft = filters[k % len(filters)]
            k += 1
            left = np.concatenate([np.zeros(c, np.int32), row[:-c]])
            ul = np.concatenate([np.zeros(c, np.int32), prev[:-c]])
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = left
            elif ft == 2:
                pred = prev
            elif ft == 3:
                pred = (left + prev) >> 1
            else:
                pp = left + prev - ul
                pa, pb, pc = abs(pp - left), abs(pp - prev), abs(pp - ul)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            raw.append(ft)
            raw += ((row - pred) & 255).astype(np.uint8).tobytes()
            prev = row

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    ihdr = struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 1)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b"")


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (8, 8), (9, 9), (33, 17), (2, 70)])
def test_adam7_interlaced_png(shape):
    h, w = shape
    rgba = RNG.integers(0, 256, (h, w, 4), dtype=np.uint8)
    data = _adam7_png(rgba, 6)
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGBA")), rgba)      # the hand-written file is valid
    assert np.array_equal(ist.decode_png(data), rgba)
    rgb = RNG.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = ist.decode_png(_adam7_png(rgb, 2))
    assert np.array_equal(got[..., :3], rgb) and (got[..., 3] == 255).all()
    grey = RNG.integers(0, 256, (h, w, 1), dtype=np.uint8)
    got = ist.decode_png(_adam7_png(grey, 0))
    assert np.array_equal(got[..., 0], grey[..., 0]) and np.array_equal(got[..., 1], got[..., 2])


def test_jpeg_header_and_exif_orientation_on_cpu():
    """ist_image_info needs no GPU: size + EXIF orientation (what getImageInfo feeds the planner, index.js:734)."""
    for endian_o in (1, 3, 6, 8):
        exif = Image.Exif()
        exif[0x0112] = endian_o
        b = io.BytesIO()
        Image.fromarray(RNG.integers(0, 256, (21, 34, 3), dtype=np.uint8)).save(b, "JPEG", exif=exif)
        assert ist.image_info(b.getvalue()) == (34, 21, endian_o)
    b = io.BytesIO()
    Image.fromarray(RNG.integers(0, 256, (5, 7, 3), dtype=np.uint8)).save(b, "JPEG")
    assert ist.image_info(b.getvalue()) == (7, 5, 0)
    b = io.BytesIO()
    Image.fromarray(RNG.integers(0, 256, (5, 7, 4), dtype=np.uint8), "RGBA").save(b, "PNG")
    assert ist.image_info(b.getvalue()) == (7, 5, 0)


def _save(img, fmt, **kw):
    b = io.BytesIO()
    img.save(b, fmt, **kw)
    return b.getvalue()


def test_bmp_decode_matches_pil():
    rgb = Image.fromarray(RNG.integers(0, 256, (19, 31, 3), dtype=np.uint8), "RGB")          # odd width: row padding
    for img in (rgb, rgb.convert("L"), rgb.convert("P", palette=Image.ADAPTIVE, colors=16), rgb.convert("1"),
                Image.fromarray(RNG.integers(0, 256, (7, 5, 4), dtype=np.uint8), "RGBA")):
        data = _save(img, "BMP")
        ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"))
        got = ist.decode_image(data)
        assert ist.image_info(data)[:2] == (img.width, img.height)
        assert np.array_equal(got, ref), img.mode


def test_gif_first_frame_matches_pil():
    idx = RNG.integers(0, 200, (33, 47), dtype=np.uint8)
    img = Image.fromarray(idx, "P")
    img.putpalette(RNG.integers(0, 256, 768, dtype=np.uint8).tolist())
    for kw in ({}, {"transparency": 7}, {"interlace": True}):
        data = _save(img, "GIF", **kw)
        ref = np.array(Image.open(io.BytesIO(data)).convert("RGBA"))
        ref[ref[..., 3] == 0] = 0          # a canvas holds transparent pixels as (0,0,0,0); PIL keeps the palette colour
        assert np.array_equal(ist.decode_image(data), ref), kw
    # a smooth image (long LZW strings, code-size growth up to 12 bits, dictionary resets)
    yy, xx = np.mgrid[0:300, 0:400]
    big = Image.fromarray(((xx // 3 + yy // 5) % 256).astype(np.uint8), "P")
    big.putpalette(list(range(256)) * 3)
    data = _save(big, "GIF")
    assert np.array_equal(ist.decode_image(data), np.asarray(Image.open(io.BytesIO(data)).convert("RGBA")))
    noise = Image.fromarray(RNG.integers(0, 256, (120, 130), dtype=np.uint8), "P")
    noise.putpalette(RNG.integers(0, 256, 768, dtype=np.uint8).tolist())
    data = _save(noise, "GIF")
    assert np.array_equal(ist.decode_image(data), np.asarray(Image.open(io.BytesIO(data)).convert("RGBA")))


def test_webp_and_garbage_are_named():
    with pytest.raises(ist.StitchError) as e:
        ist.decode_image(b"RIFF\x00\x00\x00\x00WEBPVP8 " + b"\x00" * 64)            # a RIFF/WEBP header with no chunk inside
    assert e.value.code == -6 and "WebP" in str(e.value)
    with pytest.raises(ist.StitchError) as e:
        ist.decode_png(b"RIFF\x00\x00\x00\x00WEBPVP8 " + b"\x00" * 64)              # the PNG-only entry point names the format
    assert e.value.code == -7 and "WebP" in str(e.value)
    with pytest.raises(ist.StitchError):
        ist.decode_image(b"GIF89a" + b"\x00" * 3)


def _rle_bmp(idx, bpp, palette, with_delta=False):
    """a BI_RLE8 / BI_RLE4 file of the index image `idx` (bottom-up rows): encoded runs, literal (absolute) runs with their
    16-bit padding, end-of-line / end-of-bitmap escapes and - optionally - a delta escape that skips pixels"""
    import struct
    h, w = idx.shape
    body = bytearray()
    for r, row in enumerate(idx[::-1]):
        x = 0
        if with_delta and r == 0:             # bottom row: 5 single pixels, then a delta escape that skips 2 pixels (they stay palette entry 0)
            for k in range(5):
                body += bytes([1, int(row[k]) if bpp == 8 else int(row[k]) << 4])
            body += b"\x00\x02\x02\x00"
            x = 7
        while x < w:
            run = 1
            while x + run < w and run < 255 and row[x + run] == row[x] and bpp == 8:
                run += 1
            if bpp == 4:                      # RLE4 runs alternate two nibbles: use runs of one colour pair
                run = 1
                while x + run < w and run < 254 and row[x + run] == row[x + (run & 1)]:
                    run += 1
            if run >= 3:
                v = int(row[x]) if bpp == 8 else (int(row[x]) << 4 | int(row[x + 1] if run > 1 else 0))
                body += bytes([run, v])
                x += run
            else:
                n = min(w - x, 7)
                if bpp == 4:
                    n &= ~1                   # (PIL reads n // 2 bytes of an RLE4 literal run: an odd run is not a usable witness)
                if n < 3:
                    n = min(w - x, 2)                     # literal runs hold at least 3 pixels: emit short runs instead
                    for k in range(n):
                        body += bytes([1, int(row[x + k]) if bpp == 8 else int(row[x + k]) << 4])
                else:
                    body += bytes([0, n])
                    if bpp == 8:
                        lit = bytes(int(v) for v in row[x:x + n])
                    else:
                        px = [int(v) for v in row[x:x + n]] + [0]
                        lit = bytes((px[2 * k] << 4) | px[2 * k + 1] for k in range((n + 1) // 2))
                    body += lit + (b"\x00" if len(lit) & 1 else b"")
                x += n
        body += b"\x00\x00"                   # end of line
    body += b"\x00\x01"                       # end of bitmap
    ncol = len(palette) // 3
    pal = b"".join(bytes([palette[3 * i + 2], palette[3 * i + 1], palette[3 * i], 0]) for i in range(ncol))
    off = 14 + 40 + len(pal)
    dib = struct.pack("<IiiHHIIiiII", 40, w, h, 1, bpp, 1 if bpp == 8 else 2, len(body), 2835, 2835, ncol, 0)
    return b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + dib + pal + bytes(body)


def test_run_length_bmp_matches_pil():
    """BI_RLE8 / BI_RLE4 (VERDICT r02 'niche inputs the platform decoder accepts', index.js:4): runs, literal runs, escapes"""
    for bpp, ncol in ((8, 200), (4, 16)):
        pal = RNG.integers(0, 256, 3 * ncol, dtype=np.uint8).tolist()
        flat = np.repeat(RNG.integers(0, ncol, (23, 9), dtype=np.uint8), 5, axis=1)[:, :41]      # runs of 5
        noisy = RNG.integers(0, ncol, (17, 30), dtype=np.uint8)
        for idx in (flat, noisy):
            for delta in (False, True):
                data = _rle_bmp(idx, bpp, pal, with_delta=delta)
                ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"))
                assert ist.image_info(data)[:2] == (idx.shape[1], idx.shape[0])
                assert np.array_equal(ist.decode_image(data), ref), (bpp, delta)
    # RLE4 literal runs with an ODD pixel count (padded nibble + 16-bit alignment), against the palette directly
    import struct
    pal16 = RNG.integers(0, 256, 48, dtype=np.uint8).tolist()
    row = [7, 10, 0, 10, 6, 15, 1, 12, 4, 13, 0, 3]
    body = bytes([0, 7, 0x7A, 0x0A, 0x6F, 0x10, 0, 5, 0xC4, 0xD0, 0x30, 0x00, 0, 0, 0, 1])
    palb = b"".join(bytes([pal16[3 * i + 2], pal16[3 * i + 1], pal16[3 * i], 0]) for i in range(16))
    off = 14 + 40 + 64
    odd = b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + struct.pack("<IiiHHIIiiII", 40, 12, 1, 1, 4, 2, len(body), 2835, 2835, 16, 0) + palb + body
    want = np.array([[pal16[3 * v:3 * v + 3] + [255] for v in row]], np.uint8)
    assert np.array_equal(ist.decode_image(odd), want)
    # truncated in the middle of a literal run
    data = _rle_bmp(noisy, 8, pal)
    with pytest.raises(ist.StitchError):
        ist.decode_image(data[:len(data) - 40] if data[-41] != 0 else data[:len(data) - 39])
