"""The flat form of a job, replayed on the CPU (no GPU): ist_debug_flat_form hands out the cells the kernel will walk when a strip of whole
dense rows is launched (csrc/ist_compile.cpp compile_flat_twin).  Replaying them in numpy must make exactly the bytes
the op list itself describes - every byte of the rendered rows written once, none outside, holes untouched - for random strips, gaps,
source crops, holes and whole-row clips.  Reference anchor: the vertical strip loop, pages/index/index.js:1522-1538."""
import ctypes as C

import numpy as np
import pytest
from imagestitching_amd import _lib as L

SENTINEL = 0xA7


def _op(kind, image, w, dy, dh, sy=0, rgba=(0, 0, 0, 0)):
    o = L.Op()
    o.kind, o.image = kind, image
    o.m[0] = o.m[3] = 1.0
    o.s[0], o.s[1], o.s[2], o.s[3] = 0, sy, w, dh
    o.d[0], o.d[1], o.d[2], o.d[3] = 0, dy, w, dh
    for k in range(4):
        o.rgba[k] = rgba[k]
    return o


def _flat_form(w, h, ops, descs, n_images, clip=None, clear=(0, 0, 0, 0)):
    arr = (L.Op * len(ops))(*ops)
    clr = (C.c_uint8 * 4)(*clear)
    region = C.byref(L.Region(*clip)) if clip else None
    pitch, off, n = C.c_int64(0), C.c_int64(0), C.c_int(0)
    cells = (L.FlatCell * 4096)()
    L.check(L.lib.ist_debug_flat_form(w, h, clr, arr, len(ops), descs, n_images, 1, region, C.byref(pitch), C.byref(off), cells, 4096, C.byref(n)))
    return pitch.value, off.value, [cells[k] for k in range(n.value)]


def _replay(cells, pitch, off, canvas_flat, srcs):
    """what the launch would do: canvas_flat / srcs[i] are flat uint8 arrays; returns a per-byte write count"""
    hits = np.zeros(canvas_flat.size, np.uint8)
    for c in cells:
        nb = (c.X1 - c.X0) * 4
        for y in range(c.Y0, c.Y1):
            d0 = off + y * pitch + c.X0 * 4
            assert 0 <= d0 and d0 + nb <= canvas_flat.size, "a cell reaches outside the destination"
            if c.path == 0:
                canvas_flat[d0:d0 + nb] = np.tile(np.frombuffer(np.uint32(c.bg).tobytes(), np.uint8), nb // 4)
            else:
                assert c.path == 1 and c.opaque == 1
                s0 = c.src_offset + (y - c.Y0) * pitch
                src = srcs[c.image]
                assert 0 <= s0 and s0 + nb <= src.size, "a cell reads outside its source"
                canvas_flat[d0:d0 + nb] = src[s0:s0 + nb]
            hits[d0:d0 + nb] += 1
    return hits


def _expected(w, h, ops, srcs2d, clip_rows):
    out = np.full((h, w, 4), SENTINEL, np.uint8)
    painted = np.zeros(h, bool)
    for o in ops:
        dy, dh = int(o.d[1]), int(o.d[3])
        if o.kind == 0:
            out[dy:dy + dh] = np.array(list(o.rgba), np.uint8)
            painted[dy:dy + dh] = True
        elif o.kind == 1:
            sy = int(o.s[1])
            out[dy:dy + dh] = srcs2d[o.image][sy:sy + dh]
            painted[dy:dy + dh] = True
        else:
            out[dy:dy + dh] = SENTINEL
            painted[dy:dy + dh] = False
    y0, y1 = clip_rows
    out[:y0] = SENTINEL
    out[y1:] = SENTINEL
    painted[:y0] = False
    painted[y1:] = False
    return out, painted


@pytest.mark.parametrize("seed", range(48))
def test_replaying_the_flat_form_makes_the_op_lists_bytes(seed):
    rng = np.random.default_rng(4000 + seed)
    w = int(rng.integers(300, 2600))
    if (w * 4) % 16384 == 0:
        w += 1
    n_images = int(rng.integers(1, 5))
    heights = [int(rng.integers(1, 900)) for _ in range(n_images)]
    srcs2d = [rng.integers(0, 256, (hh, w, 4), dtype=np.uint8) for hh in heights]
    descs = (L.ImageDesc * n_images)()
    for i, hh in enumerate(heights):
        descs[i].width, descs[i].height, descs[i].opaque = w, hh, 1
    # a strip: pieces in canvas order, each a draw (a crop of some image), a fill band or a hole
    pieces, y = [], 0
    while y * w * 4 < (7 if seed % 2 else 3) * (1 << 20) or len(pieces) < 3:
        kind = int(rng.choice([1, 1, 1, 0, 2]))
        if kind == 1:
            i = int(rng.integers(0, n_images))
            sy = int(rng.integers(0, heights[i]))
            dh = int(rng.integers(1, heights[i] - sy + 1))
            pieces.append(_op(1, i, w, y, dh, sy=sy))
        else:
            dh = int(rng.integers(1, 61))
            pieces.append(_op(kind, -1, w, y, dh, rgba=(int(rng.integers(0, 256)), 7, 200, 255)))
        y += dh
    h = y
    ops = [_op(0, -1, w, 0, h, rgba=(255, 255, 255, 255))] + pieces
    clip = None
    clip_rows = (0, h)
    if seed % 2:
        a = int(rng.integers(0, h // 3 + 1))               # (two thirds of a 7 MiB canvas: most clipped cases keep a flat form)
        b = int(rng.integers(h - h // 3, h + 1))
        clip, clip_rows = (0, a, w, b - a), (a, b)
    pitch, off, cells = _flat_form(w, h, ops, descs, n_images, clip)
    rendered_bytes = (clip_rows[1] - clip_rows[0]) * w * 4
    if not cells:
        assert rendered_bytes < 64 * 32768          # the only reason a strip of whole rows has no flat form here
        return
    assert pitch == 32768 and off == clip_rows[0] * w * 4
    canvas = np.full(h * w * 4, SENTINEL, np.uint8)
    hits = _replay(cells, pitch, off, canvas, [s.reshape(-1) for s in srcs2d])
    want, painted = _expected(w, h, ops, srcs2d, clip_rows)
    assert np.array_equal(canvas.reshape(h, w, 4), want)
    per_row = hits.reshape(h, w * 4)
    assert (per_row[painted] == 1).all() and (per_row[~painted] == 0).all()      # every rendered byte exactly once, nothing else


def test_jobs_without_a_flat_form():
    w, hh = 700, 1200
    descs = (L.ImageDesc * 2)()
    for i in range(2):
        descs[i].width, descs[i].height, descs[i].opaque = w, hh, 1
    base = [_op(0, -1, w, 0, 2 * hh, rgba=(255, 255, 255, 255)), _op(1, 0, w, 0, hh), _op(1, 1, w, hh, hh)]
    assert _flat_form(w, 2 * hh, base, descs, 2)[2]                                  # the plain strip has one
    narrow = L.Op.from_buffer_copy(bytes(base[2]))
    narrow.d[2] = w - 1
    narrow.s[2] = w - 1                                                              # a draw that is not as wide as the canvas
    assert not _flat_form(w, 2 * hh, base[:2] + [narrow], descs, 2)[2]
    scaled = L.Op.from_buffer_copy(bytes(base[2]))
    scaled.s[3] = hh - 100                                                           # resampled vertically
    assert not _flat_form(w, 2 * hh, base[:2] + [scaled], descs, 2)[2]
    turned = L.Op.from_buffer_copy(bytes(base[2]))
    turned.m[0], turned.m[3] = -1.0, -1.0                                            # any transform but the identity
    assert not _flat_form(w, 2 * hh, base[:2] + [turned], descs, 2)[2]
    assert not _flat_form(w, 2 * hh, base, descs, 2, clip=(5, 0, w - 5, 2 * hh))[2]    # a clip that does not take whole rows
    wide = (L.ImageDesc * 1)()
    wide[0].width, wide[0].height, wide[0].opaque = 4096, 900, 1
    assert not _flat_form(4096, 900, [_op(1, 0, 4096, 0, 900)], wide, 1)[2]           # rows of 16 KiB need no other walk
    wide[0].width = 1024
    assert _flat_form(1024, 900, [_op(1, 0, 1024, 0, 900)], wide, 1)[2]               # rows of 4 KiB do (stores 4 KiB apart: 0.76, 32 KiB apart: 0.87)
    small = [_op(0, -1, w, 0, 100, rgba=(1, 2, 3, 255)), _op(1, 0, w, 0, 100)]
    assert not _flat_form(w, 100, small, descs, 2)[2]                                  # too small for the pitch to matter


def test_a_large_strip_of_small_images_keeps_the_row_form():
    """every image boundary costs a head and a tail row of nearly empty tiles: measured (tools/exp_thin.py) the flat form loses below ~8 MB per
    image on strips large enough to leave the launch floor (64 MB)"""
    w = 2000

    def strip(n, hh):
        descs = (L.ImageDesc * n)()
        for i in range(n):
            descs[i].width, descs[i].height, descs[i].opaque = w, hh, 1
        ops = [_op(0, -1, w, 0, n * hh, rgba=(255, 255, 255, 255))] + [_op(1, i, w, i * hh, hh) for i in range(n)]
        return _flat_form(w, n * hh, ops, descs, n)[2]

    assert not strip(100, 100)          # 80 MB of 0.8 MB images
    assert strip(10, 1100)              # 88 MB of 8.8 MB images
    assert strip(50, 100)               # 40 MB: under the floor both forms cost the same; the flat one stays
