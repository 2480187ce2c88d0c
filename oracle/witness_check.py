#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — pins the oracle's pixel arithmetic against independent witnesses and mints fixtures.

The reference pins no pixels (its raster is the closed WeChat client; it has no tests).  The oracle restates the HTML
Canvas drawImage contract; this script checks that restatement against three independent implementations that are
available in the authoring container only:
    cairo 1.16 / pixman (libcairo.so.2 through ctypes) — a real Canvas-2D raster backend (FILTER_NEAREST / FILTER_BILINEAR,
                                                          EXTEND_PAD)
    torch.nn.functional.interpolate(align_corners=False, antialias=False)
    PIL.Image.transform(AFFINE, BILINEAR / NEAREST)
and writes tests/golden/pixel_fixtures.npz (inputs, oracle outputs, witness outputs) so that the CPU test-suite can
re-check the oracle anywhere without those libraries.  Run:  python oracle/witness_check.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[0] = ROOT          # not the script directory: "oracle" must resolve to the package, not oracle.py
from oracle import oracle as O  # noqa: E402

CAIRO_FORMAT_ARGB32, FILTER_NEAREST, FILTER_BILINEAR, EXTEND_PAD = 0, 3, 4, 3


def cairo_scale_draw(img, dw, dh, bilinear):
    """Draw opaque RGBA `img` scaled to dw x dh with cairo (identity CTM + pattern matrix), return RGBA."""
    lib = C.CDLL("libcairo.so.2")
    lib.cairo_image_surface_create_for_data.restype = C.c_void_p
    lib.cairo_image_surface_create_for_data.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.cairo_image_surface_create.restype = C.c_void_p
    lib.cairo_image_surface_create.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.cairo_create.restype = C.c_void_p
    lib.cairo_create.argtypes = [C.c_void_p]
    lib.cairo_get_source.restype = C.c_void_p
    lib.cairo_get_source.argtypes = [C.c_void_p]
    for f, a in (("cairo_scale", [C.c_void_p, C.c_double, C.c_double]), ("cairo_set_source_surface", [C.c_void_p, C.c_void_p, C.c_double, C.c_double]),
                 ("cairo_pattern_set_filter", [C.c_void_p, C.c_int]), ("cairo_pattern_set_extend", [C.c_void_p, C.c_int]),
                 ("cairo_rectangle", [C.c_void_p] + [C.c_double] * 4), ("cairo_fill", [C.c_void_p]), ("cairo_paint", [C.c_void_p]),
                 ("cairo_set_source_rgb", [C.c_void_p] + [C.c_double] * 3), ("cairo_surface_flush", [C.c_void_p]),
                 ("cairo_destroy", [C.c_void_p]), ("cairo_surface_destroy", [C.c_void_p])):
        getattr(lib, f).argtypes = a
        getattr(lib, f).restype = None
    lib.cairo_image_surface_get_data.restype = C.POINTER(C.c_uint8)
    lib.cairo_image_surface_get_data.argtypes = [C.c_void_p]
    lib.cairo_image_surface_get_stride.restype = C.c_int
    lib.cairo_image_surface_get_stride.argtypes = [C.c_void_p]
    sh, sw = img.shape[:2]
    bgra = np.ascontiguousarray(img[..., [2, 1, 0, 3]])
    src = lib.cairo_image_surface_create_for_data(bgra.ctypes.data, CAIRO_FORMAT_ARGB32, sw, sh, sw * 4)
    dst = lib.cairo_image_surface_create(CAIRO_FORMAT_ARGB32, dw, dh)
    cr = lib.cairo_create(dst)
    lib.cairo_set_source_rgb(cr, 1.0, 1.0, 1.0)
    lib.cairo_paint(cr)
    lib.cairo_scale(cr, dw / sw, dh / sh)
    lib.cairo_set_source_surface(cr, src, 0.0, 0.0)
    pat = lib.cairo_get_source(cr)
    lib.cairo_pattern_set_filter(pat, FILTER_BILINEAR if bilinear else FILTER_NEAREST)
    lib.cairo_pattern_set_extend(pat, EXTEND_PAD)
    lib.cairo_rectangle(cr, 0.0, 0.0, float(sw), float(sh))
    lib.cairo_fill(cr)
    lib.cairo_surface_flush(dst)
    stride = lib.cairo_image_surface_get_stride(dst)
    data = np.ctypeslib.as_array(lib.cairo_image_surface_get_data(dst), shape=(dh, stride))
    out = data[:, :dw * 4].reshape(dh, dw, 4)[..., [2, 1, 0, 3]].copy()
    lib.cairo_destroy(cr)
    lib.cairo_surface_destroy(dst)
    lib.cairo_surface_destroy(src)
    return out


def torch_scale(img, dw, dh, bilinear):
    import torch
    import torch.nn.functional as F
    t = torch.from_numpy(img[..., :3].astype(np.float32)).permute(2, 0, 1)[None]
    if bilinear:
        o = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)
    else:
        o = F.interpolate(t, size=(dh, dw), mode="nearest-exact")
    o = torch.floor(o + 0.5).clamp(0, 255).to(torch.uint8)[0].permute(1, 2, 0).numpy()
    return np.concatenate([o, np.full((dh, dw, 1), 255, np.uint8)], -1)


def pil_scale(img, dw, dh, bilinear):
    from PIL import Image
    sh, sw = img.shape[:2]
    im = Image.fromarray(img[..., :3], "RGB")
    # output (x,y) centre -> input: ((x+0.5)*sw/dw - 0.5); PIL's AFFINE maps output pixel centres with a +0.5/-0.5 convention
    o = im.transform((dw, dh), Image.AFFINE, (sw / dw, 0, 0, 0, sh / dh, 0), Image.BILINEAR if bilinear else Image.NEAREST)
    o = np.asarray(o)
    return np.concatenate([o, np.full((dh, dw, 1), 255, np.uint8)], -1)


def oracle_scale(img, dw, dh, bilinear):
    sh, sw = img.shape[:2]
    ops = [{"kind": "fill", "m": [1, 0, 0, 1, 0, 0], "rect": [0, 0, dw, dh], "rgba": (255, 255, 255, 255)},
           {"kind": "draw", "image": 0, "m": [1, 0, 0, 1, 0, 0], "s": [0, 0, sw, sh], "d": [0, 0, dw, dh]}]
    return O.render_ops(dw, dh, ops, [{"width": sw, "height": sh}], [img], "bilinear" if bilinear else "nearest")


def main():
    rng = np.random.default_rng(7)
    cases = []
    row = np.zeros((1, 4, 4), np.uint8)
    row[0, :, :3] = np.array([0, 100, 200, 50])[:, None]
    row[..., 3] = 255
    cases.append(("survey_row_4_to_8", row, 8, 1))
    for name, (sw, sh), (dw, dh) in [("up_4_3", (30, 23), (40, 31)), ("down_0_75", (40, 32), (30, 24)), ("down_0_45", (91, 67), (41, 30)),
                                     ("identity", (33, 21), (33, 21)), ("up_2x", (16, 12), (32, 24)), ("aniso", (50, 20), (35, 44))]:
        img = rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)
        img[..., 3] = 255
        cases.append((name, img, dw, dh))
    store, report = {}, {}
    for name, img, dw, dh in cases:
        for bil in (False, True):
            key = "%s_%s" % (name, "bilinear" if bil else "nearest")
            ora = oracle_scale(img, dw, dh, bil)
            wit = {"cairo": cairo_scale_draw(img, dw, dh, bil), "torch": torch_scale(img, dw, dh, bil), "pil": pil_scale(img, dw, dh, bil)}
            store[key + "__in"] = img
            store[key + "__oracle"] = ora
            rep = {}
            # nearest: a sample point that falls EXACTLY on a source pixel edge is a tie; rasters legitimately differ there
            sh_, sw_ = img.shape[:2]
            tie_x = np.array([((2 * x + 1) * sw_) % (2 * dw) == 0 for x in range(dw)])
            tie_y = np.array([((2 * y + 1) * sh_) % (2 * dh) == 0 for y in range(dh)])
            not_tie = ~(tie_y[:, None] | tie_x[None, :])
            store[key + "__not_tie"] = not_tie
            for w, arr in wit.items():
                store[key + "__" + w] = arr
                d = np.abs(arr[..., :3].astype(int) - ora[..., :3].astype(int))
                rep[w] = {"max": int(d.max()), "frac_ne": round(float((d > 0).mean()), 5)}
                if not bil:
                    rep[w]["max_off_ties"] = int(d[not_tie].max()) if not_tie.any() else 0
            report[key] = rep
            print(key, rep)
    out = os.path.join(ROOT, "tests", "golden", "pixel_fixtures.npz")
    np.savez_compressed(out, **store)
    with open(os.path.join(ROOT, "tests", "golden", "pixel_witness_report.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
