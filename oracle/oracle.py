"""ctypes front-end of the CPU oracle (oracle/ist_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product (imagestitching_amd/, node/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# IST_ORACLE_LIB: bench.py's cpu_baseline leg points this at the -march=native build it made on the box it runs on
_LIB_PATH = os.environ.get("IST_ORACLE_LIB") or os.path.join(_HERE, "libist_oracle.so")

VERTICAL, HORIZONTAL = 0, 1
MODE = {"min": 0, "max": 1, "original": 2}
PLATFORM = {"other": 0, "devtools": 0, "windows": 0, "mac": 0, "ios": 1, "android": 2}
NEAREST, BILINEAR, AREA = 0, 1, 2
EDGE_AA = 0x100     # OR-ed into a filter: coverage anti-aliasing of fractional rectangle edges


class Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("orientation", C.c_int32),
                ("bmp_w", C.c_int32), ("bmp_h", C.c_int32), ("file_size", C.c_int64)]


class Limits(C.Structure):
    _fields_ = [("platform", C.c_int32), ("max_side", C.c_double), ("max_pixels", C.c_double),
                ("max_super_sample", C.c_double)]


class Rect(C.Structure):
    _fields_ = [("image", C.c_int32), ("orientation", C.c_int32),
                ("dx", C.c_double), ("dy", C.c_double), ("dw", C.c_double), ("dh", C.c_double)]


class Plan(C.Structure):
    _fields_ = [("out_w", C.c_double), ("out_h", C.c_double), ("scale_down", C.c_double),
                ("super_sample", C.c_double), ("canvas_w", C.c_double), ("canvas_h", C.c_double),
                ("big_task", C.c_int32), ("n_rects", C.c_int32)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("image", C.c_int32), ("m", C.c_double * 6), ("s", C.c_double * 4),
                ("d", C.c_double * 4), ("rgba", C.c_uint8 * 4), ("pad", C.c_int32)]


def build(force=False):
    if os.environ.get("IST_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "ist_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_default_limits.argtypes = [C.c_int, C.POINTER(Limits)]
        L.orc_plan_compute.argtypes = [C.POINTER(Image), C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(Limits),
                                       C.POINTER(Plan), C.POINTER(Rect)]
        L.orc_plan_compute.restype = C.c_int
        L.orc_render.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(Rect), C.c_int, C.POINTER(Image),
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_void_p, C.c_size_t, C.c_int]
        L.orc_render.restype = C.c_int
        L.orc_render_ops.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(Op), C.c_int, C.POINTER(Image),
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_void_p, C.c_size_t]
        L.orc_render_ops.restype = C.c_int
        L.orc_resolve_draw.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_resolve_draw.restype = C.c_int
        L.orc_orientation_ctm.argtypes = [C.c_double] * 5 + [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def make_limits(platform="other", max_side=0.0, max_pixels=0.0, max_super_sample=0.0):
    return Limits(PLATFORM[platform] if isinstance(platform, str) else platform, float(max_side), float(max_pixels),
                  float(max_super_sample))


def default_limits(platform):
    """What the reference page holds after onLoad with empty storage (index.js:126-156)."""
    out = Limits()
    lib().orc_default_limits(PLATFORM[platform], C.byref(out))
    return out


def lifted_limits(max_super_sample=0.0):
    """The reference's own knob for lifting the caps: storage canvasLimit={size:1048576,pixels:2**40}."""
    return Limits(0, 1048576.0, float(2 ** 40), float(max_super_sample))


def _images(descs):
    arr = (Image * max(1, len(descs)))()
    for i, d in enumerate(descs):
        arr[i] = Image(int(d["width"]), int(d["height"]), int(d.get("orientation", 1) or 0),
                       int(d.get("bmp_w", 0)), int(d.get("bmp_h", 0)), int(d.get("file_size", 0)))
    return arr


def plan(descs, direction, mode="min", gap=0.0, limits=None):
    """Returns (rc, plan dict, rect list).  descs: [{'width','height','orientation'?,'file_size'?,'bmp_w'?,'bmp_h'?}]"""
    limits = limits if limits is not None else lifted_limits(1.0)
    n = len(descs)
    imgs = _images(descs)
    p = Plan()
    rects = (Rect * max(1, n))()
    d = {"vertical": 0, "horizontal": 1}[direction] if isinstance(direction, str) else direction
    rc = lib().orc_plan_compute(imgs, n, d, MODE[mode] if isinstance(mode, str) else mode, float(gap),
                                C.byref(limits), C.byref(p), rects)
    if rc != 0:
        return rc, None, []
    pd = {k: getattr(p, k) for k, _ in Plan._fields_}
    rl = [{"image": r.image, "orientation": r.orientation, "dx": r.dx, "dy": r.dy, "dw": r.dw, "dh": r.dh}
          for r in rects[:n]]
    return 0, pd, rl


def _filter(filter, edge_aa=False):
    f = {"nearest": 0, "bilinear": 1, "area": 2}[filter] if isinstance(filter, str) else int(filter)
    return f | (EDGE_AA if edge_aa else 0)


def render(pd, rl, descs, pixels, filter="bilinear", threads=1, out=None, edge_aa=False):
    """Render a plan with the oracle raster.  pixels: list of HxWx4 uint8 arrays (C-contiguous rows)."""
    n = len(rl)
    cw, ch = int(pd["canvas_w"]), int(pd["canvas_h"])
    rects = (Rect * max(1, n))()
    for i, r in enumerate(rl):
        rects[i] = Rect(r["image"], r["orientation"], r["dx"], r["dy"], r["dw"], r["dh"])
    imgs = _images(descs)
    ptrs = (C.c_void_p * max(1, len(pixels)))()
    pitches = (C.c_size_t * max(1, len(pixels)))()
    keep = []
    for i, a in enumerate(pixels):
        a = np.ascontiguousarray(a) if a.strides[-1] != 1 or a.strides[-2] != 4 else a
        keep.append(a)
        ptrs[i] = a.ctypes.data
        pitches[i] = a.strides[0]
    if out is None:
        out = np.empty((ch, cw, 4), np.uint8)
    f = _filter(filter, edge_aa)
    rc = lib().orc_render(cw, ch, float(pd["super_sample"]), rects, n, imgs, ptrs, pitches, f,
                          out.ctypes.data, out.strides[0], int(threads))
    if rc != 0:
        raise RuntimeError("oracle render failed rc=%d" % rc)
    return out


def stitch(pixels, direction, mode="min", gap=0.0, limits=None, filter="bilinear", orientations=None, threads=1, edge_aa=False):
    """The restated onStitch stages 2-5 for decoded RGBA8 inputs."""
    descs = [{"width": a.shape[1], "height": a.shape[0], "orientation": (orientations[i] if orientations else 1)}
             for i, a in enumerate(pixels)]
    rc, pd, rl = plan(descs, direction, mode, gap, limits)
    if rc != 0:
        raise RuntimeError("oracle plan rc=%d" % rc)
    return render(pd, rl, descs, pixels, filter, threads, edge_aa=edge_aa), pd, rl


def render_ops(canvas_w, canvas_h, ops, descs, pixels, filter="bilinear", clear=(0, 0, 0, 0), edge_aa=False):
    """ops: [{'kind':'fill','m':[6],'rect':[x,y,w,h],'rgba':(r,g,b,a)} | {'kind':'draw','image':i,'m':[6],'s':[4],'d':[4]}]"""
    arr = (Op * max(1, len(ops)))()
    for i, o in enumerate(ops):
        op = Op()
        op.m[:] = o["m"]
        if o["kind"] == "fill":
            op.kind = 0
            op.d[:] = o["rect"]
            op.rgba[:] = o["rgba"]
        else:
            op.kind = 1
            op.image = o["image"]
            op.s[:] = o["s"]
            op.d[:] = o["d"]
        arr[i] = op
    imgs = _images(descs)
    ptrs = (C.c_void_p * max(1, len(pixels)))()
    pitches = (C.c_size_t * max(1, len(pixels)))()
    keep = [np.ascontiguousarray(a) for a in pixels]
    for i, a in enumerate(keep):
        ptrs[i] = a.ctypes.data
        pitches[i] = a.strides[0]
    out = np.empty((canvas_h, canvas_w, 4), np.uint8)
    clr = (C.c_uint8 * 4)(*clear)
    f = _filter(filter, edge_aa)
    rc = lib().orc_render_ops(canvas_w, canvas_h, clr, arr, len(ops), imgs, ptrs, pitches, f, out.ctypes.data, out.strides[0])
    if rc != 0:
        raise RuntimeError("oracle render_ops failed rc=%d" % rc)
    return out


def resolve_draw(m, cw, ch, img_w, img_h, s, d):
    k = (C.c_double * 4)()
    box = (C.c_int * 4)()
    clamp = (C.c_int * 4)()
    swap = C.c_int()
    rc = lib().orc_resolve_draw((C.c_double * 6)(*m), cw, ch, img_w, img_h, (C.c_double * 4)(*s), (C.c_double * 4)(*d),
                                k, box, clamp, C.byref(swap))
    if rc != 0:
        return rc, None
    return 0, {"k": list(k), "box": list(box), "clamp": list(clamp), "swap": swap.value}


def orientation_ctm(ss, dx, dy, dw, dh, orientation):
    m = (C.c_double * 6)()
    r = (C.c_double * 4)()
    lib().orc_orientation_ctm(ss, dx, dy, dw, dh, orientation, m, r)
    return list(m), list(r)


def synth_image(k, h, w, opaque=True):
    """BASELINE.md section 3 synthetic input: image k = default_rng(1000+k) uniform bytes, alpha forced to 255."""
    a = np.random.default_rng(1000 + k).integers(0, 256, (h, w, 4), dtype=np.uint8)
    if opaque:
        a[..., 3] = 255
    return a
