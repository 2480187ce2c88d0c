"""Host-side mirror of the reference's stitch surface, over the C-ABI (no pixel arithmetic in Python).

Reference: Page.onStitch (miniprogram-stitch/miniprogram/pages/index/index.js:1186-1633) reads
this.data.{images, direction, gap, verticalStitchMode, horizontalStitchMode}; here that is
`stitch(images, direction, opts)` with opts = {mode, gap, platform, maxSide, maxPixels, superSample, filter}.

Two ways in:
  stitch(images, direction, opts)      host RGBA8 arrays in, host RGBA8 array out   (ist_stitch_rgba8)
  Stitcher(device).compile(...)        device-resident: torch CUDA tensors in/out, one fused launch per call
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L

_DIRECTIONS = {"vertical": L.VERTICAL, "horizontal": L.HORIZONTAL}
_MODES = {"min": L.MODE_MIN, "max": L.MODE_MAX, "original": L.MODE_ORIGINAL}
_FILTERS = {"nearest": L.FILTER_NEAREST, "bilinear": L.FILTER_BILINEAR, "area": L.FILTER_AREA}
FILTER_EDGE_AA = 0x100


def edge_aa_of(o):
    """Coverage rule for fractional rectangle edges.  Unset: ON whenever a reference platform's plan is requested
    (opts.platform: the reference's own default behaviour scales the canvas by superSample 2.2 / 2.6 for fewer than 7
    images, index.js:1363,1426-1428, and leaves the cursor unrounded when gap > 0 and scaleDown < 1, :1432, so fractional
    edges are the normal case there and a Canvas raster anti-aliases them); OFF for the lifted MI355X default, whose
    plans have integer edges unless a gap meets a shrink, so that every output pixel is owned by exactly one image."""
    v = o.get("edgeAA")
    return (o.get("platform") is not None) if v is None else bool(v)


def _filter_of(o):
    return _FILTERS[o["filter"]] | (FILTER_EDGE_AA if edge_aa_of(o) else 0)
_PLATFORMS = {"ios": L.PLATFORM_IOS, "android": L.PLATFORM_ANDROID, "devtools": L.PLATFORM_OTHER,
              "windows": L.PLATFORM_OTHER, "mac": L.PLATFORM_OTHER, "other": L.PLATFORM_OTHER}

DEFAULT_OPTS = {
    "mode": "min",          # data.verticalStitchMode / horizontalStitchMode default (index.js:19-20)
    "gap": 0,               # data.gap default (index.js:17)
    "filter": "bilinear",   # imageSmoothingEnabled = true (index.js:1416-1418); 'nearest' = false; 'area' = opt-in box average of minified axes (IST_FILTER_AREA)
    "platform": None,       # None: MI355X default = caps lifted; 'ios' / 'android' / 'devtools' reproduce the phone caps
    "maxSide": None,        # deviceMaxCanvasSize override
    "maxPixels": None,      # deviceMaxCanvasPixels override
    "superSample": None,    # None: 1 when platform is None, reference rule (index.js:1363) otherwise
    "edgeAA": None,         # anti-alias fractional rectangle edges by area coverage (IST_FILTER_EDGE_AA); None: on iff `platform` is given (edge_aa_of)
    "pngLevel": None,       # PNG export form of the *_png / stitch_files calls: 0 stored, 1 compressed on the GPU; None = DEFAULT_PNG_LEVEL
    "devices": None,        # list of GPU indices (devices[0] = root): shard the stitch over them from this one process (ist_stitch_rgba8_multi)
    "split": "auto",        # with devices: "image" (image i -> devices[i mod n], BASELINE configs[3]), "band" (equal output rows per device, cut draw
                            # by draw), "rows" (device s owns a band of canvas rows across ALL draws: full-width bands for horizontal strips and
                            # centred rects too, index.js:1540-1553), "auto" = "image" when its parts are full-width (vertical min / max), else "rows"
}

_SPLITS = {"image": L.SPLIT_IMAGE, "band": L.SPLIT_BAND, "rows": L.SPLIT_ROWS, "auto": L.SPLIT_AUTO}

DEFAULT_PNG_LEVEL = 1


def _limits(opts):
    lim = L.Limits()
    if opts.get("platform") is None:
        L.lib.ist_limits_unlimited(C.byref(lim))
    else:
        L.lib.ist_limits_default(_PLATFORMS[opts["platform"]], C.byref(lim))
    if opts.get("maxSide") is not None:
        lim.max_side = float(opts["maxSide"])
    if opts.get("maxPixels") is not None:
        lim.max_pixels = float(opts["maxPixels"])
    if opts.get("superSample") is not None:
        lim.max_super_sample = float(opts["superSample"])
    return lim


def _merge(opts):
    o = dict(DEFAULT_OPTS)
    if opts:
        unknown = set(opts) - set(o)
        if unknown:
            raise TypeError("unknown stitch option(s): %s" % sorted(unknown))
        o.update(opts)
    return o


def _descs(images):
    """images: list of {'width','height','orientation'?,'fileSize'?,'opaque'?,'data'?} or HxWx4 uint8 arrays."""
    arr = (L.ImageDesc * max(1, len(images)))()
    for i, im in enumerate(images):
        if isinstance(im, dict):
            data = im.get("data")
            w = im.get("width", data.shape[1] if hasattr(data, "shape") else 0)
            h = im.get("height", data.shape[0] if hasattr(data, "shape") else 0)
            bw = im.get("bmpWidth", data.shape[1] if hasattr(data, "shape") and data.ndim == 3 else 0)
            bh = im.get("bmpHeight", data.shape[0] if hasattr(data, "shape") and data.ndim == 3 else 0)
            arr[i] = L.ImageDesc(int(w or 0), int(h or 0), int(im.get("orientation", 1) or 0), int(bw or 0), int(bh or 0),
                                 1 if im.get("opaque") else 0, int(im.get("fileSize", 0) or 0))
        else:
            arr[i] = L.ImageDesc(int(im.shape[1]), int(im.shape[0]), 1, 0, 0, 0, 0)
    return arr


class StitchPlan:
    """Result of the planner (index.js stage 2 + rect loop).  Owns the C plan; freed on garbage collection."""

    def __init__(self, cplan, descs, n_images):
        self._c = cplan
        self._descs = descs
        self.n_images = n_images
        self.out_w, self.out_h = cplan.out_w, cplan.out_h
        self.scale_down, self.super_sample = cplan.scale_down, cplan.super_sample
        self.canvas_w, self.canvas_h = int(cplan.canvas_w), int(cplan.canvas_h)
        self.big_task = bool(cplan.big_task)
        self.rects = [{"image": r.image, "orientation": r.orientation, "dx": r.dx, "dy": r.dy, "dw": r.dw, "dh": r.dh}
                      for r in cplan.rects[:cplan.n_rects]]

    def ops(self):
        """The Canvas call sequence (white fill + one drawImage per rect with its CTM) as C ops."""
        n = self._c.n_rects + 1
        arr = (L.Op * n)()
        cnt = C.c_int(0)
        L.check(L.lib.ist_plan_ops(C.byref(self._c), self._descs, self.n_images, arr, C.byref(cnt)))
        return arr, cnt.value

    def ops_as_dicts(self):
        arr, n = self.ops()
        return [{"kind": "fill" if o.kind == 0 else "draw", "image": o.image, "m": list(o.m), "s": list(o.s),
                 "d": list(o.d), "rgba": tuple(o.rgba)} for o in arr[:n]]

    def __del__(self):
        try:
            L.lib.ist_plan_free(C.byref(self._c))
        except Exception:
            pass


def plan(images, direction, opts=None):
    """Pure-CPU planner.  Returns a StitchPlan, or None when there is nothing to stitch (index.js:1189)."""
    o = _merge(opts)
    descs = _descs(images)
    cplan = L.Plan()
    lim = _limits(o)
    rc = L.check(L.lib.ist_plan_compute(descs, len(images), _DIRECTIONS[direction], _MODES[o["mode"]], float(o["gap"] or 0),
                                        C.byref(lim), C.byref(cplan)))
    if rc == L.IST_NOTHING_TO_DO:
        return None
    return StitchPlan(cplan, descs, len(images))


_ctx_cache = {}


def _drain_at_exit():
    """Interpreter exit: wait for whatever the cached contexts still have in flight and hand the idle pinned result blocks
    back, so that no DMA of the library is pending when the HIP runtime (or a profiler attached to it: rocprofv3's copy
    tracing waited 30 s for completion callbacks otherwise) shuts down.  Contexts are NOT destroyed: jobs that are garbage
    collected later still refer to them."""
    for c in list(_ctx_cache.values()):
        try:
            L.lib.ist_ctx_sync(c)
        except Exception:
            pass
    try:
        L.lib.ist_pool_trim()
    except Exception:
        pass


import atexit  # noqa: E402
atexit.register(_drain_at_exit)


def _ctx(device=0):
    c = _ctx_cache.get(device)
    if c is None:
        c = L.lib.ist_ctx_create(int(device))
        if not c:
            raise L.StitchError(-5, L.last_error())
        _ctx_cache[device] = c
    return c


def _ctx_png(device, level):
    """The context with its PNG export form set (ist_ctx_set_png_level): a per-context setting, so concurrent callers
    that want different forms on one device should serialise."""
    c = _ctx(device)
    L.check(L.lib.ist_ctx_set_png_level(c, int(DEFAULT_PNG_LEVEL if level is None else level)))
    return c


def stitch(images, direction, opts=None, device=0):
    """stitch(images, direction, opts) -> {'width', 'height', 'data'}: host arrays through the HIP path.

    images[i] = {'width', 'height', 'data': HxWx4 uint8 (RGBA, straight alpha), 'orientation'?: 1..8, 'fileSize'?}
    or simply an HxWx4 uint8 array.  Returns None when images is empty (the reference returns early).
    """
    o = _merge(opts)
    n = len(images)
    if n == 0:
        return None
    descs = _descs(images)
    keep, ptrs, pitches = [], (C.c_void_p * n)(), (C.c_size_t * n)()
    for i, im in enumerate(images):
        a = im["data"] if isinstance(im, dict) else im
        if a is None:
            raise L.StitchError(-6, "图片%d解码异常" % i)
        a = np.asarray(a)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 4:
            raise TypeError("image %d: expected an HxWx4 uint8 RGBA array" % i)
        if a.strides[2] != 1 or a.strides[1] != 4:
            a = np.ascontiguousarray(a)
        keep.append(a)
        ptrs[i] = a.ctypes.data
        pitches[i] = a.strides[0]
    # ist_stitch_rgba8 (what the N-API addon binds): plan, render, and the export as ONE DMA into a pinned block of the
    # library's pool; the numpy array below is a view of that block (no host copy) and returns it to the pool when it
    # is garbage collected
    cplan = L.Plan()
    lim = _limits(o)
    out = C.POINTER(C.c_uint8)()
    if o.get("devices"):
        devs = (C.c_int * len(o["devices"]))(*[int(d) for d in o["devices"]])
        rc = L.check(L.lib.ist_stitch_rgba8_multi(devs, len(o["devices"]), descs, ptrs, pitches, n, _DIRECTIONS[direction], _MODES[o["mode"]],
                                                  float(o["gap"] or 0), C.byref(lim), _filter_of(o), _SPLITS[o["split"]], C.byref(cplan), C.byref(out)))
    else:
        rc = L.check(L.lib.ist_stitch_rgba8(_ctx(device), descs, ptrs, pitches, n, _DIRECTIONS[direction], _MODES[o["mode"]],
                                            float(o["gap"] or 0), C.byref(lim), _filter_of(o), C.byref(cplan), C.byref(out)))
    if rc == L.IST_NOTHING_TO_DO:
        return None
    w, h = int(cplan.canvas_w), int(cplan.canvas_h)
    L.lib.ist_plan_free(C.byref(cplan))
    return {"width": w, "height": h, "data": _take_pixels(out, w, h)}


def _take_pixels(out, w, h):
    """HxWx4 uint8 view of a library-owned result; ist_free runs when the last view dies."""
    import weakref
    raw = (C.c_uint8 * (w * h * 4)).from_address(C.addressof(out.contents))
    weakref.finalize(raw, L.lib.ist_free, C.cast(out, C.c_void_p))
    return np.frombuffer(raw, np.uint8).reshape(h, w, 4)


def render_ops(canvas_w, canvas_h, ops, n_ops, descs, srcs, filter="bilinear", clear=(0, 0, 0, 0), region=None, device=0):
    """A recorded Canvas op list -> HxWx4 uint8 (ist_render_rgba8: what the Canvas-2D shim's export / getImageData binds).
    srcs: list of HxWx4 uint8 arrays (None for images no op draws)."""
    n = len(srcs)
    keep, ptrs, pitches = [], (C.c_void_p * max(1, n))(), (C.c_size_t * max(1, n))()
    for i, a in enumerate(srcs):
        if a is None:
            continue
        a = np.ascontiguousarray(a)
        keep.append(a)
        ptrs[i] = a.ctypes.data
        pitches[i] = a.strides[0]
    reg, rw, rh = None, int(canvas_w), int(canvas_h)
    if region is not None:
        x, y, w, h = [int(v) for v in region]
        reg = C.byref(L.Region(x, y, w, h))
        rw, rh = min(canvas_w, x + w) - max(0, x), min(canvas_h, y + h) - max(0, y)
    data = np.empty((rh, rw, 4), np.uint8)
    clr = (C.c_uint8 * 4)(*clear)
    f = (_FILTERS[filter] if isinstance(filter, str) else int(filter))
    L.check(L.lib.ist_render_rgba8(_ctx(device), int(canvas_w), int(canvas_h), clr, ops, int(n_ops), descs, ptrs, pitches, n,
                                   f, reg, data.ctypes.data, data.strides[0]))
    return data


def stitch_via_c_abi(images, direction, opts=None, device=0):
    """stitch() with the result copied into ordinary Python-owned memory (the pinned block goes straight back to the pool)."""
    r = stitch(images, direction, opts, device)
    if r is not None:
        r["data"] = r["data"].copy()
    return r


def _take_png(out, n, copy=True):
    """The library's malloc'ed PNG as Python bytes (one copy), or with copy=False as a memoryview over the C buffer
    itself, released through ist_free when the view is garbage collected (a 146 MB file costs ~20 ms to copy)."""
    if copy:
        try:
            return C.string_at(out, n.value)
        finally:
            L.lib.ist_free(out)
    import weakref
    arr = (C.c_uint8 * n.value).from_address(C.addressof(out.contents))
    weakref.finalize(arr, L.lib.ist_free, C.cast(out, C.c_void_p))
    return memoryview(arr).cast("B")


def decode_png(data):
    """PNG file bytes -> HxWx4 uint8 RGBA (straight alpha).  Host decode (zlib + the PNG predictors); no GPU needed."""
    buf = bytes(data)
    w, h = C.c_int32(0), C.c_int32(0)
    L.check(L.lib.ist_png_info(buf, len(buf), C.byref(w), C.byref(h)))
    out = np.empty((h.value, w.value, 4), np.uint8)
    L.check(L.lib.ist_png_decode_rgba8(buf, len(buf), out.ctypes.data, out.strides[0], out.shape[0]))
    return out


def image_info(data):
    """(width, height, orientation) of a PNG or JPEG file; orientation = EXIF tag 0x0112 (0 when absent)."""
    buf = bytes(data)
    w, h, o = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    L.check(L.lib.ist_image_info(buf, len(buf), C.byref(w), C.byref(h), C.byref(o)))
    return w.value, h.value, o.value


def decode_image(data, device=0):
    """PNG or JPEG file bytes -> HxWx4 uint8 RGBA.  JPEG: Huffman decoding on the host, IDCT / upsampling / colour
    conversion on the GPU.  The bitmap is returned as stored (EXIF orientation is applied by the stitch, like the
    reference's drawWithOrientation)."""
    buf = bytes(data)
    w, h, _ = image_info(buf)
    out = np.empty((h, w, 4), np.uint8)
    ctx = _ctx(device) if buf[:2] == b"\xff\xd8" else None          # only JPEG needs the GPU
    L.check(L.lib.ist_image_decode_rgba8(ctx, buf, len(buf), out.ctypes.data, out.strides[0], out.shape[0]))
    return out


def decode_files_device(blobs, device=0, out=None):
    """File bytes -> bitmaps in HBM (ist_decode_files_device): returns ([HxWx4 uint8 CUDA tensors], [image dicts for
    plan/compile]).  Baseline JPEG: Huffman decoding + reconstruction on the GPU; only the file bytes cross PCIe.
    out: optional list of preallocated tensors (one spare row behind each is the caller's business)."""
    import torch
    n = len(blobs)
    sizes = [image_info(b) for b in blobs]
    dev = torch.device("cuda", device)
    if out is None:
        out = [torch.empty((h + 1, w, 4), dtype=torch.uint8, device=dev)[:h] for (w, h, _) in sizes]
    else:
        # the library writes `out` from its own streams: what the caller queued on these tensors (a launch still reading the
        # previous bitmaps) must be done first (include/imagestitch.h, ist_decode_files_device "Ordering")
        for d in {t.device for t in out}:
            torch.cuda.current_stream(d).synchronize()
    files = (C.c_char_p * n)(*blobs)
    lens = (C.c_int64 * n)(*[len(b) for b in blobs])
    dst, pitch, rows = (C.c_void_p * n)(), (C.c_size_t * n)(), (C.c_int64 * n)()
    for i, t in enumerate(out):
        dst[i], pitch[i], rows[i] = t.data_ptr(), t.stride(0), t.shape[0]
    descs = (L.ImageDesc * n)()
    L.check(L.lib.ist_decode_files_device(_ctx(device), files, lens, n, dst, pitch, rows, descs))
    imgs = [{"width": d.width, "height": d.height, "orientation": d.orientation, "opaque": bool(d.opaque), "fileSize": d.file_size} for d in descs]
    return out, imgs


PHASES = ("host_decode", "plan_arena", "entropy_gpu", "reconstruct", "stitch", "png", "d2h")


def last_phase_times(device=0):
    """{phase: ms} of the last file-pipeline call on the device's context (enable with set_phase_timing)."""
    ms = (C.c_double * 8)()
    L.check(L.lib.ist_ctx_last_timing(_ctx(device), ms, 8))
    return {k: ms[i] for i, k in enumerate(PHASES)}


def set_phase_timing(on, device=0):
    L.check(L.lib.ist_ctx_set_timing(_ctx(device), 1 if on else 0))


def stitch_files(paths, direction, opts=None, out_path=None, device=0, copy=True):
    """File to file, device-resident (ist_stitch_files_png): decode (Huffman / inflate on host threads, JPEG
    reconstruction on the GPU) -> plan (EXIF orientation from the file, like getImageInfo, index.js:734) -> one fused
    stitch launch -> PNG export on the GPU.  Only file bytes go in and PNG bytes come out over PCIe.
    Returns {'width','height','png'} and writes out_path when given.  The mini-program's whole onStitch: index.js:1441-1581."""
    o = _merge(opts)
    n = len(paths)
    if n == 0:
        return None
    # the library reads the files itself (ist_stitch_paths_png: one parked worker per file, into blocks its context keeps).
    # (Reading nine 12 MP JPEGs into Python bytes cost ~1 ms of the call; opening and mapping them from Python still 0.2 ms.)
    cpaths = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    cplan = L.Plan()
    lim = _limits(o)
    out, ln = C.POINTER(C.c_uint8)(), C.c_int64(0)
    rc = L.check(L.lib.ist_stitch_paths_png(_ctx_png(device, o["pngLevel"]), cpaths, n, _DIRECTIONS[direction], _MODES[o["mode"]], float(o["gap"] or 0),
                                            C.byref(lim), _filter_of(o), C.byref(cplan), C.byref(out), C.byref(ln)))
    if rc == L.IST_NOTHING_TO_DO:
        return None
    w, h = int(cplan.canvas_w), int(cplan.canvas_h)
    L.lib.ist_plan_free(C.byref(cplan))
    res = {"width": w, "height": h, "png": _take_png(out, ln, copy)}
    if out_path:
        with open(out_path, "wb") as f:
            f.write(res["png"])
    return res


def encode_png(pixels, device=0, level=None):
    """Lossless PNG (colour type 6) of an HxWx4 uint8 array, encoded on the GPU (export step, utils/canvas.js:205-242).
    level 0: stored deflate blocks; 1: Paeth + run-length + Huffman (ist_ctx_set_png_level)."""
    a = np.asarray(pixels)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 4:
        raise TypeError("expected an HxWx4 uint8 RGBA array")
    if a.strides[2] != 1 or a.strides[1] != 4 or a.strides[0] < 4 * a.shape[1]:
        a = np.ascontiguousarray(a).copy()
    out, n = C.POINTER(C.c_uint8)(), C.c_int64(0)
    L.check(L.lib.ist_png_encode_rgba8(_ctx_png(device, level), a.ctypes.data, a.strides[0], a.shape[1], a.shape[0], C.byref(out), C.byref(n)))
    return _take_png(out, n)


def stitch_png(images, direction, opts=None, device=0):
    """stitch(images, direction, opts) with the reference's export: returns {'width','height','png': bytes}.  The
    canvas stays on the device; only the PNG crosses PCIe."""
    o = _merge(opts)
    n = len(images)
    if n == 0:
        return None
    descs = _descs(images)
    keep, ptrs, pitches = [], (C.c_void_p * n)(), (C.c_size_t * n)()
    for i, im in enumerate(images):
        a = im["data"] if isinstance(im, dict) else im
        if a is None:
            raise L.StitchError(-6, "图片%d解码异常" % i)
        a = np.ascontiguousarray(a)
        keep.append(a)
        ptrs[i] = a.ctypes.data
        pitches[i] = a.strides[0]
    cplan = L.Plan()
    lim = _limits(o)
    out, ln = C.POINTER(C.c_uint8)(), C.c_int64(0)
    rc = L.check(L.lib.ist_stitch_png(_ctx_png(device, o["pngLevel"]), descs, ptrs, pitches, n, _DIRECTIONS[direction], _MODES[o["mode"]],
                                      float(o["gap"] or 0), C.byref(lim), _filter_of(o), C.byref(cplan), C.byref(out), C.byref(ln)))
    if rc == L.IST_NOTHING_TO_DO:
        return None
    w, h = int(cplan.canvas_w), int(cplan.canvas_h)
    L.lib.ist_plan_free(C.byref(cplan))
    return {"width": w, "height": h, "png": _take_png(out, ln)}


def encode_png_device(canvas, out=None, stream=None, device=None, level=None):
    """PNG of a canvas that is resident in HBM (HxWx4 uint8 CUDA tensor) into a CUDA uint8 tensor; returns (tensor, length)."""
    import torch
    h, w = int(canvas.shape[0]), int(canvas.shape[1])
    cap = int(L.lib.ist_png_bound(w, h))
    if out is None:
        out = torch.empty(cap + 16, dtype=torch.uint8, device=canvas.device)
    base = out.data_ptr()
    aligned = (base + 15) & ~15
    st = stream if stream is not None else torch.cuda.current_stream(canvas.device)
    n = C.c_int64(0)
    dev = canvas.device.index if device is None else device
    L.check(L.lib.ist_png_encode_device(_ctx_png(dev or 0, level), C.c_void_p(canvas.data_ptr()), canvas.stride(0), w, h,
                                        C.c_void_p(aligned), out.numel() - (aligned - base), C.byref(n), C.c_void_p(st.cuda_stream)))
    off = aligned - base
    return out[off:off + n.value], n.value


class StitchJob:
    """A compiled op list on one device: re-launchable on new source / destination buffers with no upload."""

    def __init__(self, ctx, handle, n_images, device=0):
        self._ctx, self._h, self.n_images, self._device = ctx, handle, n_images, int(device)
        info = L.JobInfo()
        L.check(L.lib.ist_job_info_get(handle, C.byref(info)))
        self.info = {k: getattr(info, k) for k, _ in L.JobInfo._fields_}
        self.canvas_w, self.canvas_h = int(info.canvas_w), int(info.canvas_h)
        self._src = (C.c_void_p * max(1, n_images))()
        self._pitch = (C.c_size_t * max(1, n_images))()

    @property
    def preferred_pitch(self):
        """bytes per canvas row this job runs fastest on (ist_job_preferred_dst_pitch): dense rows for a strip the library walks in its
        flat form, otherwise rows padded to a multiple of 4 KiB"""
        return int(L.lib.ist_job_preferred_dst_pitch(self._h))

    def empty_canvas(self, device=None):
        """an uninitialised canvas tensor (canvas_h x canvas_w x 4, uint8) on the job's device whose row pitch is preferred_pitch"""
        import torch
        dev = torch.device("cuda", self._device if device is None else device)
        pitch = self.preferred_pitch
        raw = torch.empty((self.canvas_h * pitch + 4096,), dtype=torch.uint8, device=dev)
        off = (-raw.data_ptr()) % 4096
        return raw[off:off + self.canvas_h * pitch].view(self.canvas_h, pitch // 4, 4)[:, :self.canvas_w]

    def launch_ptrs(self, src_ptrs, src_pitches, dst_ptr, dst_pitch, stream=0):
        for i, (p, q) in enumerate(zip(src_ptrs, src_pitches)):
            self._src[i] = p
            self._pitch[i] = q
        L.check(L.lib.ist_job_launch(self._h, self._src, self._pitch, self.n_images, C.c_void_p(dst_ptr), dst_pitch,
                                     C.c_void_p(stream)))

    def launch(self, srcs, out, stream=None):
        """srcs: list of HxWx4 uint8 CUDA tensors (None for images the job does not sample); out: canvas tensor."""
        import torch
        st = stream if stream is not None else torch.cuda.current_stream(out.device)
        ptrs = [0 if t is None else t.data_ptr() for t in srcs]
        pitches = [0 if t is None else t.stride(0) for t in srcs]
        self.launch_ptrs(ptrs, pitches, out.data_ptr(), out.stride(0), st.cuda_stream)

    def close(self):
        if self._h:
            L.lib.ist_job_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stitcher:
    """Device-resident stitcher: one per GPU (one process per GPU in the multi-GPU layout)."""

    def __init__(self, device=0):
        self.device = int(device)
        self._ctx = _ctx(self.device)

    def compile_ops(self, canvas_w, canvas_h, ops, n_ops, descs, n_images, filter="bilinear", clear=(0, 0, 0, 0), clip=None):
        clr = (C.c_uint8 * 4)(*clear)
        region = None
        if clip is not None:
            region = C.byref(L.Region(*[int(v) for v in clip]))
        h = L.lib.ist_job_create(self._ctx, int(canvas_w), int(canvas_h), clr, ops, int(n_ops), descs, int(n_images),
                                 _FILTERS[filter] if isinstance(filter, str) else int(filter), region)
        if not h:
            raise L.StitchError(-1, L.last_error())
        return StitchJob(self._ctx, h, n_images, self.device)

    def compile(self, images, direction, opts=None, only_images=None):
        """Plan + compile.  only_images: iterable of image indices this device renders (multi-GPU sharding);
        the other rects are dropped from the op list (their canvas area is left to whoever owns them)."""
        o = _merge(opts)
        p = plan(images, direction, o)
        if p is None:
            return None, None
        ops, n_ops = p.ops()
        if only_images is not None:
            keep = set(int(i) for i in only_images)
            sel = [ops[0]] + [ops[k] for k in range(1, n_ops) if ops[k].image in keep]
            ops = (L.Op * len(sel))(*sel)
            n_ops = len(sel)
        job = self.compile_ops(p.canvas_w, p.canvas_h, ops, n_ops, p._descs, len(images), _filter_of(o))
        return p, job


class GroupJob:
    """A stitch compiled for a device group (ist_group_job_*): parts[k] = {image, slot, device, box, rows} and one source
    pointer per part at launch."""

    def __init__(self, group, handle, plan):
        self._g, self._h, self.plan = group, handle, plan
        n = C.c_int(0)
        L.check(L.lib.ist_group_job_parts(handle, None, 0, C.byref(n)))
        arr = (L.Part * max(1, n.value))()
        L.check(L.lib.ist_group_job_parts(handle, arr, n.value, C.byref(n)))
        self.parts = [{"image": p.image, "slot": p.slot, "device": group.devices[p.slot], "box": (p.X0, p.Y0, p.X1, p.Y1),
                       "rows": (p.sy0, p.sy1), "in_place": bool(p.in_place)} for p in arr[:n.value]]

    def launch(self, part_srcs, out):
        """part_srcs[k]: HxWx4 uint8 CUDA tensor on parts[k]['device'] holding the WHOLE image, or (tensor, first_row) for a
        partial holding (rows first_row ... ; one spare row behind the last must be readable).  out: canvas on the root.
        The group's streams do not synchronise with the caller's: work the caller queued on these buffers (uploads, fills) is
        waited for here, on each tensor's current torch stream."""
        import torch
        for dev in {(s[0] if isinstance(s, tuple) else s).device for s in part_srcs} | {out.device}:
            torch.cuda.current_stream(dev).synchronize()
        n = len(self.parts)
        ptrs, pitches = (C.c_void_p * n)(), (C.c_size_t * n)()
        for k, s in enumerate(part_srcs):
            t, first = (s if isinstance(s, tuple) else (s, 0))
            ptrs[k] = t.data_ptr() - first * t.stride(0)
            pitches[k] = t.stride(0)
        L.check(L.lib.ist_group_job_launch(self._h, ptrs, pitches, n, C.c_void_p(out.data_ptr()), out.stride(0)))

    def close(self):
        if self._h:
            L.lib.ist_group_job_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StitchGroup:
    """Several GPUs driven from this one process (the N-API host's layout; ist_group_*).  devices[0] is the root."""

    def __init__(self, devices):
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        self._h = L.lib.ist_group_create(arr, len(self.devices))
        if not self._h:
            raise L.StitchError(-5, L.last_error())

    def compile(self, images, direction, opts=None):
        o = _merge(opts)
        p = plan(images, direction, o)
        if p is None:
            return None
        ops, n_ops = p.ops()
        clr = (C.c_uint8 * 4)(0, 0, 0, 0)
        h = L.lib.ist_group_job_create(self._h, p.canvas_w, p.canvas_h, clr, ops, n_ops, p._descs, len(images), _filter_of(o), _SPLITS[o["split"]])
        if not h:
            raise L.StitchError(-1, L.last_error())
        return GroupJob(self, h, p)

    def sync(self):
        L.check(L.lib.ist_group_sync(self._h))

    def close(self):
        if self._h:
            L.lib.ist_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
