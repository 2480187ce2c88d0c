"""ctypes binding of libimagestitch.so (the C-ABI in include/imagestitch.h).

There is no Python or CPU fallback: if the shared library (HIP kernels + C-ABI) is missing, import fails loudly.
Build it with `make -C imagestitching_amd/csrc` or `python -c "import __graft_entry__ as g; g.build()"`.
"""
import ctypes as C
import os

# torch ships its own libamdhip64.so (SONAME libamdhip64.so.7).  It must be in the process BEFORE libimagestitch.so is
# loaded so that both resolve to ONE HIP runtime; loading the system runtime first and torch's second gives two HSA
# runtimes in one process and the second one finds no GPU.
import torch  # noqa: F401  (plumbing: device memory, streams, torch.distributed)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libimagestitch.so")
if os.environ.get("IST_TUNING") == "1" and os.environ.get("IST_LIB_PATH"):      # experiments: a variant build of the library (tools/sweep_*.py)
    LIB_PATH = os.environ["IST_LIB_PATH"]

VERTICAL, HORIZONTAL = 0, 1
MODE_MIN, MODE_MAX, MODE_ORIGINAL = 0, 1, 2
PLATFORM_OTHER, PLATFORM_IOS, PLATFORM_ANDROID = 0, 1, 2
FILTER_NEAREST, FILTER_BILINEAR, FILTER_AREA = 0, 1, 2
SPLIT_IMAGE, SPLIT_BAND, SPLIT_ROWS, SPLIT_AUTO = 0, 1, 2, 3

IST_OK, IST_NOTHING_TO_DO = 0, 1
ERROR_NAMES = {-1: "IST_E_INVALID", -2: "IST_E_SIZE_UNAVAILABLE", -3: "IST_E_OUTPUT_SIZE", -4: "IST_E_NO_CONTEXT",
               -5: "IST_E_NO_DEVICE", -6: "IST_E_DECODE", -7: "IST_E_UNSUPPORTED", -8: "IST_E_NOMEM", -9: "IST_E_HIP"}


class ImageDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("orientation", C.c_int32),
                ("bmp_width", C.c_int32), ("bmp_height", C.c_int32), ("opaque", C.c_int32),
                ("file_size", C.c_int64)]


class Limits(C.Structure):
    _fields_ = [("platform", C.c_int32), ("reserved", C.c_int32), ("max_side", C.c_double),
                ("max_pixels", C.c_double), ("max_super_sample", C.c_double)]


class Rect(C.Structure):
    _fields_ = [("image", C.c_int32), ("orientation", C.c_int32),
                ("dx", C.c_double), ("dy", C.c_double), ("dw", C.c_double), ("dh", C.c_double)]


class Plan(C.Structure):
    _fields_ = [("out_w", C.c_double), ("out_h", C.c_double), ("scale_down", C.c_double),
                ("super_sample", C.c_double), ("canvas_w", C.c_int64), ("canvas_h", C.c_int64),
                ("big_task", C.c_int32), ("n_rects", C.c_int32), ("rects", C.POINTER(Rect))]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("image", C.c_int32), ("m", C.c_double * 6), ("s", C.c_double * 4),
                ("d", C.c_double * 4), ("rgba", C.c_uint8 * 4), ("reserved", C.c_int32)]


class Region(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


class Part(C.Structure):
    _fields_ = [("image", C.c_int32), ("op", C.c_int32), ("slot", C.c_int32),
                ("X0", C.c_int32), ("Y0", C.c_int32), ("X1", C.c_int32), ("Y1", C.c_int32),
                ("sx0", C.c_int32), ("sy0", C.c_int32), ("sx1", C.c_int32), ("sy1", C.c_int32), ("in_place", C.c_int32)]


class FlatCell(C.Structure):
    _fields_ = [("path", C.c_int32), ("image", C.c_int32), ("X0", C.c_int32), ("Y0", C.c_int32), ("X1", C.c_int32), ("Y1", C.c_int32),
                ("src_offset", C.c_int64), ("bg", C.c_uint32), ("opaque", C.c_int32)]


class JobInfo(C.Structure):
    _fields_ = [("canvas_w", C.c_int64), ("canvas_h", C.c_int64), ("n_ops", C.c_int32), ("n_cells", C.c_int32),
                ("n_tiles", C.c_int64), ("out_pixels", C.c_int64), ("src_pixels_touched", C.c_int64),
                ("algorithmic_bytes", C.c_int64), ("tiles_fill", C.c_int64), ("tiles_copy", C.c_int64),
                ("tiles_sample", C.c_int64), ("tiles_general", C.c_int64)]


# every symbol include/imagestitch.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("ist_abi_version", C.c_int, []),
    ("ist_last_error", C.c_char_p, []),
    ("ist_device_count", C.c_int, []),
    ("ist_debug_device_allocs", C.c_int64, []),
    ("ist_debug_gpu_entropy_files", C.c_int64, []),
    ("ist_debug_direct_images", C.c_int64, []),
    ("ist_debug_host_sink_stitches", C.c_int64, []),
    ("ist_debug_flat_launches", C.c_int64, []),
    ("ist_debug_duplex_stitches", C.c_int64, []),
    ("ist_limits_default", None, [C.c_int, C.POINTER(Limits)]),
    ("ist_limits_unlimited", None, [C.POINTER(Limits)]),
    ("ist_plan_compute", C.c_int, [C.POINTER(ImageDesc), C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(Limits), C.POINTER(Plan)]),
    ("ist_plan_free", None, [C.POINTER(Plan)]),
    ("ist_plan_ops", C.c_int, [C.POINTER(Plan), C.POINTER(ImageDesc), C.c_int, C.POINTER(Op), C.POINTER(C.c_int)]),
    ("ist_op_box", C.c_int, [C.POINTER(Op), C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int32)]),
    ("ist_debug_flat_form", C.c_int, [C.c_int64, C.c_int64, C.POINTER(C.c_uint8), C.POINTER(Op), C.c_int, C.POINTER(ImageDesc), C.c_int, C.c_int,
                                      C.POINTER(Region), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(FlatCell), C.c_int, C.POINTER(C.c_int)]),
    ("ist_shard_parts", C.c_int, [C.POINTER(Op), C.c_int, C.c_int64, C.c_int64, C.POINTER(ImageDesc), C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(Part), C.c_int, C.POINTER(C.c_int)]),
    ("ist_shard_row_cuts", C.c_int, [C.c_int64, C.c_int, C.POINTER(C.c_int32)]),
    ("ist_shard_resolve", C.c_int, [C.POINTER(Op), C.c_int, C.c_int64, C.c_int64, C.POINTER(ImageDesc), C.c_int, C.c_int, C.c_int]),
    ("ist_ctx_create", C.c_void_p, [C.c_int]),
    ("ist_ctx_destroy", None, [C.c_void_p]),
    ("ist_ctx_sync", C.c_int, [C.c_void_p]),
    ("ist_ctx_set_png_level", C.c_int, [C.c_void_p, C.c_int]),
    ("ist_job_create", C.c_void_p, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_uint8), C.POINTER(Op), C.c_int,
                                    C.POINTER(ImageDesc), C.c_int, C.c_int, C.POINTER(Region)]),
    ("ist_job_info_get", C.c_int, [C.c_void_p, C.POINTER(JobInfo)]),
    ("ist_job_preferred_dst_pitch", C.c_size_t, [C.c_void_p]),
    ("ist_job_launch", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("ist_job_destroy", None, [C.c_void_p]),
    ("ist_group_create", C.c_void_p, [C.POINTER(C.c_int), C.c_int]),
    ("ist_group_destroy", None, [C.c_void_p]),
    ("ist_group_slots", C.c_int, [C.c_void_p]),
    ("ist_group_device", C.c_int, [C.c_void_p, C.c_int]),
    ("ist_group_job_create", C.c_void_p, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_uint8), C.POINTER(Op), C.c_int,
                                          C.POINTER(ImageDesc), C.c_int, C.c_int, C.c_int]),
    ("ist_group_job_destroy", None, [C.c_void_p]),
    ("ist_group_job_parts", C.c_int, [C.c_void_p, C.POINTER(Part), C.c_int, C.POINTER(C.c_int)]),
    ("ist_group_job_launch", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_void_p, C.c_size_t]),
    ("ist_group_sync", C.c_int, [C.c_void_p]),
    ("ist_group_stitch_rgba8", C.c_int, [C.c_void_p, C.POINTER(ImageDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int,
                                         C.c_int, C.c_int, C.c_double, C.POINTER(Limits), C.c_int, C.c_int, C.POINTER(Plan),
                                         C.POINTER(C.POINTER(C.c_uint8))]),
    ("ist_stitch_rgba8_multi", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(ImageDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int,
                                         C.c_int, C.c_int, C.c_double, C.POINTER(Limits), C.c_int, C.c_int, C.POINTER(Plan),
                                         C.POINTER(C.POINTER(C.c_uint8))]),
    ("ist_stitch_rgba8", C.c_int, [C.c_void_p, C.POINTER(ImageDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int,
                                   C.c_int, C.c_int, C.c_double, C.POINTER(Limits), C.c_int, C.POINTER(Plan),
                                   C.POINTER(C.POINTER(C.c_uint8))]),
    ("ist_render_rgba8", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_uint8), C.POINTER(Op), C.c_int,
                                   C.POINTER(ImageDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int,
                                   C.POINTER(Region), C.c_void_p, C.c_size_t]),
    ("ist_free", None, [C.c_void_p]),
    ("ist_pool_trim", None, []),
    ("ist_png_info", C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("ist_png_decode_rgba8", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_int64]),
    ("ist_jpeg_info", C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("ist_jpeg_decode_rgba8", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_int64]),
    ("ist_image_info", C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("ist_image_decode_rgba8", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_int64]),
    ("ist_decode_files_device", C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                          C.POINTER(C.c_int64), C.POINTER(ImageDesc)]),
    ("ist_ctx_set_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("ist_ctx_last_timing", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    ("ist_stitch_files_png", C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int, C.c_double,
                                       C.POINTER(Limits), C.c_int, C.POINTER(Plan), C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int64)]),
    ("ist_stitch_paths_png", C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_double,
                                       C.POINTER(Limits), C.c_int, C.POINTER(Plan), C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int64)]),
    ("ist_png_bound", C.c_int64, [C.c_int64, C.c_int64]),
    ("ist_png_encode_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                        C.POINTER(C.c_int64), C.c_void_p]),
    ("ist_png_encode_rgba8", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int64,
                                       C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int64)]),
    ("ist_render_png", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_uint8), C.POINTER(Op), C.c_int,
                                 C.POINTER(ImageDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int,
                                 C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int64)]),
    ("ist_stitch_png", C.c_int, [C.c_void_p, C.POINTER(ImageDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int,
                                 C.c_int, C.c_int, C.c_double, C.POINTER(Limits), C.c_int, C.POINTER(Plan),
                                 C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int64)]),
]

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "imagestitching_amd: %s is missing. The stitch path is HIP-only (no CPU fallback); build it with "
        "`make -C imagestitching_amd/csrc` (hipcc, --offload-arch=gfx950)." % LIB_PATH)

lib = C.CDLL(LIB_PATH)
for _name, _res, _args in SYMBOLS:
    _f = getattr(lib, _name)          # AttributeError here = the library does not export a declared symbol
    _f.restype = _res
    _f.argtypes = _args


class StitchError(RuntimeError):
    """Mirrors the reference's single catch: Error(msg) -> toast '拼图失败：'+msg (pages/index/index.js:1618-1624)."""

    def __init__(self, code, message):
        self.code = code
        self.reason = message
        super().__init__("拼图失败：%s [%s]" % (message, ERROR_NAMES.get(code, code)))


def last_error():
    s = lib.ist_last_error()
    return s.decode("utf-8", "replace") if s else ""


def check(rc):
    if rc < 0:
        raise StitchError(rc, last_error())
    return rc
