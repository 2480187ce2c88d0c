"""imagestitching_amd — MI355X-native strip stitcher (the Canvas-2D concatenation path of Iamctb/ImageStitching).

Everything that touches pixels is hand-written HIP behind the C-ABI in include/imagestitch.h
(imagestitching_amd/libimagestitch.so).  Importing this package fails if that library has not been built.
"""
from ._lib import (FILTER_BILINEAR, FILTER_NEAREST, HORIZONTAL, VERTICAL, StitchError, last_error)  # noqa: F401
from .stitch import (DEFAULT_OPTS, GroupJob, StitchGroup, Stitcher, StitchJob, StitchPlan, decode_files_device, decode_image, decode_png, encode_png, encode_png_device, image_info, last_phase_times, plan, set_phase_timing,  # noqa: F401
                     stitch, stitch_files, stitch_png)

__all__ = ["stitch", "stitch_png", "encode_png", "encode_png_device", "decode_png", "decode_image", "decode_files_device", "image_info", "set_phase_timing", "last_phase_times", "stitch_files", "plan", "Stitcher", "StitchGroup", "GroupJob", "StitchJob", "StitchPlan", "StitchError",
           "DEFAULT_OPTS"]
