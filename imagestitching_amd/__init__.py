"""imagestitching_amd — MI355X-native strip stitcher (the Canvas-2D concatenation path of Iamctb/ImageStitching).

Everything that touches pixels is hand-written HIP behind the C-ABI in include/imagestitch.h
(imagestitching_amd/libimagestitch.so).  Importing this package fails if that library has not been built.
"""
from ._lib import (FILTER_BILINEAR, FILTER_NEAREST, HORIZONTAL, VERTICAL, StitchError, last_error)  # noqa: F401
from .stitch import (DEFAULT_OPTS, Stitcher, StitchJob, StitchPlan, decode_image, decode_png, encode_png, encode_png_device, image_info, plan,  # noqa: F401
                     stitch, stitch_files, stitch_png)

__all__ = ["stitch", "stitch_png", "encode_png", "encode_png_device", "decode_png", "decode_image", "image_info", "stitch_files", "plan", "Stitcher", "StitchJob", "StitchPlan", "StitchError",
           "DEFAULT_OPTS"]
