"""Checks on the gfx950 machine code that ships in libimagestitch.so (CPU only: llvm-objdump on the code objects of the
.hip_fatbin section).  Reference anchor of what these kernels compute: utils/canvas.js:153-202 (drawImage) and the platform
decode / export either side of it (utils/canvas.js:27-121, 205-242)."""
import os
import shutil
import struct
import subprocess

import pytest

from tests import util as U

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
SO = os.path.join(U.ROOT, "imagestitching_amd", "libimagestitch.so")


def _code_objects(tmp):
    fat = os.path.join(tmp, "fatbin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", SO, fat])
    d = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        k = d.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
        if k < 0:
            return out
        cnt = struct.unpack_from("<Q", d, k + 24)[0]
        p = k + 32
        for _ in range(cnt):
            off, size, tl = struct.unpack_from("<QQQ", d, p)
            p += 24
            triple = d[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                path = os.path.join(tmp, "co%d.o" % len(out))
                with open(path, "wb") as f:
                    f.write(d[k + off:k + off + size])
                out.append(path)
        pos = k + 24


@pytest.mark.skipif(not os.path.exists(OBJDUMP) or shutil.which("objcopy") is None, reason="needs llvm-objdump and objcopy")
def test_the_shipped_kernels_are_gfx950_and_free_of_the_packed_shift_hazard(tmp_path):
    """Round 4, found on the GPU: written as clamp(x >> 16, 0, 255) per channel + shifts + ors, the colour conversion compiled
    (ROCm 7.2) to v_ashr_pk_u8_i32, whose destination KEPT its upper 16 bits where the compiler assumed zeros - blue came out as
    (right value | whatever the register held).  The source now clamps before the shift; no kernel of the library may contain
    that instruction again without someone looking.  Also: every code object is for gfx950 and every kernel family is in."""
    objs = _code_objects(str(tmp_path))
    assert len(objs) >= 5                                      # one per .hip translation unit
    names = ""
    for o in objs:
        dis = subprocess.run([OBJDUMP, "-d", o], capture_output=True, text=True, check=True).stdout
        assert "v_ashr_pk_u8_i32" not in dis and "v_ashr_pk_i8_i32" not in dis, o
        names += subprocess.run([OBJDUMP, "-t", o], capture_output=True, text=True, check=True).stdout
    for kernel in ("ist_stitch_kernel", "ist_jpeg_fused_kernel", "ist_jpeg_idct_kernel", "ist_jpeg_sync_kernel", "ist_jpeg_write_kernel", "ist_png"):
        assert kernel in names, kernel
