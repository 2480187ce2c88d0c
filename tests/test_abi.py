"""The C-ABI library loads, exports every symbol include/imagestitch.h declares, and fails loudly without a GPU.
CPU only: no compute entry point succeeds here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "imagestitch.h")


def _declared():
    src = open(HEADER, encoding="utf-8").read()
    return sorted(set(re.findall(r"IST_API\s+[\w\s\*]+?\b(ist_\w+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = _declared()
    for must in ["ist_plan_compute", "ist_plan_free", "ist_plan_ops", "ist_stitch_rgba8", "ist_render_rgba8",
                 "ist_job_create", "ist_job_launch", "ist_last_error", "ist_ctx_create"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    from imagestitching_amd import _lib as L
    lib = C.CDLL(L.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), "libimagestitch.so does not export %s" % name
    bound = {n for n, _, _ in L.SYMBOLS}
    assert bound == set(_declared()), "python binding and header disagree: %s" % (bound ^ set(_declared()))
    assert L.lib.ist_abi_version() == 2


def test_header_compiles_as_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "imagestitch.h"\nint main(void){ ist_plan p; (void)p; return sizeof(ist_op) == 128 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_struct_layouts_match_ctypes():
    from imagestitching_amd import _lib as L
    assert C.sizeof(L.ImageDesc) == 32 and C.sizeof(L.Limits) == 32 and C.sizeof(L.Rect) == 40
    assert C.sizeof(L.Op) == 128 and C.sizeof(L.Plan) == 64 and C.sizeof(L.JobInfo) == 88


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import imagestitching_amd as ist
    import numpy as np
    with pytest.raises(ist.StitchError) as e:
        ist.stitch([np.zeros((2, 2, 4), np.uint8)], "vertical")
    assert e.value.code == -5
    with pytest.raises(ist.StitchError):
        ist.Stitcher(0)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under imagestitching_amd/ or node/ may reference it."""
    bad = []
    for base in ("imagestitching_amd", "node"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".js", ".c", ".cc", ".ts")):
                    txt = open(os.path.join(dp, f), encoding="utf-8", errors="replace").read()
                    if re.search(r"(from|import)\s+oracle|oracle/|libist_oracle|orc_", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_planner_argument_errors_map_to_codes():
    import ctypes as C
    from imagestitching_amd import _lib as L
    lim = L.Limits()
    L.lib.ist_limits_unlimited(C.byref(lim))
    plan = L.Plan()
    descs = (L.ImageDesc * 1)(L.ImageDesc(4, 4, 1, 0, 0, 0, 0))
    assert L.lib.ist_plan_compute(descs, 0, 0, 0, 0.0, C.byref(lim), C.byref(plan)) == L.IST_NOTHING_TO_DO
    assert L.lib.ist_plan_compute(descs, 1, 7, 0, 0.0, C.byref(lim), C.byref(plan)) == -1          # bad direction
    assert "direction" in L.last_error()
    assert L.lib.ist_plan_compute(None, 1, 0, 0, 0.0, C.byref(lim), C.byref(plan)) == -1           # NULL images
    assert L.lib.ist_plan_compute(descs, 1, 0, 99, 0.0, C.byref(lim), C.byref(plan)) == 0          # unknown mode -> 'min' (index.js:1257)
    L.lib.ist_plan_free(C.byref(plan))
    huge = (L.ImageDesc * 2)(L.ImageDesc(2 ** 30, 8, 1, 0, 0, 0, 0), L.ImageDesc(2 ** 30, 8, 1, 0, 0, 0, 0))
    assert L.lib.ist_plan_compute(huge, 2, 1, 0, 0.0, C.byref(lim), C.byref(plan)) == 0            # the caps shrink it (index.js:1337-1357)
    assert plan.canvas_w <= 1048576 and plan.scale_down < 1.0
    L.lib.ist_plan_free(C.byref(plan))
    lim.max_side, lim.max_pixels = float(2 ** 40), float(2 ** 70)
    assert L.lib.ist_plan_compute(huge, 2, 1, 0, 0.0, C.byref(lim), C.byref(plan)) == -3           # canvas side beyond 2^29
    assert L.lib.ist_png_bound(0, 5) == 0 and L.lib.ist_png_bound(4032, 27216) > 4032 * 27216 * 4


def test_job_and_render_need_a_context():
    import ctypes as C
    from imagestitching_amd import _lib as L
    assert not L.lib.ist_job_create(None, 4, 4, None, None, 0, None, 0, 1, None)
    assert "绘图上下文" in L.last_error()                                                              # index.js:1412
    out, n = C.POINTER(C.c_uint8)(), C.c_int64(0)
    assert L.lib.ist_png_encode_rgba8(None, None, 0, 1, 1, C.byref(out), C.byref(n)) == -4


def test_group_entry_points_reject_bad_lists_without_a_device():
    """ist_group_create / ist_stitch_rgba8_multi (SURVEY 8b `devices`): argument errors come first, then - on a box
    without a GPU - IST_E_NO_DEVICE; never a crash, never a CPU fallback"""
    from imagestitching_amd import _lib as L
    assert not L.lib.ist_group_create(None, 0) and "device list" in L.last_error()
    devs = (C.c_int * 2)(0, 1)
    assert not L.lib.ist_group_create(devs, 0)
    assert not L.lib.ist_group_create(devs, 65)
    if L.lib.ist_device_count() == 0:
        assert not L.lib.ist_group_create(devs, 2) and "no HIP device" in L.last_error()
    assert L.lib.ist_group_slots(None) == 0 and L.lib.ist_group_device(None, 0) == -1
    plan, out = L.Plan(), C.POINTER(C.c_uint8)()
    descs = (L.ImageDesc * 1)(L.ImageDesc(4, 4, 1, 0, 0, 0, 0))
    rc = L.lib.ist_stitch_rgba8_multi(None, 0, descs, None, None, 1, 0, 0, 0.0, None, 1, 0, C.byref(plan), C.byref(out))
    assert rc == -1


def test_shard_parts_is_pure_cpu_and_checks_its_arguments():
    import imagestitching_amd as ist
    from imagestitching_amd import _lib as L
    p = ist.plan([{"width": 40, "height": 30}] * 3, "vertical")
    ops, n = p.ops()
    arr, cnt = (L.Part * 16)(), C.c_int(0)
    assert L.lib.ist_shard_parts(ops, n, p.canvas_w, p.canvas_h, p._descs, 3, 1, 2, 0, arr, 16, C.byref(cnt)) == 0 and cnt.value == 3
    assert [arr[k].slot for k in range(3)] == [0, 1, 0]
    assert L.lib.ist_shard_parts(ops, n, p.canvas_w, p.canvas_h, p._descs, 3, 1, 0, 0, arr, 16, C.byref(cnt)) == -1      # no slots
    assert L.lib.ist_shard_parts(ops, n, p.canvas_w, p.canvas_h, p._descs, 3, 1, 2, 7, arr, 16, C.byref(cnt)) == -1      # unknown split
    assert L.lib.ist_shard_parts(ops, n, p.canvas_w, p.canvas_h, p._descs, 3, 1, 2, 0, arr, 2, C.byref(cnt)) == -1       # table too small
    assert L.lib.ist_shard_parts(ops, n, p.canvas_w, p.canvas_h, p._descs, 3, 1, 3, 1, arr, 16, C.byref(cnt)) == 0
    rows = sorted((arr[k].Y0, arr[k].Y1, arr[k].slot) for k in range(cnt.value))
    assert rows[0][0] == 0 and rows[-1][1] == 90 and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    assert [r[2] for r in rows] == sorted(r[2] for r in rows)            # canvas order = slot order
