"""Probe (not a pytest module): re-creates case N of test_random_op_lists_against_oracle and reports where the HIP
render differs most from the oracle.  usage: python tests/probe_random_ops.py N"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagestitching_amd import _lib as L          # noqa: E402
from imagestitching_amd.stitch import _ctx         # noqa: E402
from oracle import oracle as O                     # noqa: E402
from tests import util as U                        # noqa: E402

want = int(sys.argv[1])
rng = np.random.default_rng(777)
for case in range(want + 1):
    cw, ch = int(rng.integers(8, 300)), int(rng.integers(8, 300))
    n_img = int(rng.integers(1, 4))
    px = [U.rand_image(900 + 7 * case + k, int(rng.integers(2, 120)), int(rng.integers(2, 120)), opaque=bool(rng.integers(0, 2))) for k in range(n_img)]
    descs_o = [{"width": a.shape[1], "height": a.shape[0]} for a in px]
    ops_o = []
    if rng.integers(0, 2):
        ops_o.append({"kind": "fill", "m": [1, 0, 0, 1, 0, 0], "rect": [0, 0, cw, ch], "rgba": tuple(int(v) for v in rng.integers(0, 256, 3)) + (255,)})
    for _ in range(int(rng.integers(1, 6))):
        k = int(rng.integers(0, n_img))
        h, w = px[k].shape[:2]
        sc = 1.0 if rng.integers(0, 3) == 0 else float(rng.uniform(0.3, 3.0))
        t = int(rng.integers(0, 8))
        sx, sy = (-sc if t & 1 else sc), (-sc if t & 2 else sc)
        e, f = float(rng.integers(0, cw)), float(rng.integers(0, ch))
        if rng.integers(0, 2):
            e += float(rng.uniform(0, 1)); f += float(rng.uniform(0, 1))
        m = [0, sx, sy, 0, e, f] if t & 4 else [sx, 0, 0, sy, e, f]
        if rng.integers(0, 2):
            s = [0, 0, w, h]
        else:
            s = [float(rng.uniform(-5, w / 2)), float(rng.uniform(-5, h / 2)), float(rng.uniform(1, w)), float(rng.uniform(1, h))]
        d = [float(rng.uniform(-20, 20)), float(rng.uniform(-20, 20)), float(rng.uniform(4, 150)), float(rng.uniform(4, 150))]
        if rng.integers(0, 3) == 0:
            d = [round(v) for v in d]
        ops_o.append({"kind": "draw", "image": k, "m": m, "s": s, "d": d})
    clear = (0, 0, 0, 0) if rng.integers(0, 2) else tuple(int(v) for v in rng.integers(0, 256, 3)) + (255,)
    filt = "nearest" if rng.integers(0, 2) else "bilinear"
    aa = bool(rng.integers(0, 3) == 0)


def run(ops_sel):
    ref = O.render_ops(cw, ch, ops_sel, descs_o, px, filt, clear=clear, edge_aa=aa)
    ops = (L.Op * len(ops_sel))()
    for i, o in enumerate(ops_sel):
        ops[i].m[:] = o["m"]
        if o["kind"] == "fill":
            ops[i].kind = 0; ops[i].image = -1; ops[i].d[:] = o["rect"]; ops[i].rgba[:] = o["rgba"]
        else:
            ops[i].kind = 1; ops[i].image = o["image"]; ops[i].s[:] = o["s"]; ops[i].d[:] = o["d"]
    descs = (L.ImageDesc * n_img)(*[L.ImageDesc(a.shape[1], a.shape[0], 1, 0, 0, 0, 0) for a in px])
    ptrs = (C.c_void_p * n_img)(*[a.ctypes.data for a in px])
    pit = (C.c_size_t * n_img)(*[a.strides[0] for a in px])
    out = np.zeros((ch, cw, 4), np.uint8)
    fl = {"nearest": 0, "bilinear": 1}[filt] | (0x100 if aa else 0)
    L.check(L.lib.ist_render_rgba8(_ctx(0), cw, ch, (C.c_uint8 * 4)(*clear), ops, len(ops_sel), descs, ptrs, pit, n_img, fl, None, out.ctypes.data, out.strides[0]))
    return out, ref


print("case", want, "canvas", cw, ch, "filter", filt, "aa", aa, "clear", clear, "images", [(a.shape, bool((a[..., 3] == 255).all())) for a in px])
for o in ops_o:
    print("  ", o)
out, ref = run(ops_o)
diff = np.abs(out.astype(np.int16) - ref.astype(np.int16))
print("max diff", diff.max(), "count>1", int((diff > 1).sum()))
ys, xs, cs = np.nonzero(diff > 1)
for y, x, c in list(zip(ys, xs, cs))[:10]:
    print("  at (x=%d,y=%d,c=%d): hip %s oracle %s" % (x, y, c, out[y, x].tolist(), ref[y, x].tolist()))
for k in range(1, len(ops_o) + 1):
    o2, r2 = run(ops_o[:k])
    print("first %d ops: max diff %d" % (k, np.abs(o2.astype(np.int16) - r2.astype(np.int16)).max()))
