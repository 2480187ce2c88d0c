"""GPU side of the multi-GPU layout on ONE device: every HIP piece of imagestitching_amd/dist.py except the RCCL
transfer itself (band rendering on a non-root rank, HOLE ops + in-place rows and staged bands on the root).  The
transfer is emulated by copying the bands; the sharding/assembly logic is covered with gloo in test_dist_gloo.py."""
import numpy as np
import pytest

from tests import util as U

pytestmark = pytest.mark.gpu

SIZES = [(640, 480), (480, 640), (600, 450), (300, 170), (640, 480)]


class _LoopbackDist:
    """Stands in for torch.distributed in ONE process: isend parks the tensor, irecv copies it."""

    def __init__(self):
        self.mail = []

    class _Req:
        def wait(self):
            return None

    def P2POp(self, op, tensor, peer):
        return (op, tensor, peer)

    def isend(self, *a):
        raise NotImplementedError

    def irecv(self, *a):
        raise NotImplementedError

    def batch_isend_irecv(self, ops):
        for op, tensor, peer in ops:
            if op == self.isend:
                self.mail.append(tensor.clone())
            else:
                tensor.copy_(self.mail.pop(0))
        return [self._Req() for _ in ops]


@pytest.mark.parametrize("direction,opts", [
    ("vertical", {"filter": "bilinear", "mode": "min", "gap": 4}),
    ("horizontal", {"filter": "bilinear", "mode": "max", "gap": 0}),
    ("vertical", {"filter": "nearest", "mode": "original", "gap": 3}),
    ("vertical", {"filter": "bilinear", "mode": "min", "gap": 0}),
])
def test_two_rank_layout_on_one_gpu_matches_oracle(direction, opts):
    import torch
    from imagestitching_amd import dist as D
    pixels = [U.rand_image(300 + i, h, w) for i, (w, h) in enumerate(SIZES)]
    imgs = U.hip_images(pixels)
    world = 2
    loop = _LoopbackDist()
    # rank 1 first (its sends are parked), then the root
    sh1 = D.ShardedStitch(imgs, direction, opts, 1, world, 0)
    be1 = D.HipBackend(sh1, 0)
    srcs1 = [torch.from_numpy(a).cuda() if D.owner_of(i, world) == 1 else None for i, a in enumerate(pixels)]
    assert D.run_step(sh1, be1, srcs1, None, loop) is None
    # each band must equal the oracle's canvas cropped to the box
    ref, pd, _ = U.oracle_stitch(pixels, direction, opts)
    tol = 0 if opts["filter"] == "nearest" else 1
    for i in sh1.mine:
        X0, Y0, X1, Y1, _ = sh1.boxes[i]
        band = be1.bands[i].cpu().numpy()
        assert band.shape == (Y1 - Y0, X1 - X0, 4)
        assert U.max_abs_diff(band, ref[Y0:Y1, X0:X1]) <= tol
    sh0 = D.ShardedStitch(imgs, direction, opts, 0, world, 0)
    be0 = D.HipBackend(sh0, 0)
    srcs0 = [torch.from_numpy(a).cuda() if D.owner_of(i, world) == 0 else None for i, a in enumerate(pixels)]
    canvas = be0.new_canvas()
    canvas.fill_(0x5A)
    out = D.run_step(sh0, be0, srcs0, canvas, loop)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    assert U.max_abs_diff(got, ref) <= tol
    assert not loop.mail


def test_opaque_hint_does_not_change_opaque_results():
    import torch
    import imagestitching_amd as ist
    px = [U.rand_image(320 + i, h, w) for i, (w, h) in enumerate([(403, 302), (302, 403), (400, 300)])]
    st = ist.Stitcher(0)
    outs = []
    for flag in (False, True):
        imgs = [{"width": a.shape[1], "height": a.shape[0], "opaque": flag} for a in px]
        p, job = st.compile(imgs, "vertical", {"filter": "bilinear", "mode": "max"})
        out = torch.empty((p.canvas_h, p.canvas_w, 4), dtype=torch.uint8, device="cuda")
        job.launch([torch.from_numpy(a).cuda() for a in px], out)
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
