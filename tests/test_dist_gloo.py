"""world_size-2 coverage of the multi-GPU layout (imagestitching_amd/dist.py) on CPU with gloo: sharding by image,
band rendering, in-place vs staged receives, root assembly.  The render backend is the oracle here (test-only
injection); the product backend is HipBackend."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import util as U

SIZES = [(64, 48), (48, 64), (60, 45), (30, 17), (64, 48)]


class OracleBackend:
    """Same interface as dist.HipBackend, CPU tensors + oracle raster."""

    def __init__(self, sh, pixels):
        from oracle import oracle as O
        self.O, self.sh, self.pixels = O, sh, pixels
        self.descs = [{"width": a.shape[1], "height": a.shape[0], "orientation": 1} for a in pixels]
        self.bands = {}
        self.staged = [i for i in sh.remote if not sh.in_place[i]]
        self.staging = {i: torch.empty((sh.boxes[i][3] - sh.boxes[i][1], sh.boxes[i][2] - sh.boxes[i][0], 4), dtype=torch.uint8) for i in self.staged}

    def new_canvas(self):
        return torch.full((self.sh.plan.canvas_h, self.sh.plan.canvas_w, 4), 0x5A, dtype=torch.uint8)

    @staticmethod
    def _ops(arr, n):
        return [{"kind": "fill", "m": list(o.m), "rect": list(o.d), "rgba": tuple(o.rgba)} if o.kind == 0 else
                {"kind": "hole", "rect": list(o.d)} if o.kind == 2 else
                {"kind": "draw", "image": o.image, "m": list(o.m), "s": list(o.s), "d": list(o.d)} for o in arr[:n]]

    def render_band(self, i, srcs):
        ops, n, clip = self.sh.band_ops(i)
        full = self.O.render_ops(self.sh.plan.canvas_w, self.sh.plan.canvas_h, self._ops(ops, n), self.descs,
                                 [np.zeros((d["height"], d["width"], 4), np.uint8) if s is None else s.numpy() for d, s in zip(self.descs, srcs)],
                                 self.sh.opts["filter"])
        x, y, w, h = clip
        self.bands[i] = torch.from_numpy(np.ascontiguousarray(full[y:y + h, x:x + w]))
        return self.bands[i]

    def render_root(self, srcs, canvas):
        ops, n, descs, n_img, staged = self.sh.root_ops()
        lst = self._ops(ops, n)
        d2 = list(self.descs) + [{"width": self.staging[i].shape[1], "height": self.staging[i].shape[0], "orientation": 1} for i in staged]
        px = [np.zeros((d["height"], d["width"], 4), np.uint8) if s is None else s.numpy() for d, s in zip(self.descs, srcs)]
        px += [self.staging[i].numpy() for i in staged]
        img = self.O.render_ops(self.sh.plan.canvas_w, self.sh.plan.canvas_h, [o for o in lst if o["kind"] != "hole"], d2, px, self.sh.opts["filter"])
        keep = np.ones(img.shape[:2], bool)
        for o in lst:
            if o["kind"] == "hole":
                x, y, w, h = [int(v) for v in o["rect"]]
                keep[y:y + h, x:x + w] = False
        c = canvas.numpy()
        c[keep] = img[keep]


def _worker(rank, world, port, direction, opts, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from imagestitching_amd import dist as D
        pixels = [U.rand_image(200 + i, h, w) for i, (w, h) in enumerate(SIZES)]
        imgs = U.hip_images(pixels)
        sh = D.ShardedStitch(imgs, direction, opts, rank, world, 0)
        be = OracleBackend(sh, pixels)
        srcs = [torch.from_numpy(a) if D.owner_of(i, world) == rank else None for i, a in enumerate(pixels)]
        canvas = be.new_canvas() if rank == 0 else None
        for _ in range(2):      # two steps: buffers are reusable
            D.run_step(sh, be, srcs, canvas, dist)
        dist.barrier()
        if rank == 0:
            np.save(out_path, canvas.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("direction,opts,expect_in_place", [
    ("vertical", {"filter": "bilinear", "mode": "min", "gap": 4}, True),       # full-width rows: received in place
    ("horizontal", {"filter": "bilinear", "mode": "max", "gap": 0}, False),    # column bands: staged + fused blit
    ("vertical", {"filter": "nearest", "mode": "original", "gap": 3}, False),  # centred rects: staged
])
def test_sharded_stitch_world2_matches_single_process(direction, opts, expect_in_place, tmp_path):
    from imagestitching_amd import dist as D
    pixels = [U.rand_image(200 + i, h, w) for i, (w, h) in enumerate(SIZES)]
    sh = D.ShardedStitch(U.hip_images(pixels), direction, opts, 0, 2, 0)
    assert sh.mine == [0, 2, 4] and sh.remote == [1, 3]
    assert all(sh.in_place[i] == expect_in_place for i in sh.remote)
    out = str(tmp_path / "canvas.npy")
    mp.spawn(_worker, args=(2, _free_port(), direction, opts, out), nprocs=2, join=True)
    got = np.load(out)
    ref, _, _ = U.oracle_stitch(pixels, direction, opts)
    assert np.array_equal(got, ref)


def test_round_robin_ownership_and_hole_ops():
    from imagestitching_amd import dist as D
    imgs = [{"width": 4032, "height": 3024}] * 9
    sh = D.ShardedStitch(imgs, "vertical", None, 0, 8, 0)
    assert sh.mine == [0, 8]                                   # GPU0 holds images 0 and 8
    assert [D.owner_of(i, 8) for i in range(9)] == [0, 1, 2, 3, 4, 5, 6, 7, 0]
    ops, n, descs, n_img, staged = sh.root_ops()
    kinds = [ops[k].kind for k in range(n)]
    assert kinds == [0, 1] + [2] * 7 + [1] and staged == []   # fill, own draw, 7 holes, own draw
    assert all(sh.in_place[i] for i in sh.remote)
    hole = ops[2]
    assert list(hole.d) == [0.0, 3024.0, 4032.0, 3024.0]
    sh5 = D.ShardedStitch([{"width": 8000, "height": 6000}] * 64, "vertical", None, 3, 8, 0)
    assert sh5.mine == list(range(3, 64, 8))                  # BASELINE configs[4]: 8 images per GPU


def test_overlapping_draws_are_refused():
    from imagestitching_amd import dist as D
    import imagestitching_amd as ist
    # the reference's orientation-7 placement draws image k one rect-height ABOVE its rect (utils/canvas.js:187-192):
    # with heights 30, 20, 50 image 2 lands on [0,50) and covers image 1 at [10,30)
    imgs = [{"width": 40, "height": h, "orientation": 7} for h in (30, 20, 50)]
    with pytest.raises(ist.StitchError):
        D.ShardedStitch(imgs, "vertical", None, 0, 2, 0)
    ok = D.ShardedStitch([{"width": 40, "height": 30, "orientation": 7}] * 2, "vertical", None, 0, 2, 0)
    assert sorted(ok.boxes) == [1]                            # image 0 is drawn entirely off-canvas


@pytest.mark.parametrize("world", [3, 4, 8])
def test_sharded_stitch_more_ranks_than_two(world, tmp_path):
    """several senders, uneven ownership (5 images over 3, 4 or 8 ranks; with 4 ranks rank 0 owns two, with 8 ranks three
    ranks own nothing and only join the barrier):
    the grouped send/recv batch must pair up per (sender, root) in image order"""
    pixels = [U.rand_image(200 + i, h, w) for i, (w, h) in enumerate(SIZES)]
    opts = {"filter": "bilinear", "mode": "min", "gap": 2}
    out = str(tmp_path / "canvas.npy")
    mp.spawn(_worker, args=(world, _free_port(), "vertical", opts, out), nprocs=world, join=True)
    ref, _, _ = U.oracle_stitch(pixels, "vertical", opts)
    assert np.array_equal(np.load(out), ref)
