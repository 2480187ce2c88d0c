"""Multi-GPU stitch: disjoint image subsets per rank + ONE gather of finished bands to the root (RCCL over xGMI).

Reference anchor: the per-image loop of onStitch (pages/index/index.js:1439-1554) — iterations share only the
cursor, which the planner precomputes, so every image's destination box is an independent unit.  Layout
(BASELINE.json north_star / SURVEY.md section 8e): one process per GPU, image i -> rank i mod world; each rank renders
its images into compact bands (canvas-space boxes); the root assembles the strip:
  * boxes that span the full canvas width are contiguous byte ranges of the canvas -> received IN PLACE
    (the root's own launch leaves those rows untouched: op kind HOLE);
  * other boxes (horizontal strips, centred 'original' rects) are received into staging bands and placed by the
    root's fused launch as 1:1 draws (one extra read+write of the band on the root).
The exchange is one grouped batch of point-to-point sends/recvs (RCCL has no gatherv): torch.distributed
batch_isend_irecv -> ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd.

The render backend is injected so that the sharding / assembly logic can be covered on CPU with gloo
(tests pass an oracle-backed renderer; the product default is the HIP path and nothing else).
"""
import importlib

from . import _lib as L

S = importlib.import_module(".stitch", __package__)   # the package attribute `stitch` is the function, not the module

OP_FILL, OP_DRAW, OP_HOLE = 0, 1, 2


def owner_of(image_index, world):
    """BASELINE configs[3]: images round-robin over the GPUs."""
    return image_index % world


def _boxes(plan, filter_name):
    """Canvas-space pixel boxes of every draw, from the C-ABI's own resolve step (ist_op_box: pure CPU)."""
    import ctypes as C
    ops, n_ops = plan.ops()
    boxes = {}
    box = (C.c_int32 * 4)()
    f = {"nearest": L.FILTER_NEAREST, "bilinear": L.FILTER_BILINEAR}[filter_name]
    for k in range(1, n_ops):
        rc = L.check(L.lib.ist_op_box(C.byref(ops[k]), plan.canvas_w, plan.canvas_h, f, box))
        if rc == 0 and box[2] > box[0] and box[3] > box[1]:      # a draw clipped away entirely (the reference's orientation-7 first image) has no band
            boxes[ops[k].image] = (box[0], box[1], box[2], box[3], k)
    return boxes


def _overlap(p, q):
    return p[0] < q[2] and q[0] < p[2] and p[1] < q[3] and q[1] < p[3]


class ShardedStitch:
    """One stitch job sharded over `world` ranks.  Construct on every rank with identical arguments."""

    def __init__(self, images, direction, opts=None, rank=0, world=1, root=0):
        self.rank, self.world, self.root = rank, world, root
        self.opts = S._merge(opts)
        if self.opts.get("edgeAA"):
            raise L.StitchError(-7, "edge anti-aliasing blends neighbouring images in one pixel row: stitch on one GPU")
        self.plan = S.plan(images, direction, self.opts)
        if self.plan is None:
            raise ValueError("nothing to stitch")
        self.n = len(images)
        self.boxes = _boxes(self.plan, self.opts["filter"])
        bl = [self.boxes[i] for i in sorted(self.boxes)]
        for i in range(len(bl)):
            for j in range(i + 1, len(bl)):
                if _overlap(bl[i], bl[j]):
                    raise L.StitchError(-7, "overlapping draws cannot be sharded across GPUs (stitch on one GPU)")
        self.mine = [i for i in range(self.n) if owner_of(i, world) == rank and i in self.boxes]
        self.remote = [i for i in range(self.n) if owner_of(i, world) != root and i in self.boxes]
        cw = self.plan.canvas_w
        # a box that spans the full width is a contiguous byte range of the canvas -> in-place receive
        self.in_place = {i: (self.boxes[i][0] == 0 and self.boxes[i][2] == cw) for i in self.boxes}

    # ---- op lists ------------------------------------------------------------------------------------------------
    def band_ops(self, i):
        """ops + clip for rendering image i's band on its owner: white fill + that draw, clipped to its box."""
        ops, _ = self.plan.ops()
        X0, Y0, X1, Y1, k = self.boxes[i]
        sel = (L.Op * 2)(ops[0], ops[k])
        return sel, 2, (X0, Y0, X1 - X0, Y1 - Y0)

    def root_ops(self):
        """ops for the root's single fused launch: fill, its own draws, HOLEs for in-place bands, 1:1 draws for
        staged bands.  Returns (ops, n_ops, descs, n_images, staged) where staged[j] = image index behind extra
        source slot n + j."""
        ops, n_ops = self.plan.ops()
        staged = [i for i in self.remote if not self.in_place[i]]
        out = [ops[0]]
        for k in range(1, n_ops):
            o = ops[k]
            i = o.image
            if owner_of(i, self.world) == self.root:
                out.append(o)
                continue
            if i not in self.boxes:
                continue
            X0, Y0, X1, Y1, _ = self.boxes[i]
            r = L.Op()
            r.m[:] = [1.0, 0.0, 0.0, 1.0, 0.0, 0.0]
            r.d[:] = [float(X0), float(Y0), float(X1 - X0), float(Y1 - Y0)]
            if self.in_place[i]:
                r.kind, r.image = OP_HOLE, -1
            else:
                r.kind, r.image = OP_DRAW, self.n + staged.index(i)
                r.s[:] = [0.0, 0.0, float(X1 - X0), float(Y1 - Y0)]
            out.append(r)
        descs = (L.ImageDesc * (self.n + len(staged)))()
        for i in range(self.n):
            descs[i] = self.plan._descs[i]
        for j, i in enumerate(staged):
            X0, Y0, X1, Y1, _ = self.boxes[i]
            descs[self.n + j] = L.ImageDesc(X1 - X0, Y1 - Y0, 1, 0, 0, 1, 0)   # finished band: opaque
        arr = (L.Op * len(out))(*out)
        return arr, len(out), descs, self.n + len(staged), staged


class HipBackend:
    """Product backend: HIP kernels through the C-ABI, torch CUDA tensors for memory, torch.distributed (RCCL)."""

    def __init__(self, sharded, device):
        import torch
        self.torch = torch
        self.sh = sharded
        self.device = torch.device("cuda", device)
        self.st = S.Stitcher(device)
        sh = sharded
        f = sh.opts["filter"]
        self.band_jobs, self.bands = {}, {}
        if sh.rank != sh.root:
            for i in sh.mine:
                ops, n_ops, clip = sh.band_ops(i)
                self.band_jobs[i] = self.st.compile_ops(sh.plan.canvas_w, sh.plan.canvas_h, ops, n_ops, sh.plan._descs, sh.n, f, clip=clip)
                X0, Y0, X1, Y1, _ = sh.boxes[i]
                self.bands[i] = torch.empty((Y1 - Y0, X1 - X0, 4), dtype=torch.uint8, device=self.device)
        else:
            ops, n_ops, descs, n_img, staged = sh.root_ops()
            self.root_job = self.st.compile_ops(sh.plan.canvas_w, sh.plan.canvas_h, ops, n_ops, descs, n_img, f)
            self.staged = staged
            self.staging = {}
            for i in staged:
                X0, Y0, X1, Y1, _ = sh.boxes[i]
                self.staging[i] = torch.empty((Y1 - Y0, X1 - X0, 4), dtype=torch.uint8, device=self.device)

    def new_canvas(self):
        p = self.sh.plan
        return self.torch.empty((p.canvas_h, p.canvas_w, 4), dtype=self.torch.uint8, device=self.device)

    def render_band(self, i, srcs):
        """Owner side: band i = canvas box of image i, rendered straight into a compact buffer (the launch addresses
        the band as if it were the canvas: dst is biased by -(Y0*pitch + X0*4), the clip keeps writes inside)."""
        X0, Y0, X1, Y1, _ = self.sh.boxes[i]
        band = self.bands[i]
        pitch = band.stride(0)
        ptrs = [0 if t is None else t.data_ptr() for t in srcs]
        pitches = [0 if t is None else t.stride(0) for t in srcs]
        self.band_jobs[i].launch_ptrs(ptrs, pitches, band.data_ptr() - (Y0 * pitch + X0 * 4), pitch,
                                      self.torch.cuda.current_stream(self.device).cuda_stream)
        return band

    def render_root(self, srcs, canvas):
        full = list(srcs) + [self.staging[i] for i in self.staged]
        self.root_job.launch(full, canvas)


def run_step(sh, backend, srcs, canvas, dist):
    """One sharded stitch.  srcs: per-image source buffers (None for images this rank does not own).
    canvas: root's output buffer (None elsewhere).  dist: torch.distributed (initialised).  Returns the canvas on root."""
    ops = []
    if sh.rank == sh.root:
        recv_staged = False
        for i in sh.remote:
            X0, Y0, X1, Y1, _ = sh.boxes[i]
            if sh.in_place[i]:
                ops.append(dist.P2POp(dist.irecv, canvas[Y0:Y1], owner_of(i, sh.world)))
            else:
                ops.append(dist.P2POp(dist.irecv, backend.staging[i], owner_of(i, sh.world)))
                recv_staged = True
        # post the receives first: the communication stream then runs beside the root's own launch, which never
        # touches the in-place rows (op kind HOLE)
        reqs = dist.batch_isend_irecv(ops) if ops else []
        if not recv_staged:
            backend.render_root(srcs, canvas)
        for r in reqs:
            r.wait()
        if recv_staged:
            backend.render_root(srcs, canvas)          # staged bands are sources of the fused launch
        return canvas
    for i in sh.mine:
        band = backend.render_band(i, srcs)
        ops.append(dist.P2POp(dist.isend, band, sh.root))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    for r in reqs:
        r.wait()
    return None
