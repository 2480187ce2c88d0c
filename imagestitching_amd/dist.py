"""Multi-GPU stitch, one process per GPU: disjoint parts per rank + ONE gather of finished bands to the root (RCCL over xGMI).

Reference anchor: the per-image loop of onStitch (pages/index/index.js:1439-1554) — iterations share only the
cursor, which the planner precomputes, so every draw's destination box (and any range of its rows) is an independent
unit.  Layout (BASELINE.json north_star / SURVEY.md section 8e): the job is cut into PARTS by the C-ABI's own
ist_shard_parts (the single-process device group, ist_mgpu_*, cuts the same way):
  split="image"  image i -> rank i mod world (BASELINE configs[3]: images round-robin; 9 images on 8 GPUs leave GPU 0
                 with two)
  split="band"   canvas rows dealt out so that every rank renders the same number of output pixels; a rank then needs
                 only the source rows its rows sample (still disjoint input subsets, up to one shared row at a cut)
  split="rows"   rank s owns a band of canvas rows ACROSS ALL DRAWS and renders the whole op list clipped to it; it holds
                 the rows of every image its band samples.  The unit that travels is the band - full canvas width for any
                 layout, so horizontal strips (index.js:1540-1553) and centred rects are received in place like vertical
                 ones, need no staging / placement launch, and can use the host sink; overlapping draws and anti-aliased
                 seams are allowed
  split="auto"   "image" when its parts are full-width (vertical min / max strips), else "rows" (the default)
Each rank renders its parts into compact bands; the root assembles the strip:
  * parts that span the full canvas width are contiguous byte ranges of the canvas -> received IN PLACE while the root's
    own launch runs (that launch leaves those rows untouched: op kind HOLE);
  * other parts (horizontal strips, centred 'original' rects) are received into staging bands; each is placed by its own
    small launch as soon as ITS receive has completed, so placement overlaps the receives still in flight.
The exchange is one grouped batch of point-to-point sends/recvs (RCCL has no gatherv): torch.distributed
batch_isend_irecv -> ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd.  No ring, no all-gather: xGMI is point to
point and every peer has its own link to the root.

The render backend is injected so that the sharding / assembly logic can be covered on CPU with gloo
(tests pass an oracle-backed renderer; the product default is the HIP path and nothing else).
"""
import ctypes as C
import importlib

from . import _lib as L

S = importlib.import_module(".stitch", __package__)   # the package attribute `stitch` is the function, not the module

OP_FILL, OP_DRAW, OP_HOLE = 0, 1, 2
_SPLITS = {"image": L.SPLIT_IMAGE, "band": L.SPLIT_BAND, "rows": L.SPLIT_ROWS, "auto": L.SPLIT_AUTO}
_SPLIT_NAMES = {v: k for k, v in _SPLITS.items()}


def owner_of(image_index, world):
    """BASELINE configs[3] (split="image"): images round-robin over the GPUs."""
    return image_index % world


class Part:
    """One unit of work: canvas box (X0, Y0, X1, Y1) of draw `op` (image `image`), rendered by `slot`."""
    __slots__ = ("index", "image", "op", "slot", "X0", "Y0", "X1", "Y1", "sy0", "sy1", "in_place")

    def __init__(self, index, c):
        self.index = index
        self.image, self.op, self.slot = c.image, c.op, c.slot
        self.X0, self.Y0, self.X1, self.Y1 = c.X0, c.Y0, c.X1, c.Y1
        self.sy0, self.sy1 = c.sy0, c.sy1
        self.in_place = bool(c.in_place)

    @property
    def shape(self):
        return (self.Y1 - self.Y0, self.X1 - self.X0, 4)

    @property
    def nbytes(self):
        return (self.Y1 - self.Y0) * (self.X1 - self.X0) * 4


class Band:
    """split="rows": the band of canvas rows [Y0, Y1) slot `slot` owns, across all draws - what that slot renders into one
    buffer and delivers in one piece.  Quacks like a Part (index, slot, box, in_place) for run_step / run_step_host_sink."""
    __slots__ = ("index", "image", "op", "slot", "X0", "Y0", "X1", "Y1", "sy0", "sy1", "in_place", "pieces")

    def __init__(self, index, slot, width, y0, y1, pieces):
        self.index, self.slot = index, slot
        self.image, self.op = -1, -1
        self.X0, self.Y0, self.X1, self.Y1 = 0, y0, width, y1
        self.sy0 = self.sy1 = 0
        self.in_place = True
        self.pieces = pieces                  # the (slot's rows) x (one draw's box) parts: which rows of which image the slot samples

    shape = Part.shape
    nbytes = Part.nbytes


def resolve_split(plan, filter_code, split):
    """the cut "auto" stands for on this plan (ist_shard_resolve)"""
    ops, n_ops = plan.ops()
    return _SPLIT_NAMES[L.check(L.lib.ist_shard_resolve(ops, n_ops, plan.canvas_w, plan.canvas_h, plan._descs, plan.n_images, filter_code, _SPLITS[split]))]


def row_cuts(canvas_h, n_slots):
    arr = (C.c_int32 * (n_slots + 1))()
    L.check(L.lib.ist_shard_row_cuts(canvas_h, n_slots, arr))
    return list(arr)


def shard_parts(plan, filter_code, n_slots, split):
    """ist_shard_parts through the C-ABI (pure CPU)."""
    ops, n_ops = plan.ops()
    cap = n_ops * n_slots + 8 if split in ("rows", "auto") else n_ops + n_slots + 8
    arr = (L.Part * cap)()
    cnt = C.c_int(0)
    L.check(L.lib.ist_shard_parts(ops, n_ops, plan.canvas_w, plan.canvas_h, plan._descs, plan.n_images, filter_code,
                                  n_slots, _SPLITS[split], arr, cap, C.byref(cnt)))
    return [Part(k, arr[k]) for k in range(cnt.value)]


class ShardedStitch:
    """One stitch job sharded over `world` ranks.  Construct on every rank with identical arguments."""

    def __init__(self, images, direction, opts=None, rank=0, world=1, root=0, split="auto"):
        self.rank, self.world, self.root, self.split = rank, world, root, split
        self.opts = S._merge(opts)
        self.plan = S.plan(images, direction, self.opts)
        if self.plan is None:
            raise ValueError("nothing to stitch")
        self.n = len(images)
        self.filter = S._filter_of(self.opts)
        # edge anti-aliasing makes neighbouring draws share a pixel row: ist_shard_parts refuses (IST_E_UNSUPPORTED)
        self.split = split = resolve_split(self.plan, self.filter, split)
        self.parts = shard_parts(self.plan, self.filter, world, split)
        self.pieces = self.parts              # the per-draw parts: what rows_needed reads
        if split == "rows":                   # the units that are rendered and delivered are the slots' bands
            cuts = row_cuts(self.plan.canvas_h, world)
            self.parts = [Band(k, s, self.plan.canvas_w, cuts[s], cuts[s + 1], [p for p in self.pieces if p.slot == s])
                          for k, s in enumerate(s for s in range(world) if cuts[s + 1] > cuts[s])]
        self.slot = (rank - root) % world
        self.mine = [p for p in self.parts if p.slot == self.slot]
        self.remote = [p for p in self.parts if p.slot != 0]

    def rank_of(self, part):
        return (part.slot + self.root) % self.world

    def rows_needed(self, slot=None):
        """{image: (first_row, end_row)} of the source rows the slot's parts sample: what that rank must hold."""
        slot = self.slot if slot is None else slot
        need = {}
        for p in self.pieces:
            if p.slot != slot:
                continue
            a, b = need.get(p.image, (p.sy0, p.sy1))
            need[p.image] = (min(a, p.sy0), max(b, p.sy1))
        return need

    def root_rows(self):
        """Host sink (the reference's export is host-destined, index.js:1577-1581): when every remote part spans the canvas
        width, each rank can DMA its finished bands straight into its byte range of a host canvas; the root then delivers
        only the canvas rows NO remote part covers.  Returns those row ranges [(y0, y1), ...], or None when some remote part
        is not full-width (horizontal strips, centred rects: they need the gather)."""
        if any(not p.in_place for p in self.remote):
            return None
        out, y = [], 0
        for a, b in sorted((p.Y0, p.Y1) for p in self.remote):
            if a > y:
                out.append((y, a))
            y = max(y, b)
        if y < self.plan.canvas_h:
            out.append((y, self.plan.canvas_h))
        return out

    # ---- op lists ------------------------------------------------------------------------------------------------
    def band_ops(self, part):
        """ops + clip for rendering a part on its owner: white fill + that draw, clipped to the part's box; a band of the
        rows split: the whole op list minus the draws that do not reach its rows, clipped to the band."""
        ops, n_ops = self.plan.ops()
        clip = (part.X0, part.Y0, part.X1 - part.X0, part.Y1 - part.Y0)
        if isinstance(part, Band):
            mine = {q.op for q in part.pieces}
            sel = [ops[k] for k in range(n_ops) if ops[k].kind != OP_DRAW or k in mine]
            return (L.Op * len(sel))(*sel), len(sel), clip
        sel = (L.Op * 2)(ops[0], ops[part.op])
        return sel, 2, clip

    def root_ops(self):
        """ops of the root's own fused launch: the fill, every draw the root owns a part of, and a HOLE over every part
        another rank delivers (listed last: the launch then writes nothing there, whatever lies under it)."""
        ops, n_ops = self.plan.ops()
        own = {p.op for p in self.pieces if p.slot == 0}
        out = [ops[k] for k in range(n_ops) if ops[k].kind != OP_DRAW or k in own]
        for p in self.remote:
            r = L.Op()
            r.kind, r.image = OP_HOLE, -1
            r.m[:] = [1.0, 0.0, 0.0, 1.0, 0.0, 0.0]
            r.d[:] = [float(p.X0), float(p.Y0), float(p.X1 - p.X0), float(p.Y1 - p.Y0)]
            out.append(r)
        return (L.Op * len(out))(*out), len(out)

    @staticmethod
    def place_ops(part):
        """ops + descs + clip of the launch that places one staged band: a 1:1 draw of the band at the part's box."""
        r = L.Op()
        r.kind, r.image = OP_DRAW, 0
        r.m[:] = [1.0, 0.0, 0.0, 1.0, 0.0, 0.0]
        w, h = part.X1 - part.X0, part.Y1 - part.Y0
        r.s[:] = [0.0, 0.0, float(w), float(h)]
        r.d[:] = [float(part.X0), float(part.Y0), float(w), float(h)]
        descs = (L.ImageDesc * 1)(L.ImageDesc(w, h, 1, 0, 0, 1, 0))          # a finished band: opaque
        return (L.Op * 1)(r), 1, descs, (part.X0, part.Y0, w, h)


class SourceRows:
    """Rows [first_row, first_row + tensor.shape[0]) of an image: what a rank holds of an image it renders a band of.
    The buffer must be readable for 16 bytes past its last row (allocate one spare row, see alloc_rows)."""

    def __init__(self, tensor, first_row=0):
        self.tensor, self.first_row = tensor, int(first_row)


def alloc_rows(torch, rows, width, device):
    """An HxWx4 uint8 view of a buffer with one spare row behind it (the resample paths may read up to 12 bytes past the
    last sampled pixel of a row)."""
    return torch.empty((rows + 1, width, 4), dtype=torch.uint8, device=device)[:rows]


class HipBackend:
    """Product backend: HIP kernels through the C-ABI, torch CUDA tensors for memory, torch.distributed (RCCL)."""

    def __init__(self, sharded, device):
        import torch
        self.torch = torch
        self.sh = sharded
        self.device = torch.device("cuda", device)
        self.st = S.Stitcher(device)
        sh = sharded
        self.band_jobs, self.bands = {}, {}
        self.place_jobs, self.staging = {}, {}
        if sh.slot != 0:
            for p in sh.mine:
                ops, n_ops, clip = sh.band_ops(p)
                self.band_jobs[p.index] = self.st.compile_ops(sh.plan.canvas_w, sh.plan.canvas_h, ops, n_ops, sh.plan._descs, sh.n, sh.filter, clip=clip)
                self.bands[p.index] = torch.empty(p.shape, dtype=torch.uint8, device=self.device)
        else:
            ops, n_ops = sh.root_ops()
            self.root_job = self.st.compile_ops(sh.plan.canvas_w, sh.plan.canvas_h, ops, n_ops, sh.plan._descs, sh.n, sh.filter)
            for p in sh.remote:
                if p.in_place:
                    continue
                pops, pn, pdescs, clip = sh.place_ops(p)
                self.place_jobs[p.index] = self.st.compile_ops(sh.plan.canvas_w, sh.plan.canvas_h, pops, pn, pdescs, 1, "nearest", clip=clip)
                self.staging[p.index] = torch.empty(p.shape, dtype=torch.uint8, device=self.device)

    def new_canvas(self):
        p = self.sh.plan
        return self.torch.empty((p.canvas_h, p.canvas_w, 4), dtype=self.torch.uint8, device=self.device)

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    @staticmethod
    def _ptrs(srcs):
        """device pointer of row 0 (biased for partial holdings) and pitch per image"""
        ptrs, pitches = [], []
        for s in srcs:
            if s is None:
                ptrs.append(0); pitches.append(0)
            elif isinstance(s, SourceRows):
                pitch = s.tensor.stride(0)
                ptrs.append(s.tensor.data_ptr() - s.first_row * pitch); pitches.append(pitch)
            else:
                ptrs.append(s.data_ptr()); pitches.append(s.stride(0))
        return ptrs, pitches

    def render_band(self, part, srcs):
        """Owner side: the part's canvas box rendered straight into a compact buffer (the launch addresses the band as if
        it were the canvas: dst is biased by -(Y0*pitch + X0*4), the clip keeps writes inside)."""
        band = self.bands[part.index]
        pitch = band.stride(0)
        ptrs, pitches = self._ptrs(srcs)
        self.band_jobs[part.index].launch_ptrs(ptrs, pitches, band.data_ptr() - (part.Y0 * pitch + part.X0 * 4), pitch, self._stream())
        return band

    def render_root(self, srcs, canvas):
        ptrs, pitches = self._ptrs(srcs)
        self.root_job.launch_ptrs(ptrs, pitches, canvas.data_ptr(), canvas.stride(0), self._stream())

    def place(self, part, canvas):
        band = self.staging[part.index]
        self.place_jobs[part.index].launch_ptrs([band.data_ptr()], [band.stride(0)], canvas.data_ptr(), canvas.stride(0), self._stream())


class HostRows:
    """The root's share of a host-destined canvas for run_step_host_sink: one pinned buffer per row range of sh.root_rows()
    instead of a whole host canvas (the 64 x 48 MP canvas of BASELINE configs[4] is 12.3 GB; the root delivers 1/8 of it).
    Indexed like the full canvas: rows[a:b] for exactly the ranges of sh.root_rows()."""

    def __init__(self, torch, sh, pin=True):
        w = sh.plan.canvas_w
        self.rows = {}
        for a, b in sh.root_rows():
            t = torch.empty((b - a, w, 4), dtype=torch.uint8)
            self.rows[(a, b)] = t.pin_memory() if pin else t

    def __getitem__(self, sl):
        return self.rows[(sl.start, sl.stop)]


def run_step_host_sink(sh, backend, srcs, canvas, host_bands, host_canvas):
    """One sharded stitch whose result is HOST-destined: no exchange at all.  Every rank renders its parts and copies each
    finished band into `host_bands[part.index]` (pinned; in a deployment: the part's byte range of one shared pinned canvas);
    the root renders its own launch into `canvas` and copies the rows no remote part covers into `host_canvas`.  Asynchronous
    on the current stream.  Requires sh.root_rows() is not None."""
    if sh.slot == 0:
        backend.render_root(srcs, canvas)
        for a, b in sh.root_rows():
            host_canvas[a:b].copy_(canvas[a:b], non_blocking=True)
        return
    for p in sh.mine:
        host_bands[p.index].copy_(backend.render_band(p, srcs), non_blocking=True)


def run_step(sh, backend, srcs, canvas, dist):
    """One sharded stitch.  srcs: per-image source buffers (None for images this rank holds nothing of; SourceRows for
    partial holdings).  canvas: root's output buffer (None elsewhere).  dist: torch.distributed (initialised).
    Returns the canvas on root."""
    ops = []
    if sh.slot == 0:
        staged = []
        for p in sh.remote:
            if p.in_place:
                ops.append(dist.P2POp(dist.irecv, canvas[p.Y0:p.Y1], sh.rank_of(p)))
            else:
                ops.append(dist.P2POp(dist.irecv, backend.staging[p.index], sh.rank_of(p)))
            staged.append(None if p.in_place else p)
        # post the receives first: they then run beside the root's own launch, which touches none of those pixels
        reqs = dist.batch_isend_irecv(ops) if ops else []
        backend.render_root(srcs, canvas)
        # one request per batch (NCCL coalesces the group) or one per op (gloo): either way every receive has landed once
        # the requests it may belong to are done; staged bands are placed as their receive completes
        per_op = len(reqs) == len(ops)
        if not per_op:
            for r in reqs:
                r.wait()
        for k, p in enumerate(staged):
            if per_op:
                reqs[k].wait()
            if p is not None:
                backend.place(p, canvas)
        return canvas
    for p in sh.mine:
        band = backend.render_band(p, srcs)
        ops.append(dist.P2POp(dist.isend, band, sh.root))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    for r in reqs:
        r.wait()
    return None
