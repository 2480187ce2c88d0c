// ist_shard.cpp — cuts one stitch into PARTS for a group of GPUs (pure CPU, no HIP).
//
// Reference anchor: the per-image loop of onStitch (pages/index/index.js:1439-1554) — iterations share only the cursor,
// which the planner precomputes, so every draw's destination box is an independent unit of work, and so is any
// sub-range of its canvas rows.  BASELINE.json north_star: disjoint input-image subsets per GPU, one gather to the root.
//   IST_SPLIT_IMAGE  image i -> slot i mod n (BASELINE configs[3]: images round-robin)
//   IST_SPLIT_BAND   the draws' canvas rows are dealt out in canvas order so that every slot renders the same number of
//                    output pixels (9 images over 8 GPUs leave a 2-image straggler otherwise);
//                    a slot then needs only the source rows its canvas rows sample
//   IST_SPLIT_ROWS   slot s owns canvas rows [cuts[s], cuts[s+1]) ACROSS ALL DRAWS (ist_shard_row_cuts: equal rows, cuts on
//                    multiples of 8) and renders the whole op list clipped to them; it holds rows [sy0, sy1) of every image
//                    its rows sample.  A horizontal strip (index.js:1540-1553: every rect spans the canvas height) or a
//                    centred 'original' rect is a COLUMN band under the two cuts above - never a contiguous byte range of
//                    the canvas; under this one every slot's band is full-width whatever the layout: received in place,
//                    no staging, no placement launch, and a host sink is always possible.  Draws may overlap and edges may
//                    be anti-aliased: every canvas pixel has one owner, who paints the whole stack there.
//   IST_SPLIT_AUTO   IMAGE when that cut yields full-width parts only (vertical min / max strips), else ROWS
// Used by the single-process device group (ist_mgpu.cpp) and, through the C-ABI, by the one-process-per-GPU layout
// (imagestitching_amd/dist.py), so both cut a job the same way.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ist_internal.h"

namespace ist {

namespace {

// source index range one axis of a draw touches over canvas coordinates [lo, hi) (inclusive result, clamped)
void tap_range(double k, double o, int lo, int hi, int clo, int chi, int filter, int* a, int* b) {
  if (filter == IST_FILTER_AREA && std::fabs(k) > 1.0) {      // a box of width |k| around every sample position
    const double half = 0.5 * std::fabs(k);
    const double c0 = k * (static_cast<double>(lo) + 0.5) + o, c1 = k * (static_cast<double>(hi - 1) + 0.5) + o;
    const double s0 = std::min(c0, c1) - half, s1 = std::max(c0, c1) + half;
    const int64_t i0 = static_cast<int64_t>(std::min(std::max(std::floor(s0), -4.0e9), 4.0e9)), i1 = static_cast<int64_t>(std::min(std::max(std::ceil(s1), -4.0e9), 4.0e9)) - 1;
    *a = static_cast<int>(std::min<int64_t>(std::max<int64_t>(i0, clo), chi));
    *b = static_cast<int>(std::min<int64_t>(std::max<int64_t>(i1, clo), chi));
    return;
  }
  if (filter == IST_FILTER_AREA) filter = IST_FILTER_BILINEAR;
  auto first_tap = [&](int w) {
    const double s = k * (static_cast<double>(w) + 0.5) + o;
    double fl = std::floor(filter == IST_FILTER_BILINEAR ? s - 0.5 : s);
    fl = std::min(std::max(fl, -4.0e9), 4.0e9);
    return static_cast<int64_t>(fl);
  };
  const int64_t t0 = first_tap(lo), t1 = first_tap(hi - 1);           // monotonic in w: the ends bound the range
  const int64_t span = filter == IST_FILTER_BILINEAR ? 1 : 0;
  const int64_t mn = std::min(t0, t1), mx = std::max(t0, t1) + span;
  *a = static_cast<int>(std::min<int64_t>(std::max<int64_t>(mn, clo), chi));
  *b = static_cast<int>(std::min<int64_t>(std::max<int64_t>(mx, clo), chi));
}

}  // namespace

}  // namespace ist

using namespace ist;

extern "C" int ist_shard_row_cuts(int64_t canvas_h, int n_slots, int32_t* cuts) {
  if (!cuts || n_slots < 1 || n_slots > 4096 || canvas_h < 1 || canvas_h > 2147483647LL) return fail(IST_E_INVALID, "ist_shard_row_cuts: bad argument");
  // equal rows per slot, every cut on a multiple of 8 rows (the tile height of the copy path: no tile straddles two owners);
  // a canvas shorter than 8 * n_slots rows leaves some slots without rows (never the root)
  cuts[0] = 0;
  for (int s = 1; s < n_slots; ++s) {
    int64_t c = (static_cast<__int128>(canvas_h) * s + n_slots - 1) / n_slots;
    c = (c + 7) & ~7LL;                         // (rounded UP: the root's band is never the empty one)
    cuts[s] = static_cast<int32_t>(std::min<int64_t>(std::max<int64_t>(c, cuts[s - 1]), canvas_h));
  }
  cuts[n_slots] = static_cast<int32_t>(canvas_h);
  return IST_OK;
}

extern "C" int ist_shard_parts(const ist_op* ops, int n_ops, int64_t canvas_w, int64_t canvas_h, const ist_image_desc* images,
                               int n_images, int filter, int n_slots, int split, ist_part* parts, int max_parts, int* n_parts) {
  if (!ops || n_ops < 0 || !images || !parts || !n_parts) return fail(IST_E_INVALID, "ist_shard_parts: NULL argument");
  if (n_slots < 1 || n_slots > 4096) return fail(IST_E_INVALID, "ist_shard_parts: bad slot count");
  if (split != IST_SPLIT_IMAGE && split != IST_SPLIT_BAND && split != IST_SPLIT_ROWS && split != IST_SPLIT_AUTO) return fail(IST_E_INVALID, "ist_shard_parts: unknown split");
  if (canvas_w < 1 || canvas_h < 1 || canvas_w > (1 << 29) || canvas_h > 2147483647LL) return fail(IST_E_OUTPUT_SIZE, "输出尺寸计算失败: canvas size out of range");
  const bool aa = (filter & IST_FILTER_EDGE_AA) != 0;
  const int f = filter & 0xFF;
  if (f != IST_FILTER_NEAREST && f != IST_FILTER_BILINEAR && f != IST_FILTER_AREA) return fail(IST_E_INVALID, "unknown filter");
  *n_parts = 0;
  struct Draw { int op; DevOp r; };
  std::vector<Draw> draws;
  for (int k = 0; k < n_ops; ++k) {
    if (ops[k].kind == IST_OP_HOLE) return fail(IST_E_UNSUPPORTED, "an op list that already reserves regions cannot be sharded again");
    if (ops[k].kind != IST_OP_DRAW) continue;
    const int i = ops[k].image;
    if (i < 0 || i >= n_images) return fail(IST_E_INVALID, "op refers to a missing image");
    const int iw = images[i].bmp_width > 0 ? images[i].bmp_width : images[i].width;
    const int ih = images[i].bmp_height > 0 ? images[i].bmp_height : images[i].height;
    if (iw < 1 || ih < 1) return fail(IST_E_DECODE, "图片" + std::to_string(i) + "解码异常");
    Draw d; d.op = k;
    const int rc = resolve_op(ops[k], canvas_w, canvas_h, iw, ih, &d.r, aa);
    if (rc < 0) return rc;
    if (rc > 0) continue;                      // draws nothing (clipped away entirely): no part
    draws.push_back(d);
  }
  // parts of the per-draw cuts must not share canvas pixels: each is rendered by one GPU over the background alone
  bool overlap = false;
  for (size_t a = 0; a < draws.size() && !overlap; ++a)
    for (size_t b = a + 1; b < draws.size(); ++b) {
      const DevOp& p = draws[a].r; const DevOp& q = draws[b].r;
      if (p.X0 < q.X1 && q.X0 < p.X1 && p.Y0 < q.Y1 && q.Y0 < p.Y1) { overlap = true; break; }
    }
  if (split == IST_SPLIT_AUTO) {
    bool full_width = !overlap;
    for (const Draw& d : draws) full_width = full_width && d.r.X0 == 0 && static_cast<int64_t>(d.r.X1) == canvas_w;
    split = full_width ? IST_SPLIT_IMAGE : IST_SPLIT_ROWS;
  }
  if (overlap && split != IST_SPLIT_ROWS)
    return fail(IST_E_UNSUPPORTED, aa ? "edge anti-aliasing blends neighbouring images in one pixel row: split by rows (IST_SPLIT_ROWS) or stitch on one GPU"
                                      : "overlapping draws cannot be sharded draw by draw: split by rows (IST_SPLIT_ROWS) or stitch on one GPU");
  auto emit = [&](const Draw& d, int slot, int Y0, int Y1) -> int {
    if (*n_parts >= max_parts) return fail(IST_E_INVALID, "ist_shard_parts: part table too small");
    ist_part& p = parts[(*n_parts)++];
    std::memset(&p, 0, sizeof p);
    p.image = d.r.image; p.op = d.op; p.slot = slot;
    p.X0 = d.r.X0; p.X1 = d.r.X1; p.Y0 = Y0; p.Y1 = Y1;
    p.in_place = (split == IST_SPLIT_ROWS || (p.X0 == 0 && static_cast<int64_t>(p.X1) == canvas_w)) ? 1 : 0;     // (ROWS: the slot's BAND is the unit, always full-width)
    // the source rows / columns this canvas box samples.  Source x is driven by canvas X (or canvas Y after a quarter turn)
    const bool sw = (d.r.flags & OPF_SWAP) != 0;
    int a, b;
    tap_range(d.r.kx, d.r.ox, sw ? Y0 : p.X0, sw ? Y1 : p.X1, d.r.cx0, d.r.cx1, f, &a, &b);
    p.sx0 = a; p.sx1 = b + 1;
    tap_range(d.r.ky, d.r.oy, sw ? p.X0 : Y0, sw ? p.X1 : Y1, d.r.cy0, d.r.cy1, f, &a, &b);
    p.sy0 = a; p.sy1 = b + 1;
    return IST_OK;
  };
  if (split == IST_SPLIT_IMAGE) {
    for (const Draw& d : draws) { const int rc = emit(d, d.r.image % n_slots, d.r.Y0, d.r.Y1); if (rc) return rc; }
    return IST_OK;
  }
  if (split == IST_SPLIT_ROWS) {
    // slot by slot, draws in op order: part = (the slot's rows) x (one draw's box).  The unit that is rendered and delivered is
    // the slot's whole band [0, canvas_w) x [cuts[s], cuts[s+1]) - the parts say which rows of which image the slot must hold.
    std::vector<int32_t> cuts(static_cast<size_t>(n_slots) + 1);
    const int rc0 = ist_shard_row_cuts(canvas_h, n_slots, cuts.data());
    if (rc0) return rc0;
    for (int s = 0; s < n_slots; ++s) {
      const int y0 = cuts[static_cast<size_t>(s)], y1 = cuts[static_cast<size_t>(s) + 1];
      if (y1 <= y0) continue;
      for (const Draw& d : draws) {
        const int a = std::max(d.r.Y0, y0), b = std::min(d.r.Y1, y1);
        if (b <= a) continue;
        const int rc = emit(d, s, a, b);
        if (rc) return rc;
      }
    }
    return IST_OK;
  }
  // IST_SPLIT_BAND: equal output pixels per slot, dealt in canvas (= op) order.  Cuts fall on multiples of 8 rows inside
  // a box (the tile height of the copy path), so no tile straddles two owners.
  int64_t total = 0;
  for (const Draw& d : draws) total += static_cast<int64_t>(d.r.X1 - d.r.X0) * (d.r.Y1 - d.r.Y0);
  int slot = 0;
  int64_t given = 0;                           // pixels dealt to slots 0..slot so far
  for (const Draw& d : draws) {
    const int64_t w = d.r.X1 - d.r.X0;
    int y = d.r.Y0;
    while (y < d.r.Y1) {
      // slot s ends where the running total reaches (s + 1) * total / n_slots
      const int64_t quota_end = (static_cast<__int128>(total) * (slot + 1) + n_slots - 1) / n_slots;
      int64_t rows = d.r.Y1 - y;
      if (slot < n_slots - 1) {
        const int64_t room = std::max<int64_t>(quota_end - given, 0);
        int64_t fit = (room + w - 1) / w;                                   // rows that reach the quota
        fit = ((y - d.r.Y0 + fit + 7) & ~7LL) - (y - d.r.Y0);               // cut on a multiple of 8 rows of the box
        if (fit < rows) rows = std::max<int64_t>(fit, 0);
      }
      if (rows > 0) {
        const int rc = emit(d, slot, y, static_cast<int>(y + rows));
        if (rc) return rc;
        given += rows * w;
        y += static_cast<int>(rows);
      }
      if (slot < n_slots - 1 && given >= quota_end) ++slot;
      else if (rows == 0) ++slot;             // (cannot happen with a positive quota; guards the loop)
    }
  }
  return IST_OK;
}

extern "C" int ist_shard_resolve(const ist_op* ops, int n_ops, int64_t canvas_w, int64_t canvas_h, const ist_image_desc* images, int n_images,
                                 int filter, int split) {
  if (split == IST_SPLIT_IMAGE || split == IST_SPLIT_BAND || split == IST_SPLIT_ROWS) return split;
  if (split != IST_SPLIT_AUTO) return fail(IST_E_INVALID, "ist_shard_resolve: unknown split");
  if (!ops || n_ops < 0 || !images) return fail(IST_E_INVALID, "ist_shard_resolve: NULL argument");
  // what ist_shard_parts does with IST_SPLIT_AUTO: by image when that yields full-width parts only, else by rows
  std::vector<ist_part> parts(static_cast<size_t>(std::max(n_ops, 1)));
  int n = 0;
  const int rc = ist_shard_parts(ops, n_ops, canvas_w, canvas_h, images, n_images, filter, 1, IST_SPLIT_IMAGE, parts.data(), static_cast<int>(parts.size()), &n);
  if (rc == IST_E_UNSUPPORTED) return IST_SPLIT_ROWS;          // overlapping draws
  if (rc != IST_OK) return rc;
  for (int k = 0; k < n; ++k) if (!parts[static_cast<size_t>(k)].in_place) return IST_SPLIT_ROWS;
  return IST_SPLIT_IMAGE;
}
