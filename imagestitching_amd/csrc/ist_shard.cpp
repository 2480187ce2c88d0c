// ist_shard.cpp — cuts one stitch into PARTS for a group of GPUs (pure CPU, no HIP).
//
// Reference anchor: the per-image loop of onStitch (pages/index/index.js:1439-1554) — iterations share only the cursor,
// which the planner precomputes, so every draw's destination box is an independent unit of work, and so is any
// sub-range of its canvas rows.  BASELINE.json north_star: disjoint input-image subsets per GPU, one gather to the root.
//   IST_SPLIT_IMAGE  image i -> slot i mod n (BASELINE configs[3]: images round-robin)
//   IST_SPLIT_BAND   the draws' canvas rows are dealt out in canvas order so that every slot renders the same number of
//                    output pixels (9 images over 8 GPUs leave a 2-image straggler otherwise);
//                    a slot then needs only the source rows its canvas rows sample
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

extern "C" int ist_shard_parts(const ist_op* ops, int n_ops, int64_t canvas_w, int64_t canvas_h, const ist_image_desc* images,
                               int n_images, int filter, int n_slots, int split, ist_part* parts, int max_parts, int* n_parts) {
  if (!ops || n_ops < 0 || !images || !parts || !n_parts) return fail(IST_E_INVALID, "ist_shard_parts: NULL argument");
  if (n_slots < 1 || n_slots > 4096) return fail(IST_E_INVALID, "ist_shard_parts: bad slot count");
  if (split != IST_SPLIT_IMAGE && split != IST_SPLIT_BAND) return fail(IST_E_INVALID, "ist_shard_parts: unknown split");
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
  // parts must not share canvas pixels: each is rendered by one GPU over the background alone
  for (size_t a = 0; a < draws.size(); ++a)
    for (size_t b = a + 1; b < draws.size(); ++b) {
      const DevOp& p = draws[a].r; const DevOp& q = draws[b].r;
      if (p.X0 < q.X1 && q.X0 < p.X1 && p.Y0 < q.Y1 && q.Y0 < p.Y1)
        return fail(IST_E_UNSUPPORTED, aa ? "edge anti-aliasing blends neighbouring images in one pixel row: stitch on one GPU"
                                          : "overlapping draws cannot be sharded across GPUs (stitch on one GPU)");
    }
  auto emit = [&](const Draw& d, int slot, int Y0, int Y1) -> int {
    if (*n_parts >= max_parts) return fail(IST_E_INVALID, "ist_shard_parts: part table too small");
    ist_part& p = parts[(*n_parts)++];
    std::memset(&p, 0, sizeof p);
    p.image = d.r.image; p.op = d.op; p.slot = slot;
    p.X0 = d.r.X0; p.X1 = d.r.X1; p.Y0 = Y0; p.Y1 = Y1;
    p.in_place = (p.X0 == 0 && static_cast<int64_t>(p.X1) == canvas_w) ? 1 : 0;
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
