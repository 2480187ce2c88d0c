// ist_compile.cpp — turns a recorded Canvas op list into the device tables of one fused launch.
//
// Reference anchor: what the WeChat Canvas does between createOffscreenCanvas (utils/canvas.js:131-150) and the
// export (utils/canvas.js:205-242) for the calls onStitch issues: fillRect (pages/index/index.js:1424) and
// drawImage under a CTM (utils/canvas.js:153-202).  Instead of rasterising call by call (one full-canvas white
// pass + one pass per image = the output written twice), the op list is decomposed into CELLS — canvas rectangles
// over which the paint stack is constant — so that one launch writes every output pixel exactly once.
#include <algorithm>
#include <utility>
#include <string>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ist_internal.h"
#include "ist_launch.h"

namespace ist {

static inline uint32_t pack_rgba(const uint8_t c[4]) {
  return static_cast<uint32_t>(c[0]) | (static_cast<uint32_t>(c[1]) << 8) | (static_cast<uint32_t>(c[2]) << 16) |
         (static_cast<uint32_t>(c[3]) << 24);
}

// Shared arithmetic contract (DESIGN.md "raster contract"; the oracle implements the same formulas independently).
int resolve_op(const ist_op& op, int64_t canvas_w, int64_t canvas_h, int img_w, int img_h, DevOp* out, bool aa) {
  const double a = op.m[0], b = op.m[1], c = op.m[2], d = op.m[3], e = op.m[4], f = op.m[5];
  const bool straight = (b == 0.0 && c == 0.0 && a != 0.0 && d != 0.0);
  const bool turned = (a == 0.0 && d == 0.0 && b != 0.0 && c != 0.0);
  std::memset(out, 0, sizeof(*out));
  out->image = -1;
  // every later step (clip boxes, clamp boxes, the device's tap arithmetic) assumes finite numbers
  for (int i = 0; i < 6; ++i) if (!std::isfinite(op.m[i])) return fail(IST_E_INVALID, "op transform is not finite");
  for (int i = 0; i < 4; ++i) if (!std::isfinite(op.d[i]) || (op.kind == IST_OP_DRAW && !std::isfinite(op.s[i]))) return fail(IST_E_INVALID, "op rectangle is not finite");
  auto clip_box = [&](double xl, double xh, double yl, double yh, bool edge_aa) {
    // pixel-centre rule, or (edge AA) every pixel the rectangle touches + the sub-box it covers completely
    double X0 = std::ceil(xl - 0.5), X1 = std::ceil(xh - 0.5), Y0 = std::ceil(yl - 0.5), Y1 = std::ceil(yh - 0.5);
    double I0 = X0, I1 = X1, J0 = Y0, J1 = Y1;
    if (edge_aa) {
      X0 = std::floor(xl); X1 = std::ceil(xh); Y0 = std::floor(yl); Y1 = std::ceil(yh);
      I0 = std::ceil(xl); I1 = std::floor(xh); J0 = std::ceil(yl); J1 = std::floor(yh);
    }
    auto cx = [&](double v) { return static_cast<int32_t>(std::min(std::max(v, 0.0), static_cast<double>(canvas_w))); };
    auto cy = [&](double v) { return static_cast<int32_t>(std::min(std::max(v, 0.0), static_cast<double>(canvas_h))); };
    out->X0 = cx(X0); out->X1 = cx(X1); out->Y0 = cy(Y0); out->Y1 = cy(Y1);
    out->IX0 = cx(I0); out->IX1 = std::max(cx(I1), out->IX0); out->IY0 = cy(J0); out->IY1 = std::max(cy(J1), out->IY0);
    out->xl = xl; out->xh = xh; out->yl = yl; out->yh = yh;
    return (out->X1 > out->X0 && out->Y1 > out->Y0);
  };
  if (op.kind == IST_OP_FILL || op.kind == IST_OP_HOLE) {
    if (!straight) return fail(IST_E_UNSUPPORTED, "fillRect under a rotated transform is outside the stitch path");
    if (op.kind == IST_OP_FILL && op.rgba[3] != 255) return fail(IST_E_UNSUPPORTED, "translucent fillStyle is outside the stitch path");
    if (!(op.d[2] > 0.0) || !(op.d[3] > 0.0)) return 1;
    const double xa = a * op.d[0] + e, xb = a * (op.d[0] + op.d[2]) + e;
    const double ya = d * op.d[1] + f, yb = d * (op.d[1] + op.d[3]) + f;
    out->flags = (op.kind == IST_OP_HOLE ? OPF_HOLE : OPF_FILL) | OPF_OPAQUE;
    out->rgba = pack_rgba(op.rgba);
    return clip_box(std::min(xa, xb), std::max(xa, xb), std::min(ya, yb), std::max(ya, yb), false) ? 0 : 1;   // fills keep the pixel-centre rule
  }
  if (op.kind != IST_OP_DRAW) return fail(IST_E_INVALID, "unknown op kind");
  if (!straight && !turned) return fail(IST_E_UNSUPPORTED, "drawImage under a non axis-aligned transform is outside the stitch path");
  const double sx = op.s[0], sy = op.s[1], sw = op.s[2], sh = op.s[3];
  const double rx = op.d[0], ry = op.d[1], rw = op.d[2], rh = op.d[3];
  if (!(rw > 0.0) || !(rh > 0.0) || !(sw > 0.0) || !(sh > 0.0)) return 1;     // Canvas draws nothing
  // u (user x) is driven by canvas X (straight) or canvas Y (turned); v likewise
  const double ku = turned ? b : a, eu = turned ? f : e;
  const double kv = turned ? c : d, ev = turned ? e : f;
  const double gx = sw / rw, gy = sh / rh;
  out->kx = gx / ku;
  out->ox = sx - (eu / ku + rx) * gx;
  out->ky = gy / kv;
  out->oy = sy - (ev / kv + ry) * gy;
  if (!std::isfinite(out->kx) || !std::isfinite(out->ky) || !std::isfinite(out->ox) || !std::isfinite(out->oy) || out->kx == 0.0 || out->ky == 0.0)
    return fail(IST_E_INVALID, "op scale overflows");
  const double wa = ku * rx + eu, wb = ku * (rx + rw) + eu;
  const double za = kv * ry + ev, zb = kv * (ry + rh) + ev;
  const double wl = std::min(wa, wb), wh = std::max(wa, wb), zl = std::min(za, zb), zh = std::max(za, zb);
  const bool any = turned ? clip_box(zl, zh, wl, wh, aa) : clip_box(wl, wh, zl, zh, aa);
  // clamp box = the source rectangle's pixels that exist in the bitmap (empty: the rectangle lies outside it)
  double t;
  t = std::floor(sx);            if (t > img_w - 1) return 1;  out->cx0 = t < 0.0 ? 0 : static_cast<int32_t>(t);
  t = std::ceil(sx + sw) - 1.0;  if (t < 0.0) return 1;        out->cx1 = t > img_w - 1 ? img_w - 1 : static_cast<int32_t>(t);
  t = std::floor(sy);            if (t > img_h - 1) return 1;  out->cy0 = t < 0.0 ? 0 : static_cast<int32_t>(t);
  t = std::ceil(sy + sh) - 1.0;  if (t < 0.0) return 1;        out->cy1 = t > img_h - 1 ? img_h - 1 : static_cast<int32_t>(t);
  if (out->cx1 < out->cx0 || out->cy1 < out->cy0) return 1;
  out->image = op.image;
  out->flags = turned ? OPF_SWAP : 0;
  const bool small_off = std::fabs(out->ox) < 4.0e9 && std::fabs(out->oy) < 4.0e9;      // the fast paths hold offsets as integers
  if (!turned && small_off && std::fabs(out->kx) == 1.0 && std::fabs(out->ky) == 1.0 && out->ox == std::floor(out->ox) && out->oy == std::floor(out->oy)) {
    out->flags |= OPF_IDENTITY;
    if (out->kx < 0.0) out->flags |= OPF_FLIPX;
    if (out->ky < 0.0) out->flags |= OPF_FLIPY;
  }
  if (turned && small_off && std::fabs(out->kx) == 1.0 && std::fabs(out->ky) == 1.0 && out->ox == std::floor(out->ox) && out->oy == std::floor(out->oy))
    out->flags |= OPF_UNIT_SWAP;
  return any ? 0 : 1;
}

// number of distinct source indices one axis of a draw touches over canvas coordinates [lo, hi)
static int64_t distinct_taps(double k, double o, int lo, int hi, int clo, int chi, int filter) {
  if (hi <= lo) return 0;
  if (filter == IST_FILTER_AREA && std::fabs(k) > 1.0) {     // a box of width |k| per canvas pixel: contiguous boxes tile the whole span
    const double a = k * static_cast<double>(lo) + o, b = k * static_cast<double>(hi) + o;
    const double s0 = std::min(a, b), s1 = std::max(a, b);
    const int64_t i0 = static_cast<int64_t>(std::min(std::max(std::floor(s0), -4.0e9), 4.0e9)), i1 = static_cast<int64_t>(std::min(std::max(std::ceil(s1), -4.0e9), 4.0e9)) - 1;
    const int64_t c0 = std::min<int64_t>(std::max<int64_t>(i0, clo), chi), c1 = std::min<int64_t>(std::max<int64_t>(i1, clo), chi);
    return c1 - c0 + 1;
  }
  if (filter == IST_FILTER_AREA) filter = IST_FILTER_BILINEAR;
  // the taps are monotonic in the canvas coordinate; clamping keeps them so
  auto first_tap = [&](int w) -> int64_t {
    const double s = k * (static_cast<double>(w) + 0.5) + o;
    const double fl = std::min(std::max(std::floor(filter == IST_FILTER_BILINEAR ? s - 0.5 : s), -4.0e9), 4.0e9);
    return static_cast<int64_t>(fl);
  };
  auto clampi = [&](int64_t v) { return std::min<int64_t>(std::max<int64_t>(v, clo), chi); };
  const int64_t second = filter == IST_FILTER_BILINEAR ? 1 : 0;
  if (std::fabs(k) <= 1.0) {
    // neighbouring canvas coordinates are at most one source index apart: every index between the ends is touched
    const int64_t a = first_tap(lo), b = first_tap(hi - 1);
    return clampi(std::max(a, b) + second) - clampi(std::min(a, b)) + 1;
  }
  // a shrink skips indices: walk the coordinates in the direction the taps grow and count the new ones
  int64_t count = 0, last = INT64_MIN;
  for (int n = 0; n < hi - lo; ++n) {
    const int64_t i0 = first_tap(k > 0.0 ? lo + n : hi - 1 - n);
    for (int64_t v = i0; v <= i0 + second; ++v) {
      const int64_t c = clampi(v);
      if (c > last) { ++count; last = c; }
    }
  }
  return count;
}

// Shape of a compiled job.  Production compiles use the constants below (the measured optima on MI355X) and touch
// neither the environment nor any mutable global, so jobs can be compiled concurrently on any number of threads.  A
// process started with IST_TUNING=1 (tools/sweep_*.py, tools/exp_*.py: single-threaded benchmarks) re-reads the knobs at
// every compile: IST_COPY_TILE=WxH (W = 256, 512, ...), IST_LDS_BUDGET (bytes), IST_LDS_RUN, IST_LDS_TILE_W, IST_NO_LDS, IST_NO_BANDS, IST_NO_SORT,
// IST_NO_TILE_TABLE.
struct CompileKnobs {
  int tile_w = 256, tile_h = 8;        // FILL / COPY: ~64 KB of loads in flight per CU
  int lds_run = 2;                     // pipeline stages per workgroup on the SAMPLE_LDS path
  int lds_tile_h = 0;                  // IST_LDS_TILE_H: stages per workgroup = this / rows per stage (instead of lds_run)
  int lds_tile_w = 0;                  // 0: pick 256 / 128 / 64 per cell; IST_LDS_TILE_W pins one width
  int64_t lds_budget_words = 6144;     // 24 KiB footprint budget per workgroup
  bool no_lds = false, no_bands = false, no_tile_table = false, no_sort = false;
  // SAMPLE_STREAM (measured on MI355X, tools/exp_paths.py): a ring of 2 row pairs per wave beats 3 and 4 (LDS per
  // workgroup decides the workgroups per CU), tiles 8 rows tall beat 4 / 16 / 32, 256-pixel-wide tiles beat narrower ones
  // wherever they fit 48 KiB, and the path beats the staged one from |ky| = 2 up (9 x 12 MP onto the iOS canvas, k = 2.2:
  // 86 us against 94 us; k = 4: 47 against 74; the Android canvas, k = 6.65: 26 against 39).  Below 2 the staged path
  // re-uses the source rows that neighbouring output rows share and wins (k = 1.33: 119 us against 126-140 us).
  int stream = 2;                      // ring depth (0: never stream)
  int stream_h = 8;                    // tile height
  double stream_min_k = 2.0;           // stream for |ky| >= this
  int64_t stream_cap = 12288;          // LDS words per workgroup that decide the tile width
  int area_tile_h = 0;                 // AREA_STREAM: 0 = by the box height (see the cell classification); IST_AREA_TILE_H pins it
  int xcd_rotate = 0;                  // tile table: rotate every tile row of a band so that tile column tc sits at a launch index = tc mod 8 (IST_XCD_ROTATE, see the table builder)
  int area_passes = 1;                 // AREA_STREAM: most 64-lane passes of a tile's x footprint (IST_AREA_PASSES).  Measured (tools/sweep_area.py,
                                       // 9 x 12 MP): one pass per tile 82 us on the Android plan against 164 us with two; iOS 125 / 124, 4x 86 / 84
};
bool tuning_mode() {
  static const bool on = [] { const char* e = std::getenv("IST_TUNING"); return e && *e && std::strcmp(e, "0") != 0; }();
  return on;
}
namespace {
struct Timeline { bool on = false; std::chrono::steady_clock::time_point t0; std::vector<std::pair<double, std::string>> marks; };
Timeline& tl() { thread_local Timeline t; return t; }
bool tl_enabled() { static const bool on = tuning_mode() && std::getenv("IST_TIMELINE") != nullptr; return on; }
}  // namespace
void tl_begin() { if (!tl_enabled()) return; Timeline& t = tl(); t.on = true; t.marks.clear(); t.t0 = std::chrono::steady_clock::now(); }
bool tl_active() { return tl_enabled() && tl().on; }
void tl_mark(const char* what, long a) {
  if (!tl_enabled()) return;
  Timeline& t = tl();
  if (!t.on) return;
  t.marks.emplace_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t.t0).count(), a >= 0 ? std::string(what) + " " + std::to_string(a) : std::string(what));
}
void tl_end(const char* what) {
  if (!tl_enabled()) return;
  Timeline& t = tl();
  if (!t.on) return;
  t.on = false;
  const double total = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t.t0).count();
  static const double slow_ms = std::getenv("IST_TIMELINE_SLOW_MS") ? std::atof(std::getenv("IST_TIMELINE_SLOW_MS")) : 0.0;
  if (total < slow_ms * 1e3) return;
  std::fprintf(stderr, "[ist timeline] ---- %s: %.1f us\n", what, total);
  double prev = 0.0;
  for (const auto& m : t.marks) { std::fprintf(stderr, "[ist timeline] %9.1f us  (+%8.1f)  %s\n", m.first, m.first - prev, m.second.c_str()); prev = m.first; }
}
static CompileKnobs read_knobs() {
  CompileKnobs k;
  if (!tuning_mode()) return k;
  int w = 0, h = 0;
  const char* e = std::getenv("IST_COPY_TILE");
  if (e && std::sscanf(e, "%dx%d", &w, &h) == 2 && w >= 256 && (w & (w - 1)) == 0 && h >= 1 && h <= 4096) { k.tile_w = w; k.tile_h = h; }
  if ((e = std::getenv("IST_LDS_BUDGET")) != nullptr) k.lds_budget_words = std::max<int64_t>(256, std::atoll(e) / 4);
  if ((e = std::getenv("IST_LDS_RUN")) != nullptr) k.lds_run = std::min(16, std::max(1, std::atoi(e)));
  if ((e = std::getenv("IST_LDS_TILE_H")) != nullptr) k.lds_tile_h = std::min(64, std::max(0, std::atoi(e)));
  if ((e = std::getenv("IST_LDS_TILE_W")) != nullptr) { const int v = std::atoi(e); if (v == 64 || v == 128 || v == 256) k.lds_tile_w = v; }
  if ((e = std::getenv("IST_STREAM")) != nullptr) k.stream = std::min(8, std::max(0, std::atoi(e)));
  if ((e = std::getenv("IST_STREAM_MIN_K")) != nullptr) k.stream_min_k = std::atof(e);
  if ((e = std::getenv("IST_AREA_TILE_H")) != nullptr) k.area_tile_h = std::min(64, std::max(0, std::atoi(e)));
  if ((e = std::getenv("IST_AREA_PASSES")) != nullptr) k.area_passes = std::min(3, std::max(1, std::atoi(e)));
  if ((e = std::getenv("IST_STREAM_CAP")) != nullptr) k.stream_cap = std::max<int64_t>(1024, std::atoll(e) / 4);
  if ((e = std::getenv("IST_STREAM_H")) != nullptr) k.stream_h = std::min(64, std::max(4, std::atoi(e) & ~3));
  if ((e = std::getenv("IST_XCD_ROTATE")) != nullptr) k.xcd_rotate = std::atoi(e);
  k.no_lds = std::getenv("IST_NO_LDS") != nullptr;
  k.no_bands = std::getenv("IST_NO_BANDS") != nullptr;
  k.no_tile_table = std::getenv("IST_NO_TILE_TABLE") != nullptr;
  k.no_sort = std::getenv("IST_NO_SORT") != nullptr;
  return k;
}

int compile_ops(int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4], const ist_op* ops, int n_ops,
                const ist_image_desc* images, int n_images, int filter, const ist_region* clip, Compiled* out) {
  if (canvas_w < 1 || canvas_h < 1 || canvas_w > (1 << 29) || canvas_h > 2147483647LL)
    return fail(IST_E_OUTPUT_SIZE, "输出尺寸计算失败: canvas size out of range");
  if (n_ops < 0 || (n_ops > 0 && !ops)) return fail(IST_E_INVALID, "compile_ops: bad op list");
  const bool aa = (filter & IST_FILTER_EDGE_AA) != 0;
  filter &= 0xFF;
  if (filter != IST_FILTER_NEAREST && filter != IST_FILTER_BILINEAR && filter != IST_FILTER_AREA) return fail(IST_E_INVALID, "unknown filter");
  // AREA: minifying draws are averaged per pixel on the general path; everything else about the job is bilinear
  const bool area = filter == IST_FILTER_AREA;
  const int job_filter = filter;
  if (area) filter = IST_FILTER_BILINEAR;
  const CompileKnobs knobs = read_knobs();
  out->canvas_w = canvas_w; out->canvas_h = canvas_h; out->filter = job_filter | (aa ? IST_FILTER_EDGE_AA : 0);
  out->ops.clear(); out->cells.clear(); out->stacks.clear(); out->bands.clear(); out->tiles.clear();
  out->lds_words = 0; out->lds_half = 0; out->kernel_kind = 0;
  out->img_w.assign(static_cast<size_t>(n_images), 0);
  out->img_h.assign(static_cast<size_t>(n_images), 0);
  for (int i = 0; i < n_images; ++i) {
    out->img_w[i] = images[i].bmp_width > 0 ? images[i].bmp_width : images[i].width;
    out->img_h[i] = images[i].bmp_height > 0 ? images[i].bmp_height : images[i].height;
  }

  // region to render
  int64_t RX0 = 0, RY0 = 0, RX1 = canvas_w, RY1 = canvas_h;
  if (clip) {
    RX0 = std::max<int64_t>(0, clip->x); RY0 = std::max<int64_t>(0, clip->y);
    RX1 = std::min<int64_t>(canvas_w, static_cast<int64_t>(clip->x) + clip->w);
    RY1 = std::min<int64_t>(canvas_h, static_cast<int64_t>(clip->y) + clip->h);
    if (RX1 <= RX0 || RY1 <= RY0) return fail(IST_E_INVALID, "clip region is empty");
  }
  out->rx0 = RX0; out->ry0 = RY0; out->rx1 = RX1; out->ry1 = RY1;

  // 1. resolve every op into canvas space (ops that draw nothing are dropped)
  for (int i = 0; i < n_ops; ++i) {
    int iw = 0, ih = 0;
    if (ops[i].kind == IST_OP_DRAW) {
      if (ops[i].image < 0 || ops[i].image >= n_images) return fail(IST_E_INVALID, "op refers to a missing image");
      iw = out->img_w[ops[i].image]; ih = out->img_h[ops[i].image];
      if (iw < 1 || ih < 1) return fail(IST_E_DECODE, "图片" + std::to_string(ops[i].image) + "解码异常");
    }
    DevOp r;
    const int rc = resolve_op(ops[i], canvas_w, canvas_h, iw, ih, &r, aa);
    if (rc < 0) return rc;
    if (rc > 0) continue;
    if (ops[i].kind == IST_OP_DRAW && images[ops[i].image].opaque) r.flags |= OPF_OPAQUE;
    // clip to the render region
    r.X0 = static_cast<int32_t>(std::max<int64_t>(r.X0, RX0)); r.X1 = static_cast<int32_t>(std::min<int64_t>(r.X1, RX1));
    r.Y0 = static_cast<int32_t>(std::max<int64_t>(r.Y0, RY0)); r.Y1 = static_cast<int32_t>(std::min<int64_t>(r.Y1, RY1));
    if (r.X1 <= r.X0 || r.Y1 <= r.Y0) continue;
    r.IX0 = std::min(std::max(r.IX0, r.X0), r.X1); r.IX1 = std::min(std::max(r.IX1, r.IX0), r.X1);
    r.IY0 = std::min(std::max(r.IY0, r.Y0), r.Y1); r.IY1 = std::min(std::max(r.IY1, r.IY0), r.Y1);
    out->ops.push_back(r);
  }
  const int n = static_cast<int>(out->ops.size());

  // 2. grid of break lines
  std::vector<int32_t> xs{static_cast<int32_t>(RX0), static_cast<int32_t>(RX1)}, ys{static_cast<int32_t>(RY0), static_cast<int32_t>(RY1)};
  for (const DevOp& r : out->ops) {
    xs.push_back(r.X0); xs.push_back(r.X1); ys.push_back(r.Y0); ys.push_back(r.Y1);
    xs.push_back(r.IX0); xs.push_back(r.IX1); ys.push_back(r.IY0); ys.push_back(r.IY1);     // = the outer edges unless edge AA
  }
  std::sort(xs.begin(), xs.end()); xs.erase(std::unique(xs.begin(), xs.end()), xs.end());
  std::sort(ys.begin(), ys.end()); ys.erase(std::unique(ys.begin(), ys.end()), ys.end());

  // canvas clear colour: premultiplied (general path) and as it reads back (fill path)
  uint8_t pm[4], back[4];
  for (int c = 0; c < 3; ++c) pm[c] = static_cast<uint8_t>((clear_rgba[c] * clear_rgba[3] + 127) / 255);
  pm[3] = clear_rgba[3];
  for (int c = 0; c < 3; ++c) {
    if (pm[3] == 255) back[c] = pm[c];
    else if (pm[3] == 0) back[c] = 0;
    else back[c] = static_cast<uint8_t>(std::min(255u, (pm[c] * 255u + pm[3] / 2u) / pm[3]));
  }
  back[3] = pm[3];
  const uint32_t clear_pm = pack_rgba(pm), clear_back = pack_rgba(back);

  // 3. cells: per grid row, merge horizontally adjacent grid cells whose paint stack is identical
  ist_job_info& info = out->info;
  std::memset(&info, 0, sizeof(info));
  int64_t tiles = 0;
  std::vector<int32_t> stack, prev_stack;
  for (size_t yi = 0; yi + 1 < ys.size(); ++yi) {
    const int32_t Y0 = ys[yi], Y1 = ys[yi + 1];
    bool have_prev = false;
    for (size_t xi = 0; xi + 1 < xs.size(); ++xi) {
      const int32_t X0 = xs[xi], X1 = xs[xi + 1];
      stack.clear();
      for (int k = 0; k < n; ++k) {
        const DevOp& r = out->ops[k];
        if (r.X0 <= X0 && X1 <= r.X1 && r.Y0 <= Y0 && Y1 <= r.Y1) {
          const bool full = r.IX0 <= X0 && X1 <= r.IX1 && r.IY0 <= Y0 && Y1 <= r.IY1;     // false only on fractional edge strips
          if ((r.flags & OPF_OPAQUE) && full) stack.clear();
          stack.push_back(full ? k : ~k);          // ~k marks a partially covering op: the cell goes through the general path
        }
      }
      if (have_prev && stack == prev_stack) {       // extend the previous cell to the right
        DevCell& pc = out->cells.back();
        // an identity cell may only grow while the whole span stays inside the source (no clamping)
        pc.X1 = X1;
        continue;
      }
      if (!stack.empty() && stack[0] >= 0 && (out->ops[stack[0]].flags & OPF_HOLE)) {
        if (stack.size() > 1) return fail(IST_E_UNSUPPORTED, "drawing over a region reserved for another producer");
        have_prev = false;          // nothing is written here
        continue;
      }
      DevCell cell;
      std::memset(&cell, 0, sizeof(cell));
      cell.X0 = X0; cell.Y0 = Y0; cell.X1 = X1; cell.Y1 = Y1;
      cell.stack_off = static_cast<int32_t>(out->stacks.size());
      // the colour under the stack
      uint32_t bg_pm = clear_pm, bg_back = clear_back;
      size_t first = 0;
      if (!stack.empty() && stack[0] >= 0 && (out->ops[stack[0]].flags & OPF_FILL)) { bg_pm = bg_back = out->ops[stack[0]].rgba; first = 1; }
      bool partial = false;
      for (size_t k = first; k < stack.size(); ++k) { partial |= stack[k] < 0; out->stacks.push_back(stack[k] < 0 ? ~stack[k] : stack[k]); }
      cell.stack_len = static_cast<int32_t>(stack.size() - first);
      cell.op = cell.stack_len ? out->stacks[cell.stack_off] : -1;
      cell.tiles_x = partial ? -1 : 0;            // temporary mark, consumed by the classification below
      cell.bg = bg_pm;
      out->cells.push_back(cell);
      prev_stack = stack; have_prev = true;
    }
  }

  // 4. classify + tile each cell
  bool has_area = false, has_general = false;
  for (DevCell& cell : out->cells) {
    const bool bg_opaque = (cell.bg >> 24) == 255u;
    const bool partial = cell.tiles_x < 0;
    cell.tiles_x = 0;
    bool minified = false;                        // IST_FILTER_AREA: a draw of the stack shrinks on some axis
    if (area)
      for (int k = 0; k < cell.stack_len; ++k) {
        const DevOp& r = out->ops[out->stacks[cell.stack_off + k]];
        minified |= !(r.flags & OPF_FILL) && (std::fabs(r.kx) > 1.0 || std::fabs(r.ky) > 1.0);
      }
    if (partial || minified) {
      cell.path = PATH_GENERAL;                   // fractional edge strip: per-pixel coverage; box-averaged draw: per-pixel footprint
      // ONE axis-aligned shrinking draw over an opaque colour (or an opaque draw): the streamed box filter (tile_area_stream).
      // LDS: 4 waves x one row of float4 column sums over the tile's x footprint; the widest tile that fits 48 KiB.
      if (!partial && cell.stack_len == 1 && !knobs.no_lds) {
        const DevOp& r = out->ops[cell.op];
        if (!(r.flags & (OPF_SWAP | OPF_FILL | OPF_HOLE)) && (bg_opaque || (r.flags & OPF_OPAQUE)) && r.cx1 >= r.cx0 && r.cy1 >= r.cy0 &&
            std::fabs(r.ky) <= 64.0) {
          // Tile width: the row sums (most of the work) run in 64-lane passes of 4 source pixels each, so the x footprint of a
          // tile should fill its pass: the widest tile (<= 128 canvas pixels: 2 per lane) whose footprint fits ONE pass of 256
          // source pixels (two passes: a tuning knob; never faster, 2x slower at 6.6x).  LDS: 4 waves x 256 px x float4 = 16 KiB.
          const double akx = std::fabs(r.kx), bwx = std::max(akx, 1.0);
          auto foot_px = [&](int w) { return (static_cast<int64_t>(std::ceil((w - 1) * akx + bwx)) + 2 + 3) & ~3LL; };   // >= the kernel's
          int tw = 0; int64_t wl = 0; double best = 0.0;
          for (int m = 1; m <= knobs.area_passes; ++m) {
            const double room = 256.0 * m - 3.0 - bwx;        // ceil(span) + 2 <= 256 m  with  span = (w - 1) |kx| + box
            if (room < 0.0) continue;
            const int w = static_cast<int>(std::min(128.0, std::floor(room / std::max(akx, 1e-9)) + 1.0));
            if (w < 24) continue;
            const int64_t px = foot_px(w);
            const double score = static_cast<double>(w) / static_cast<double>((px / 4 + 63) / 64);
            if (score > best - 1e-9) { best = score; tw = w; wl = px; }
          }
          if (!tw && akx <= 200.0)                            // a very strong shrink: narrow tiles
            for (int w = 16; w >= 1 && !tw; w >>= 1)
              if (foot_px(w) <= 768) { tw = w; wl = foot_px(w); }
          if (tw) {
            // tile height: a wave owns every fourth row of the tile and reads ceil(|ky|) + 1 source rows per output row, one
            // dependent round of loads per 4 of them; tall boxes get short tiles (more workgroups, one row per wave)
            const int box_rows = static_cast<int>(std::ceil(std::max(std::fabs(r.ky), 1.0))) + 1;
            const int th = knobs.area_tile_h ? knobs.area_tile_h : (box_rows <= 4 ? 32 : box_rows <= 6 ? 8 : 4);     // (measured, tools/sweep_area.py: 2.2x 120 us at 32 rows, 124 at 16, 131 at 8; 6.6x 81 us at 4 rows, 88-95 at 32)
            cell.path = PATH_AREA_STREAM; cell.tile_w = tw; cell.tile_h = th; cell.sub_h = 0;
            out->lds_words = std::max<int32_t>(out->lds_words, static_cast<int32_t>(4 * 4 * wl));
            if (!bg_opaque) cell.bg = 0xFFFFFFFFu;            // never used: the draw is opaque
          }
        }
      }
    } else if (cell.stack_len == 0) {
      cell.path = PATH_FILL;
      if (!bg_opaque) cell.bg = clear_back;       // reads back un-premultiplied
    } else if (cell.stack_len == 1 && !(out->ops[cell.op].flags & OPF_SWAP) &&
               (bg_opaque || (out->ops[cell.op].flags & OPF_OPAQUE))) {
      const DevOp& r = out->ops[cell.op];
      bool copy = (r.flags & OPF_IDENTITY) != 0;
      if (copy) {   // every source coordinate of the cell must be inside the clamp box
        const int64_t ox = static_cast<int64_t>(r.ox), oy = static_cast<int64_t>(r.oy);
        const int64_t xa = (r.flags & OPF_FLIPX) ? ox - 1 - cell.X0 : cell.X0 + ox, xb = (r.flags & OPF_FLIPX) ? ox - cell.X1 : cell.X1 - 1 + ox;
        const int64_t ya = (r.flags & OPF_FLIPY) ? oy - 1 - cell.Y0 : cell.Y0 + oy, yb = (r.flags & OPF_FLIPY) ? oy - cell.Y1 : cell.Y1 - 1 + oy;
        copy = std::min(xa, xb) >= r.cx0 && std::max(xa, xb) <= r.cx1 && std::min(ya, yb) >= r.cy0 && std::max(ya, yb) <= r.cy1;
      }
      cell.path = copy ? PATH_COPY : PATH_SAMPLE;
      if (!bg_opaque) cell.bg = 0xFFFFFFFFu;      // never used: the draw is opaque
    } else {
      cell.path = PATH_GENERAL;
      // one quarter-turned draw over an opaque colour, bilinear: stage the footprint transposed in LDS
      if (!partial && cell.stack_len == 1 && (out->ops[cell.op].flags & OPF_SWAP) && filter == IST_FILTER_BILINEAR &&
          (bg_opaque || (out->ops[cell.op].flags & OPF_OPAQUE)) && !knobs.no_lds) {
        const DevOp& r = out->ops[cell.op];
        const double akx = std::fabs(r.kx), aky = std::fabs(r.ky);
        if (r.cx1 > r.cx0 && r.cy1 > r.cy0 && akx <= 128.0 && aky <= 128.0) {
          for (int th = 64; th >= 16; th >>= 1) {
            const int64_t fw = static_cast<int64_t>(std::floor((th - 1) * akx)) + 3;     // source columns (driven by canvas Y)
            const int64_t fh = static_cast<int64_t>(std::floor(63.0 * aky)) + 3;         // source rows (driven by canvas X)
            const int64_t need = fw * (fh | 1);
            if (need <= 8192) {                                                          // 32 KiB
              cell.path = PATH_SWAP_LDS; cell.tile_w = 64; cell.tile_h = th;
              out->lds_words = std::max<int32_t>(out->lds_words, static_cast<int32_t>(need));
              if (!bg_opaque) cell.bg = 0xFFFFFFFFu;
              break;
            }
          }
        }
      }
    }
    if (cell.path == PATH_SAMPLE && filter == IST_FILTER_BILINEAR && !knobs.no_lds) {
      // stage the tile's source footprint in LDS when it fits the budget with at least 4 output rows per tile
      const DevOp& r = out->ops[cell.op];
      const double akx = std::fabs(r.kx), aky = std::fabs(r.ky);
      const int64_t budget = knobs.lds_budget_words;
      if (knobs.stream >= 2 && r.cx1 > r.cx0 && r.cy1 > r.cy0 && aky >= knobs.stream_min_k && akx <= 64.0) {
        // a strong shrink: neighbouring output rows share no source row, so every wave streams the row pairs of its own
        // output rows (tile_sample_stream).  LDS = 4 waves x depth x 2 rows of the tile's x footprint; the widest tile
        // that fits 48 KiB (64 KiB for the narrowest)
        for (int tw = 256; tw >= 64; tw >>= 1) {
          const int64_t wl = (static_cast<int64_t>(std::floor((tw - 1) * akx)) + 3 + 3) & ~3LL;    // pixels per LDS row
          const int64_t need = 8 * knobs.stream * wl;
          if (need > knobs.stream_cap && tw > 64) continue;
          if (need > 16384) continue;                 // |kx| above ~16: the direct path
          cell.path = PATH_SAMPLE_STREAM; cell.tile_w = tw; cell.sub_h = knobs.stream; cell.tile_h = knobs.stream_h;
          out->lds_words = std::max<int32_t>(out->lds_words, static_cast<int32_t>(need));
          break;
        }
      }
      if (cell.path == PATH_SAMPLE && r.cx1 > r.cx0 && r.cy1 > r.cy0 && akx <= 4.0 && aky <= 8.0) {
        // tile shape: 256 pixels wide (4 per lane and row) whenever a stage of at least 4 rows fits the budget: measured on
        // MI355X (tools/sweep_resample.py) the wide tile wins even where a 128-wide one would carry more output pixels per
        // footprint (mixed horizontal strip, kx = ky = 1.87: 125 us against 137 us).  Narrower tiles (2 or 1 pixel per lane)
        // only take the scales where the wide one does not fit at all (|kx| from about 2.3 to 4 with |ky| below 2).
        int best_w = 0, best_h = 0; int64_t best_need = 0;
        const int w_lo = knobs.lds_tile_w ? knobs.lds_tile_w : 64, w_hi = knobs.lds_tile_w ? knobs.lds_tile_w : 256;
        for (int tw = w_hi; tw >= w_lo && !best_h; tw >>= 1) {
          const int64_t wl = (static_cast<int64_t>(std::floor((tw - 1) * akx)) + 3 + 3) & ~3LL;    // pixels per LDS row
          for (int t = 32; t >= 4; t -= 4) {
            const int64_t fh = static_cast<int64_t>(std::floor((t - 1) * aky)) + 3;
            if (wl * fh > budget) continue;
            best_w = tw; best_h = t; best_need = wl * fh;
            break;                                  // the tallest stage of this width
          }
        }
        if (best_h) {
          // a workgroup walks `run` stages down its column (see tile_sample_lds)
          const int run = knobs.lds_tile_h ? std::max(1, knobs.lds_tile_h / best_h) : knobs.lds_run;
          cell.path = PATH_SAMPLE_LDS; cell.tile_w = best_w; cell.sub_h = best_h; cell.tile_h = best_h * run;
          out->lds_half = std::max<int32_t>(out->lds_half, static_cast<int32_t>(best_need));
          out->lds_words = std::max<int32_t>(out->lds_words, out->lds_half);
        }
      }
    }
    if (cell.path == PATH_SAMPLE_LDS || cell.path == PATH_SWAP_LDS || cell.path == PATH_SAMPLE_STREAM || cell.path == PATH_AREA_STREAM) {}
    else if (cell.path == PATH_GENERAL) { cell.tile_w = 64; cell.tile_h = 64; }
    else if (cell.path == PATH_SAMPLE) { cell.tile_w = 256; cell.tile_h = 32; }
    else { cell.tile_w = knobs.tile_w; cell.tile_h = knobs.tile_h; }   // FILL / COPY: tile_w = 256 << n
    const int64_t w = cell.X1 - cell.X0, h = cell.Y1 - cell.Y0;
    cell.tiles_x = static_cast<int32_t>((w + cell.tile_w - 1) / cell.tile_w);
    const int64_t tiles_y = (h + cell.tile_h - 1) / cell.tile_h;
    const int64_t nt = cell.tiles_x * tiles_y;
    // bands: consecutive cells with the same rows and tile height are walked canvas-row-major as one unit
    if (!out->bands.empty()) {
      DevBand& b = out->bands.back();
      const DevCell& f = out->cells[b.first_cell];
      if (f.Y0 == cell.Y0 && f.Y1 == cell.Y1 && f.tile_h == cell.tile_h && !knobs.no_bands) {
        cell.band_x = b.tiles_per_row;
        b.tiles_per_row += cell.tiles_x;
        b.n_cells += 1;
      } else {
        out->bands.push_back(DevBand{tiles, static_cast<int32_t>(&cell - out->cells.data()), 1, cell.tiles_x, {0, 0, 0}});
      }
    } else {
      out->bands.push_back(DevBand{tiles, static_cast<int32_t>(&cell - out->cells.data()), 1, cell.tiles_x, {0, 0, 0}});
    }
    cell.tile_begin = tiles;
    tiles += nt;
    info.out_pixels += w * h;
    out->kernel_kind = std::max<int32_t>(out->kernel_kind, (cell.path == PATH_FILL || cell.path == PATH_COPY) ? 0 : (cell.path == PATH_SAMPLE || cell.path == PATH_SAMPLE_LDS || cell.path == PATH_SAMPLE_STREAM) ? 1 : 2);
    if (cell.path == PATH_AREA_STREAM) has_area = true; else if (cell.path == PATH_GENERAL || cell.path == PATH_SWAP_LDS) has_general = true;
    switch (cell.path) {
      case PATH_FILL: info.tiles_fill += nt; break;
      case PATH_COPY: info.tiles_copy += nt; break;
      case PATH_SAMPLE: case PATH_SAMPLE_LDS: case PATH_SAMPLE_STREAM: case PATH_SWAP_LDS: case PATH_AREA_STREAM: info.tiles_sample += nt; break;
      default: info.tiles_general += nt; break;
    }
    for (int k = 0; k < cell.stack_len; ++k) {
      const DevOp& r = out->ops[out->stacks[cell.stack_off + k]];
      if (r.flags & OPF_FILL) continue;
      const bool sw = (r.flags & OPF_SWAP) != 0;
      // source x is driven by canvas X (or Y when turned); source y by the other axis
      const int tf = (area && (std::fabs(r.kx) > 1.0 || std::fabs(r.ky) > 1.0)) ? IST_FILTER_AREA : filter;
      const int64_t nx = distinct_taps(r.kx, r.ox, sw ? cell.Y0 : cell.X0, sw ? cell.Y1 : cell.X1, r.cx0, r.cx1, tf);
      const int64_t ny = distinct_taps(r.ky, r.oy, sw ? cell.X0 : cell.Y0, sw ? cell.X1 : cell.Y1, r.cy0, r.cy1, tf);
      info.src_pixels_touched += nx * ny;
    }
  }
  if (tiles > 2147483647LL) return fail(IST_E_OUTPUT_SIZE, "canvas needs more than 2^31 tiles");
  if (has_area) out->kernel_kind = has_general ? 4 : 3;   // 3: fill / copy / resample / streamed box filter; 4: everything
  // launch order of the bands: the expensive tiles (resampling) first, copies next, fills last, so that the workgroups
  // that finish the launch are the short ones (a strip of mixed scales otherwise ends on whatever its last image needs).
  // Stable: bands of one kind keep their canvas order.
  if (!knobs.no_sort && out->bands.size() > 1) {
    auto weight = [&](const DevBand& b) {
      const int32_t path = out->cells[b.first_cell].path;
      return path == PATH_FILL ? 0 : path == PATH_COPY ? 1 : 2;
    };
    std::stable_sort(out->bands.begin(), out->bands.end(), [&](const DevBand& a, const DevBand& b) { return weight(a) > weight(b); });
    int64_t at = 0;
    for (DevBand& b : out->bands) {
      b.tile_begin = at;
      const DevCell& f = out->cells[b.first_cell];
      const int64_t rows = (f.Y1 - f.Y0 + f.tile_h - 1) / f.tile_h;
      int64_t cell_at = at;
      for (int32_t k = 0; k < b.n_cells; ++k) { DevCell& c = out->cells[b.first_cell + k]; c.tile_begin = cell_at; cell_at += static_cast<int64_t>(c.tiles_x) * rows; }
      at += static_cast<int64_t>(b.tiles_per_row) * rows;
    }
  }
  // the table pays when tiles have a long set-up (resample paths: -3..5 % measured); pure fill/copy jobs keep the
  // prefix search, whose few cache lines stay hot in the scalar cache (a per-tile entry is a cold miss: +3 % measured)
  const bool resamples = info.tiles_sample + info.tiles_general > 0;
  if (resamples && tiles <= kMaxTileTable && !knobs.no_tile_table) {
    out->tiles.reserve(static_cast<size_t>(tiles));
    for (const DevBand& b : out->bands) {
      const DevCell& f = out->cells[b.first_cell];
      const int32_t rows = (f.Y1 - f.Y0 + f.tile_h - 1) / f.tile_h;
      for (int32_t tr = 0; tr < rows; ++tr) {
        const size_t row_at = out->tiles.size();
        for (int32_t k = 0; k < b.n_cells; ++k) {
          const DevCell& c = out->cells[b.first_cell + k];
          for (int32_t tc = 0; tc < c.tiles_x; ++tc)
            out->tiles.push_back(DevTile{b.first_cell + k, c.op, c.X0 + tc * c.tile_w, c.Y0 + tr * c.tile_h});
        }
        // XCD-aware order (experiment, IST_XCD_ROTATE=1): workgroups are dealt to the 8 XCDs round-robin by launch index
        // (MI355X_MICROARCH.md), each XCD with its own L2.  Vertically adjacent SAMPLE_LDS tiles read the same halo rows; with
        // 99 tiles per row the tile under tile i is at index i + 99, on another XCD, so the halo is fetched from HBM twice.
        // Rotating row tr by (its first launch index) mod 8 puts tile column tc at an index = tc (mod 8) in every row - all but
        // the <= 7 wrapped tiles of a row - while the walk stays canvas-row-major.
        if (knobs.xcd_rotate && b.tiles_per_row >= 16) {
          const size_t n_row = out->tiles.size() - row_at;
          const size_t s = row_at % 8;                  // launch index of the row's first tile, mod 8
          if (s) std::rotate(out->tiles.begin() + static_cast<std::ptrdiff_t>(row_at), out->tiles.begin() + static_cast<std::ptrdiff_t>(row_at + n_row - s), out->tiles.end());
        }
      }
    }
  }
  info.canvas_w = canvas_w; info.canvas_h = canvas_h;
  info.n_ops = n; info.n_cells = static_cast<int32_t>(out->cells.size());
  info.n_tiles = tiles;
  info.algorithmic_bytes = 4 * info.src_pixels_touched + 4 * info.out_pixels;
  return IST_OK;
}

// ------------------------------------------------------------------------------------------------ the flat form of a job
static bool whole(double v) { return v == std::floor(v) && std::fabs(v) < 9.0e15; }

// The op list on a canvas kFlatPitch / 4 pixels wide, when that is the same copy (ist_ctx.h, FlatTwin).  A job qualifies when every
// op - fills, draws and holes alike - covers whole canvas rows under the identity transform and every draw takes whole rows of a bitmap
// as wide as the canvas at unit scale: its destination (and source) is then ONE byte range of a dense buffer, which becomes a head
// row, whole rows and a tail row of the wide canvas.  Returns nothing for every other job - and for one whose rows are already a
// multiple of 16 KiB or whose canvas is too small for the row pitch to matter.
std::unique_ptr<FlatTwin> compile_flat_twin(int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4], const ist_op* ops, int n_ops,
                                            const ist_image_desc* images, int n_images, int filter, const Compiled& primary) {
  static const bool off = tuning_mode() && std::getenv("IST_FLAT") && std::atoi(std::getenv("IST_FLAT")) == 0;
  if (off || primary.kernel_kind != 0) return nullptr;
  const int64_t row = canvas_w * 4, P = static_cast<int64_t>(kFlatPitch), vw = P / 4;
  // a clip that takes whole rows (the band of a device group, a band of the file pipeline) is a shorter canvas that starts at the clip's
  // first row: the twin addresses bytes from there (dst_delta), so its rows are aligned like the band buffer itself
  if (primary.rx0 != 0 || primary.rx1 != canvas_w) return nullptr;
  const int64_t ry0 = primary.ry0, ry1 = primary.ry1;
  const int64_t total = (ry1 - ry0) * row;
  if (row % 16384 == 0 || total < 64 * P) return nullptr;      // (rows of 16 KiB multiples are in the stores' best class as they are: 4096 px 0.840, 8192 px 0.845)
  std::vector<ist_op> vops;
  std::vector<ist_image_desc> vimg;
  std::unique_ptr<FlatTwin> t(new FlatTwin);
  t->dst_delta = ry0 * row;
  for (int i = 0; i < n_ops; ++i) {
    const ist_op& o = ops[i];
    if (o.kind != IST_OP_FILL && o.kind != IST_OP_DRAW && o.kind != IST_OP_HOLE) return nullptr;
    if (o.m[0] != 1.0 || o.m[1] != 0.0 || o.m[2] != 0.0 || o.m[3] != 1.0 || o.m[4] != 0.0 || o.m[5] != 0.0) return nullptr;
    if (!whole(o.d[1]) || !whole(o.d[3]) || o.d[0] != 0.0 || o.d[2] != static_cast<double>(canvas_w)) return nullptr;
    const int64_t oy = static_cast<int64_t>(o.d[1]), oh = static_cast<int64_t>(o.d[3]);
    if (oy < 0 || oh < 1 || oy + oh > canvas_h) return nullptr;
    int64_t sy = 0;
    if (o.kind == IST_OP_DRAW) {
      if (o.image < 0 || o.image >= n_images) return nullptr;
      const ist_image_desc& im = images[o.image];
      const int64_t iw = im.bmp_width > 0 ? im.bmp_width : im.width, ih = im.bmp_height > 0 ? im.bmp_height : im.height;
      if (iw != canvas_w || o.s[0] != 0.0 || o.s[2] != static_cast<double>(canvas_w) || !whole(o.s[1]) || o.s[3] != o.d[3]) return nullptr;
      sy = static_cast<int64_t>(o.s[1]);
      if (sy < 0 || sy + oh > ih) return nullptr;
    }
    // the rows of the op inside the rendered region, counted from the region's first row
    const int64_t y_lo = std::max(oy, ry0), y_hi = std::min(oy + oh, ry1);
    if (y_hi <= y_lo) continue;
    const int64_t dy = y_lo - ry0, dh = y_hi - y_lo;
    sy += y_lo - oy;
    int vi = -1;
    if (o.kind == IST_OP_DRAW) {
      if (static_cast<int>(vimg.size()) >= kMaxImages) return nullptr;
      vi = static_cast<int>(vimg.size());
      t->src.push_back(FlatTwin::Src{o.image, sy * row - (dy * row) % P});
    }
    // the byte range [D, D + L) of the region as rectangles of the wide canvas
    const int64_t D = dy * row, Lpx = dh * canvas_w, hx = (D % P) / 4, y0 = D / P;
    int64_t left = Lpx, y = 0;
    auto piece = [&](int64_t x, int64_t yy, int64_t w, int64_t h) {
      ist_op v = o;
      v.image = vi;
      v.s[0] = static_cast<double>(x); v.s[1] = static_cast<double>(yy); v.s[2] = static_cast<double>(w); v.s[3] = static_cast<double>(h);
      v.d[0] = static_cast<double>(x); v.d[1] = static_cast<double>(y0 + yy); v.d[2] = static_cast<double>(w); v.d[3] = static_cast<double>(h);
      vops.push_back(v);
    };
    if (hx) { const int64_t w = std::min(vw - hx, left); piece(hx, 0, w, 1); left -= w; y = 1; }
    if (left >= vw) { piece(0, y, vw, left / vw); y += left / vw; left %= vw; }
    if (left) { piece(0, y, left, 1); ++y; }
    if (vi >= 0) {
      ist_image_desc d = images[o.image];
      d.width = d.bmp_width = static_cast<int32_t>(vw);
      d.height = d.bmp_height = static_cast<int32_t>(y);
      d.orientation = 1;
      vimg.push_back(d);
    }
  }
  // Every op boundary that is no multiple of kFlatPitch costs a head and a tail row of tiles with one row in eight used.  Measured
  // (tools/exp_thin.py, 400 MB strips): images of 1.6 / 4 / 8 / 16 MB each -> flat form -4.6 % / -2 % / 0 / +2.5 % against the row
  // form; strips under ~64 MB run at the launch floor either way.  A large strip of small images keeps the row form.
  if (total >= (64ll << 20) && total / std::max<int64_t>(1, static_cast<int64_t>(vimg.size())) < (8ll << 20)) return nullptr;
  const int64_t vh = (total + P - 1) / P;
  if (total % P) {                                    // the wide canvas's last row ends past the region: never written
    ist_op h;
    std::memset(&h, 0, sizeof(h));
    h.kind = IST_OP_HOLE; h.image = -1;
    h.m[0] = h.m[3] = 1.0;
    h.d[0] = static_cast<double>((total % P) / 4); h.d[1] = static_cast<double>(vh - 1); h.d[2] = static_cast<double>(vw - (total % P) / 4); h.d[3] = 1.0;
    vops.push_back(h);
  }
  const std::string keep_msg = g_last_error;
  const int keep_code = g_last_code;
  ist_image_desc none;
  std::memset(&none, 0, sizeof(none));
  if (compile_ops(vw, vh, clear_rgba, vops.data(), static_cast<int>(vops.size()), vimg.empty() ? &none : vimg.data(), static_cast<int>(vimg.size()),
                  filter, nullptr, &t->host) != IST_OK || t->host.kernel_kind != 0) {
    g_last_error = keep_msg; g_last_code = keep_code;             // the twin is an optimisation: its failure is nobody's error
    return nullptr;
  }
  return t;
}

}  // namespace ist

extern "C" int ist_op_box(const ist_op* op, int64_t canvas_w, int64_t canvas_h, int filter, int32_t box[4]) {
  if (!op || !box) return ist::fail(IST_E_INVALID, "ist_op_box: NULL argument");
  ist::DevOp r;
  // the source clamp box does not matter for the destination box: pretend the bitmap is as large as the source rect
  const int iw = static_cast<int>(std::min(std::max(op->s[0] + op->s[2], 1.0), 2147483647.0));
  const int ih = static_cast<int>(std::min(std::max(op->s[1] + op->s[3], 1.0), 2147483647.0));
  const int rc = ist::resolve_op(*op, canvas_w, canvas_h, iw, ih, &r, (filter & IST_FILTER_EDGE_AA) != 0);
  box[0] = r.X0; box[1] = r.Y0; box[2] = r.X1; box[3] = r.Y1;
  return rc;
}

// The flat form of a job as the kernel will walk it - pure CPU, for tests (tests/test_flat_form.py replays it in numpy against the op list):
// one record per cell of the twin's table.  *n_cells = 0 (and IST_OK) when the job has no flat form.
extern "C" int ist_debug_flat_form(int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4], const ist_op* ops, int n_ops,
                                   const ist_image_desc* images, int n_images, int filter, const ist_region* clip, int64_t* pitch,
                                   int64_t* dst_offset, ist_flat_cell* cells, int max_cells, int* n_cells) {
  if (!n_cells || (max_cells > 0 && !cells)) return ist::fail(IST_E_INVALID, "ist_debug_flat_form: NULL argument");
  *n_cells = 0;
  static const uint8_t transparent[4] = {0, 0, 0, 0};
  ist::Compiled primary;
  int rc = ist::compile_ops(canvas_w, canvas_h, clear_rgba ? clear_rgba : transparent, ops, n_ops, images, n_images, filter, clip, &primary);
  if (rc != IST_OK) return rc;
  const std::unique_ptr<ist::FlatTwin> t = ist::compile_flat_twin(canvas_w, canvas_h, clear_rgba ? clear_rgba : transparent, ops, n_ops, images, n_images, filter, primary);
  if (!t) return IST_OK;
  if (pitch) *pitch = static_cast<int64_t>(ist::kFlatPitch);
  if (dst_offset) *dst_offset = t->dst_delta;
  const int64_t P = static_cast<int64_t>(ist::kFlatPitch);
  int n = 0;
  for (const ist::DevCell& c : t->host.cells) {
    if (n < max_cells) {
      ist_flat_cell& o = cells[n];
      std::memset(&o, 0, sizeof(o));
      o.path = c.path; o.image = -1; o.X0 = c.X0; o.Y0 = c.Y0; o.X1 = c.X1; o.Y1 = c.Y1; o.bg = c.bg;
      if (c.path == ist::PATH_COPY) {
        const ist::DevOp& d = t->host.ops[static_cast<size_t>(c.op)];
        const ist::FlatTwin::Src& v = t->src[static_cast<size_t>(d.image)];
        o.image = v.image;
        o.opaque = (d.flags & ist::OPF_OPAQUE) ? 1 : 0;
        o.src_offset = v.delta + (static_cast<int64_t>(c.Y0) + static_cast<int64_t>(d.oy)) * P + (static_cast<int64_t>(c.X0) + static_cast<int64_t>(d.ox)) * 4;
      }
    }
    ++n;
  }
  *n_cells = n;
  return n > max_cells && max_cells > 0 ? ist::fail(IST_E_INVALID, "ist_debug_flat_form: more cells than max_cells") : IST_OK;
}
