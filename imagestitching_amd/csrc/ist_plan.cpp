// ist_plan.cpp — host-side planner of the stitch path (pure CPU, no HIP).
//
// Mirrors, operation for operation in IEEE double, the reference page controller
// (miniprogram-stitch/miniprogram/pages/index/index.js):
//   bigTask                     :1211-1216
//   output size per mode        :1251-1321
//   device caps -> scaleDown    :1323-1357
//   superSample                 :1360-1386
//   rect / cursor loop          :1432-1433, 1522-1554
// and the Canvas call sequence of drawWithOrientation (utils/canvas.js:153-202) as an op list.
// Must be built with -ffp-contract=off: JavaScript never fuses a*b+c.
#include "ist_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ist {

thread_local std::string g_last_error;
thread_local int g_last_code = 0;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  g_last_code = code;
  return code;
}

// Math.round(): half-way cases go toward +infinity; x+0.5 is NOT used because it double-rounds near .5
static inline double js_round(double v) {
  const double lo = std::floor(v);
  return (v - lo >= 0.5) ? lo + 1.0 : lo;
}

// `a || b` on JS numbers: 0 and NaN are falsy
static inline double js_or(double a, double b) { return (a != 0.0 && a == a) ? a : b; }

struct Size { double w, h; };

// Matrix helpers for the Canvas CTM. X = a*u + c*v + e, Y = b*u + d*v + f.
void Ctm::translate(double x, double y) {
  e = a * x + c * y + e;
  f = b * x + d * y + f;
}
void Ctm::scale(double x, double y) {
  a *= x; b *= x; c *= y; d *= y;
}
void Ctm::rotate(double rad) {
  // Quarter turns (EXIF, utils/canvas.js:168,178,184,189,195) use exact cos/sin so the map stays axis aligned;
  // Math.PI*0.5 is not exactly pi/2 and cos() of it is 6e-17, which no raster would honour.
  const double q = rad / 1.5707963267948966;
  const double qr = js_round(q);
  double co, si;
  if (std::fabs(q - qr) < 1e-9) {
    const int k = static_cast<int>(std::fmod(std::fmod(qr, 4.0) + 4.0, 4.0));
    static const double C[4] = {1.0, 0.0, -1.0, 0.0};
    static const double S[4] = {0.0, 1.0, 0.0, -1.0};
    co = C[k]; si = S[k];
  } else {
    co = std::cos(rad); si = std::sin(rad);
  }
  const Ctm o = *this;
  a = o.a * co + o.c * si;
  b = o.b * co + o.d * si;
  c = o.c * co - o.a * si;
  d = o.d * co - o.b * si;
}

}  // namespace ist

using namespace ist;

extern "C" {

int ist_abi_version(void) { return IST_ABI_VERSION; }

const char* ist_last_error(void) { return g_last_error.c_str(); }

void ist_limits_default(int platform, ist_limits* out) {
  // index.js:128-138 — what onLoad stores when the 'canvasLimit' storage entry is absent
  const double side = (platform == IST_PLATFORM_IOS) ? 12288.0 : 4096.0;
  const double cap = side * std::min(side, platform == IST_PLATFORM_ANDROID ? 4096.0 : 12288.0);
  out->platform = platform;
  out->reserved = 0;
  out->max_side = side;
  out->max_pixels = std::max(cap, 4096.0 * 2048.0);
  out->max_super_sample = 0.0;
}

void ist_limits_unlimited(ist_limits* out) {
  // the reference's own way of lifting the caps is the storage entry canvasLimit={size,pixels} (index.js:141-153);
  // these are the values SURVEY.md section 8c used for the "lifted" goldens, with superSample pinned to 1
  out->platform = IST_PLATFORM_OTHER;
  out->reserved = 0;
  out->max_side = 1048576.0;
  out->max_pixels = 1099511627776.0;
  out->max_super_sample = 1.0;
}

int ist_plan_compute(const ist_image_desc* images, int n_images, int direction, int mode, double gap,
                     const ist_limits* limits, ist_plan* out) {
  if (!out) return fail(IST_E_INVALID, "ist_plan_compute: out is NULL");
  std::memset(out, 0, sizeof(*out));
  if (n_images <= 0) return IST_NOTHING_TO_DO;                       // index.js:1189
  if (!images || !limits) return fail(IST_E_INVALID, "ist_plan_compute: NULL argument");
  if (direction != IST_VERTICAL && direction != IST_HORIZONTAL) return fail(IST_E_INVALID, "direction must be vertical or horizontal");
  if (mode < IST_MODE_MIN || mode > IST_MODE_ORIGINAL) mode = IST_MODE_MIN;   // `|| 'min'` (index.js:1257)
  const int n = n_images;
  const bool vertical = direction == IST_VERTICAL;

  // index.js:1211-1212
  double bytes = 0.0;
  for (int i = 0; i < n; ++i) bytes += images[i].file_size > 0 ? static_cast<double>(images[i].file_size) : 0.0;
  const bool big_task = n >= 7 || bytes >= 25.0 * 1024.0 * 1024.0;

  // index.js:1236-1244: every image ends stage 1 with natural sizes >= 1
  std::vector<Size> nat(n);
  for (int i = 0; i < n; ++i) {
    nat[i].w = std::max(1.0, js_or(static_cast<double>(images[i].width), 1.0));
    nat[i].h = std::max(1.0, js_or(static_cast<double>(images[i].height), 1.0));
  }
  // index.js:1252-1254
  if (nat.empty()) return fail(IST_E_SIZE_UNAVAILABLE, "图片尺寸不可用");
  const auto by_w = std::minmax_element(nat.begin(), nat.end(), [](const Size& p, const Size& q) { return p.w < q.w; });
  const auto by_h = std::minmax_element(nat.begin(), nat.end(), [](const Size& p, const Size& q) { return p.h < q.h; });

  const double gap_px = js_or(gap, 0.0);                              // index.js:1256

  // index.js:1260-1315.  `along` is the stacking axis, `across` the common one.
  double across;
  if (mode == IST_MODE_MIN) across = vertical ? by_w.first->w : by_h.first->h;
  else                      across = vertical ? by_w.second->w : by_h.second->h;
  double along = 0.0;
  for (int i = 0; i < n; ++i) {
    double extent;
    if (mode == IST_MODE_ORIGINAL) extent = vertical ? nat[i].h : nat[i].w;
    else if (vertical)             extent = nat[i].h * (across / nat[i].w);
    else                           extent = nat[i].w * (across / nat[i].h);
    along = along + extent + (i ? gap_px : 0.0);
  }
  double out_w = vertical ? across : along;
  double out_h = vertical ? along : across;
  out_w = std::max(1.0, js_round(out_w));                             // index.js:1318-1319
  out_h = std::max(1.0, js_round(out_h));
  if (!(out_w >= 1.0) || !(out_h >= 1.0) || !std::isfinite(out_w) || !std::isfinite(out_h))
    return fail(IST_E_OUTPUT_SIZE, "输出尺寸计算失败");                // index.js:1320

  // index.js:1324-1336
  const bool ios = limits->platform == IST_PLATFORM_IOS, android = limits->platform == IST_PLATFORM_ANDROID;
  const double side_cap = js_or(limits->max_side, android ? 4096.0 : 12288.0);
  double platform_px;
  if (ios)          platform_px = 16384.0 * 1400.0;
  else if (android) platform_px = side_cap * std::min(side_cap, 8192.0);
  else              platform_px = side_cap * side_cap;
  const double px_cap = std::min(js_or(limits->max_pixels, platform_px), platform_px);

  // index.js:1337-1357
  double shrink = 1.0;
  if (out_w > side_cap || out_h > side_cap) shrink = std::min(side_cap / out_w, side_cap / out_h);
  const double total = out_w * out_h;
  if (total > px_cap) shrink = std::min(shrink, std::sqrt(px_cap / total));
  if (shrink < 1.0) {
    out_w = std::max(1.0, std::floor(out_w * shrink));
    out_h = std::max(1.0, std::floor(out_h * shrink));
  }

  // index.js:1360-1383
  const double base_px = out_w * out_h;
  double ss_cap = big_task ? 1.0 : (ios ? 2.2 : 2.6);
  if (limits->max_super_sample > 0.0) ss_cap = limits->max_super_sample;
  double ss = 1.0;
  if (base_px > 0.0 && base_px < px_cap) {
    const double ratio = std::sqrt(px_cap / base_px);
    if (ratio > 1.01) ss = std::min(std::min(ss_cap, ratio), std::min(side_cap / out_w, side_cap / out_h));
  }
  if (!std::isfinite(ss) || ss < 1.0) ss = 1.0;
  double canvas_w = std::max(1.0, js_round(out_w * ss));
  double canvas_h = std::max(1.0, js_round(out_h * ss));
  for (int guard = 0; canvas_w * canvas_h > px_cap && guard < 20; ++guard) {
    ss *= 0.96;
    if (ss < 1.0) { ss = 1.0; break; }
    canvas_w = std::max(1.0, std::floor(out_w * ss));
    canvas_h = std::max(1.0, std::floor(out_h * ss));
  }
  if (canvas_w > 2147483647.0 / 4.0 || canvas_h > 2147483647.0)
    return fail(IST_E_OUTPUT_SIZE, "输出尺寸计算失败: canvas side exceeds 2^29");

  // index.js:1432-1433, 1522-1554
  ist_rect* rects = static_cast<ist_rect*>(std::calloc(static_cast<size_t>(n), sizeof(ist_rect)));
  if (!rects) return fail(IST_E_NOMEM, "out of memory");
  const double step_gap = gap_px * shrink;
  double cursor = 0.0;
  for (int i = 0; i < n; ++i) {
    ist_rect& r = rects[i];
    r.image = i;
    r.orientation = images[i].orientation;
    if (mode == IST_MODE_ORIGINAL) {
      r.dw = js_round(nat[i].w * shrink);
      r.dh = js_round(nat[i].h * shrink);
      if (vertical) { r.dx = std::floor((out_w - r.dw) / 2.0); r.dy = cursor; cursor += r.dh + step_gap; }
      else          { r.dy = std::floor((out_h - r.dh) / 2.0); r.dx = cursor; cursor += r.dw + step_gap; }
    } else if (vertical) {
      r.dx = 0.0; r.dy = cursor; r.dw = out_w;
      r.dh = js_round(nat[i].h * (out_w / nat[i].w));
      cursor += r.dh + step_gap;
    } else {
      r.dy = 0.0; r.dx = cursor; r.dh = out_h;
      r.dw = js_round(nat[i].w * (out_h / nat[i].h));
      cursor += r.dw + step_gap;
    }
  }
  out->out_w = out_w; out->out_h = out_h;
  out->scale_down = shrink; out->super_sample = ss;
  out->canvas_w = static_cast<int64_t>(canvas_w); out->canvas_h = static_cast<int64_t>(canvas_h);
  out->big_task = big_task ? 1 : 0;
  out->n_rects = n;
  out->rects = rects;
  return IST_OK;
}

void ist_plan_free(ist_plan* plan) {
  if (!plan) return;
  std::free(plan->rects);
  plan->rects = nullptr;
  plan->n_rects = 0;
}

int ist_plan_ops(const ist_plan* plan, const ist_image_desc* images, int n_images, ist_op* ops, int* n_ops) {
  if (!plan || !images || !ops || !n_ops) return fail(IST_E_INVALID, "ist_plan_ops: NULL argument");
  int k = 0;
  // index.js:1423-1424: fillStyle '#ffffff'; fillRect(0,0,canvasOutW,canvasOutH) under the identity CTM
  ist_op& bg = ops[k++];
  std::memset(&bg, 0, sizeof(bg));
  bg.kind = 0; bg.image = -1;
  bg.m[0] = 1.0; bg.m[3] = 1.0;
  bg.d[2] = static_cast<double>(plan->canvas_w); bg.d[3] = static_cast<double>(plan->canvas_h);
  bg.rgba[0] = bg.rgba[1] = bg.rgba[2] = bg.rgba[3] = 255;

  const double kHalfPi = 0.5 * 3.141592653589793;                     // 0.5 * Math.PI
  for (int i = 0; i < plan->n_rects; ++i) {
    const ist_rect& r = plan->rects[i];
    if (r.image < 0 || r.image >= n_images) return fail(IST_E_INVALID, "rect refers to a missing image");
    const ist_image_desc& im = images[r.image];
    const double bw = im.bmp_width > 0 ? im.bmp_width : im.width;
    const double bh = im.bmp_height > 0 ? im.bmp_height : im.height;
    if (!(bw >= 1.0) || !(bh >= 1.0)) return fail(IST_E_DECODE, "图片" + std::to_string(r.image) + "解码异常");  // index.js:1512-1514
    Ctm t;                                                            // fresh canvas: identity
    if (plan->super_sample != 1.0) t.scale(plan->super_sample, plan->super_sample);   // index.js:1426-1428
    double rx = r.dx, ry = r.dy, rw = r.dw, rh = r.dh;
    switch (r.orientation) {                                          // utils/canvas.js:160-200
      case 2: t.translate(r.dx + r.dw, r.dy);        t.scale(-1.0, 1.0);                      rx = ry = 0.0; break;
      case 3: t.translate(r.dx + r.dw, r.dy + r.dh); t.rotate(3.141592653589793);             rx = ry = 0.0; break;
      case 4: t.translate(r.dx, r.dy + r.dh);        t.scale(1.0, -1.0);                      rx = ry = 0.0; break;
      case 5: t.translate(r.dx, r.dy);               t.rotate(kHalfPi); t.scale(1.0, -1.0);   rx = ry = 0.0; rw = r.dh; rh = r.dw; break;
      case 6: t.translate(r.dx + r.dw, r.dy);        t.rotate(kHalfPi);                       rx = ry = 0.0; rw = r.dh; rh = r.dw; break;
      case 7: t.translate(r.dx + r.dw, r.dy);        t.rotate(kHalfPi); t.scale(-1.0, 1.0);   rx = ry = 0.0; rw = r.dh; rh = r.dw; break;
      case 8: t.translate(r.dx, r.dy + r.dh);        t.rotate(-kHalfPi);                      rx = ry = 0.0; rw = r.dh; rh = r.dw; break;
      default: break;                                                 // falsy, 1 and unknown values: plain draw (:155-158, :198-199)
    }
    ist_op& o = ops[k++];
    std::memset(&o, 0, sizeof(o));
    o.kind = 1; o.image = r.image;
    o.m[0] = t.a; o.m[1] = t.b; o.m[2] = t.c; o.m[3] = t.d; o.m[4] = t.e; o.m[5] = t.f;
    o.s[0] = 0.0; o.s[1] = 0.0; o.s[2] = bw; o.s[3] = bh;             // always the full bitmap (index.js:1532)
    o.d[0] = rx; o.d[1] = ry; o.d[2] = rw; o.d[3] = rh;
  }
  *n_ops = k;
  return IST_OK;
}

}  // extern "C"
