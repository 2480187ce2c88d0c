/*
 * ist_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under imagestitching_amd/ or node/ links, imports or calls it.
 *
 * What it restates (reference = Iamctb/ImageStitching, paths relative to
 * miniprogram-stitch/miniprogram/):
 *   - the stitch planner            pages/index/index.js:1211-1216, 1251-1386, 1432-1433, 1522-1554
 *   - drawWithOrientation           utils/canvas.js:153-202   (the Canvas call sequence it issues)
 *   - canvas init                   pages/index/index.js:1423-1428 (white fill, optional ctx.scale)
 *   - a Canvas-2D raster for exactly the calls above (fillRect / save / restore / translate / rotate /
 *     scale / 9-argument drawImage).  The reference's raster is the closed WeChat client (base library
 *     3.10.3, project.private.config.json:2) and is absent from /root/reference, so this part restates the
 *     published HTML Canvas drawImage contract instead of reference source:
 *        destination pixel centre (X+0.5, Y+0.5) is inverse-mapped through the CTM and the
 *        (sx,sy,sw,sh)->(dx,dy,dw,dh) affine; imageSmoothingEnabled=true -> bilinear, false -> nearest;
 *        samples outside the source rectangle clamp to its edge; source-over compositing.
 *
 * PARITY PIN STATUS
 *   planner / call sequence : PINNED  - tests/golden/plan_goldens.json holds call traces captured from the
 *                             reference's own index.js run unmodified under Node (oracle/capture_plan_goldens.js).
 *   pixels                  : PARITY UNPINNED by the reference (it has no tests, no fixtures and its raster is
 *                             not in the repo).  Pinned instead against independent witnesses (cairo/pixman,
 *                             torch interpolate, PIL affine) by oracle/witness_check.py -> tests/golden/pixel_*.npz.
 *
 * Arithmetic contract (shared with the product so that results can be compared bit for bit):
 *   all coordinate arithmetic in IEEE double, no FMA contraction (build with -ffp-contract=off);
 *   a draw is "resolved" into canvas space as
 *        sxf = kx * Wc + ox      syf = ky * Zc + oy        (Wc,Zc) = swap ? (Yc,Xc) : (Xc,Yc), Xc = X + 0.5
 *   with kx, ox, ky, oy computed exactly as in orc_resolve() below;
 *   nearest : ix = clamp(floor(sxf)), iy = clamp(floor(syf)); integer source-over
 *             out = (c*a + dst*(255-a) + 127) / 255
 *   bilinear: f = sxf-0.5, i0 = floor(f), t = f-i0, taps clamp(i0), clamp(i0+1) (same in y); taps are
 *             premultiplied (c*a/255, real-valued), filtered, composited, then rounded half up once.
 *   coverage: a canvas pixel belongs to a draw iff its centre lies inside the transformed destination
 *             rectangle; with ORC_EDGE_AA fractional edges are anti-aliased instead: cov = area of the pixel
 *             inside the rectangle, out = floor(P*cov + dst*(1 - cov*A/255) + 0.5) in double.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

enum { ORC_VERTICAL = 0, ORC_HORIZONTAL = 1 };
enum { ORC_MODE_MIN = 0, ORC_MODE_MAX = 1, ORC_MODE_ORIGINAL = 2 };
enum { ORC_PLATFORM_OTHER = 0, ORC_PLATFORM_IOS = 1, ORC_PLATFORM_ANDROID = 2 };
enum { ORC_NEAREST = 0, ORC_BILINEAR = 1, ORC_AREA = 2, ORC_EDGE_AA = 0x100 };   /* filter | ORC_EDGE_AA: coverage anti-aliasing of fractional edges */

typedef struct {
  int32_t width, height;     /* naturalWidth / naturalHeight (index.js:724-739) */
  int32_t orientation;       /* EXIF 1..8, 0 = unset */
  int32_t bmp_w, bmp_h;      /* decoded bitmap size (bmp.width/height, index.js:1532); 0 -> same as natural */
  int64_t file_size;         /* bytes, feeds bigTask (index.js:1211-1212) */
} orc_image;

typedef struct {
  int32_t platform;          /* ORC_PLATFORM_* : sys.platform */
  double max_side;           /* this.deviceMaxCanvasSize   (0 = unset -> reference fallback) */
  double max_pixels;         /* this.deviceMaxCanvasPixels (0 = unset -> reference fallback) */
  double max_super_sample;   /* <= 0: reference rule (index.js:1363); > 0: replaces MAX_SUPER_SAMPLE */
} orc_limits;

typedef struct {
  int32_t image;             /* index into images */
  int32_t orientation;
  double dx, dy, dw, dh;     /* drawWithOrientation destination rectangle, user space */
} orc_rect;

typedef struct {
  double out_w, out_h;       /* target size after caps (index.js:1360-1361) */
  double scale_down, super_sample;
  double canvas_w, canvas_h; /* canvasOutW/H (index.js:1373-1383) */
  int32_t big_task;
  int32_t n_rects;
} orc_plan;

/* Math.round: round half toward +inf */
static double js_round(double x) {
  double r = floor(x);
  return (x - r >= 0.5) ? r + 1.0 : r;
}
static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }

/* index.js:126-156 (fallback branch): what onLoad leaves in deviceMaxCanvasSize/Pixels when storage is empty */
ORC_API void orc_default_limits(int platform, orc_limits* out) {
  double side = platform == ORC_PLATFORM_IOS ? 12288.0 : 4096.0;
  double cap = platform == ORC_PLATFORM_ANDROID ? side * dmin(side, 4096.0) : side * dmin(side, 12288.0);
  out->platform = platform;
  out->max_side = side;
  out->max_pixels = dmax(cap, 4096.0 * 2048.0);
  out->max_super_sample = 0.0;
}

/* return: 0 ok, 1 nothing to do (no images), <0 error (-1 sizes unusable, -2 output size failed) */
ORC_API int orc_plan_compute(const orc_image* imgs, int n, int direction, int mode, double gap,
                             const orc_limits* lim, orc_plan* plan, orc_rect* rects) {
  if (n <= 0) return 1;                                         /* index.js:1189 */
  /* index.js:1211-1212 */
  double total_bytes = 0.0;
  for (int i = 0; i < n; i++) total_bytes += imgs[i].file_size > 0 ? (double)imgs[i].file_size : 0.0;
  int big_task = (n >= 7) || (total_bytes >= 25.0 * 1024.0 * 1024.0);

  /* index.js:1236-1244: naturalWidth = max(1, naturalWidth || width || 1) */
  double* nw = (double*)malloc(sizeof(double) * 2 * (size_t)n);
  double* nh = nw + n;
  for (int i = 0; i < n; i++) {
    nw[i] = dmax(1.0, imgs[i].width != 0 ? (double)imgs[i].width : 1.0);
    nh[i] = dmax(1.0, imgs[i].height != 0 ? (double)imgs[i].height : 1.0);
  }
  /* index.js:1252-1254 (after stage 1 every entry is >= 1, so the filter keeps all of them) */
  double min_w = nw[0], max_w = nw[0], min_h = nh[0], max_h = nh[0];
  for (int i = 1; i < n; i++) {
    min_w = dmin(min_w, nw[i]); max_w = dmax(max_w, nw[i]);
    min_h = dmin(min_h, nh[i]); max_h = dmax(max_h, nh[i]);
  }
  double gap_px = (gap == gap && gap != 0.0) ? gap : 0.0;        /* gap || 0 (index.js:1256) */

  double out_w = 0.0, out_h = 0.0;
  if (direction == ORC_VERTICAL) {                               /* index.js:1261-1287 */
    if (mode == ORC_MODE_MIN || mode == ORC_MODE_MAX) {
      out_w = mode == ORC_MODE_MIN ? min_w : max_w;
      double sum = 0.0;
      for (int i = 0; i < n; i++) {
        double draw_h = nh[i] * (out_w / nw[i]);
        sum = sum + draw_h + (i ? gap_px : 0.0);
      }
      out_h = sum;
    } else {
      out_w = max_w;
      double sum = 0.0;
      for (int i = 0; i < n; i++) sum = sum + nh[i] + (i ? gap_px : 0.0);
      out_h = sum;
    }
  } else {                                                       /* index.js:1288-1315 */
    if (mode == ORC_MODE_MIN || mode == ORC_MODE_MAX) {
      out_h = mode == ORC_MODE_MIN ? min_h : max_h;
      double sum = 0.0;
      for (int i = 0; i < n; i++) {
        double draw_w = nw[i] * (out_h / nh[i]);
        sum = sum + draw_w + (i ? gap_px : 0.0);
      }
      out_w = sum;
    } else {
      out_h = max_h;
      double sum = 0.0;
      for (int i = 0; i < n; i++) sum = sum + nw[i] + (i ? gap_px : 0.0);
      out_w = sum;
    }
  }
  out_w = dmax(1.0, js_round(out_w));                            /* index.js:1318-1320 */
  out_h = dmax(1.0, js_round(out_h));
  if (!(out_w > 0.0) || !(out_h > 0.0)) { free(nw); return -2; }

  /* index.js:1323-1357 */
  int ios = lim->platform == ORC_PLATFORM_IOS, android = lim->platform == ORC_PLATFORM_ANDROID;
  double max_side = lim->max_side != 0.0 ? lim->max_side : (android ? 4096.0 : 12288.0);
  double max_total;
  if (ios) {
    double l = 16384.0 * 1400.0;
    max_total = dmin(lim->max_pixels != 0.0 ? lim->max_pixels : l, l);
  } else if (android) {
    double l = max_side * dmin(max_side, 8192.0);
    max_total = dmin(lim->max_pixels != 0.0 ? lim->max_pixels : l, l);
  } else {
    double l = max_side * max_side;
    max_total = dmin(lim->max_pixels != 0.0 ? lim->max_pixels : l, l);
  }
  double scale_down = 1.0;
  if (out_w > max_side || out_h > max_side) scale_down = dmin(max_side / out_w, max_side / out_h);
  double total_pixels = out_w * out_h;
  if (total_pixels > max_total) scale_down = dmin(scale_down, sqrt(max_total / total_pixels));
  if (scale_down < 1.0) {
    out_w = dmax(1.0, floor(out_w * scale_down));
    out_h = dmax(1.0, floor(out_h * scale_down));
  }

  /* index.js:1360-1383 */
  double target_w = out_w, target_h = out_h, base = target_w * target_h;
  double max_ss = big_task ? 1.0 : (ios ? 2.2 : 2.6);
  if (lim->max_super_sample > 0.0) max_ss = lim->max_super_sample;
  double ss = 1.0;
  if (base > 0.0 && base < max_total) {
    double ratio = sqrt(max_total / base);
    if (ratio > 1.01) {
      double side_cap = dmin(max_side / target_w, max_side / target_h);
      ss = dmin(dmin(max_ss, ratio), side_cap);
    }
  }
  if (!isfinite(ss) || ss < 1.0) ss = 1.0;
  double cw = dmax(1.0, js_round(target_w * ss)), ch = dmax(1.0, js_round(target_h * ss));
  for (int guard = 0; cw * ch > max_total && guard < 20; guard++) {
    ss *= 0.96;
    if (ss < 1.0) { ss = 1.0; break; }
    cw = dmax(1.0, floor(target_w * ss));
    ch = dmax(1.0, floor(target_h * ss));
  }

  /* index.js:1432-1433, 1522-1554 */
  double scaled_gap = gap_px * scale_down, cx = 0.0, cy = 0.0;
  for (int i = 0; i < n; i++) {
    orc_rect* r = &rects[i];
    r->image = i;
    r->orientation = imgs[i].orientation;
    if (direction == ORC_VERTICAL) {
      if (mode == ORC_MODE_ORIGINAL) {
        double dw = js_round(nw[i] * scale_down), dh = js_round(nh[i] * scale_down);
        r->dx = floor((out_w - dw) / 2.0); r->dy = cy; r->dw = dw; r->dh = dh;
        cy += dh + scaled_gap;
      } else {
        double draw_h = js_round(nh[i] * (out_w / nw[i]));
        r->dx = 0.0; r->dy = cy; r->dw = out_w; r->dh = draw_h;
        cy += draw_h + scaled_gap;
      }
    } else {
      if (mode == ORC_MODE_ORIGINAL) {
        double dw = js_round(nw[i] * scale_down), dh = js_round(nh[i] * scale_down);
        r->dx = cx; r->dy = floor((out_h - dh) / 2.0); r->dw = dw; r->dh = dh;
        cx += dw + scaled_gap;
      } else {
        double draw_w = js_round(nw[i] * (out_h / nh[i]));
        r->dx = cx; r->dy = 0.0; r->dw = draw_w; r->dh = out_h;
        cx += draw_w + scaled_gap;
      }
    }
  }
  plan->out_w = out_w; plan->out_h = out_h;
  plan->scale_down = scale_down; plan->super_sample = ss;
  plan->canvas_w = cw; plan->canvas_h = ch;
  plan->big_task = big_task; plan->n_rects = n;
  free(nw);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Canvas-2D emulator: state + the calls the path uses.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { double a, b, c, d, e, f; } orc_mat;   /* X = a*u + c*v + e ; Y = b*u + d*v + f */

typedef struct {
  int32_t w, h;
  uint8_t* px;          /* premultiplied RGBA8, row-major */
  size_t pitch;
  orc_mat m;
  orc_mat stack[16];
  int sp;
  int smoothing;        /* imageSmoothingEnabled */
  int area;             /* ORC_AREA: minified axes average the source over the output pixel's footprint (box), others stay bilinear */
  int aa;               /* anti-alias fractional rectangle edges by area coverage (off: pixel-centre rule) */
  int y0, y1;           /* raster band [y0,y1) for the multithreaded baseline */
} orc_canvas;

static void cv_init(orc_canvas* cv, int w, int h, uint8_t* px, size_t pitch, int smoothing, int y0, int y1) {
  cv->w = w; cv->h = h; cv->px = px; cv->pitch = pitch; cv->sp = 0; cv->smoothing = (smoothing & 3) != 0; cv->area = (smoothing & 3) == ORC_AREA; cv->aa = (smoothing & ORC_EDGE_AA) != 0;
  cv->m.a = 1; cv->m.b = 0; cv->m.c = 0; cv->m.d = 1; cv->m.e = 0; cv->m.f = 0;
  cv->y0 = y0 < 0 ? 0 : y0; cv->y1 = y1 > h ? h : y1;
}
static void cv_save(orc_canvas* cv) { if (cv->sp < 16) cv->stack[cv->sp++] = cv->m; }
static void cv_restore(orc_canvas* cv) { if (cv->sp > 0) cv->m = cv->stack[--cv->sp]; }
static void cv_translate(orc_canvas* cv, double x, double y) {
  cv->m.e = cv->m.a * x + cv->m.c * y + cv->m.e;
  cv->m.f = cv->m.b * x + cv->m.d * y + cv->m.f;
}
static void cv_scale(orc_canvas* cv, double x, double y) {
  cv->m.a *= x; cv->m.b *= x; cv->m.c *= y; cv->m.d *= y;
}
/* rotate: angles within 1e-9 of a multiple of pi/2 use exact (cos,sin) so that EXIF turns stay axis aligned */
static void cv_rotate(orc_canvas* cv, double r) {
  double q = r / 1.5707963267948966, qr = js_round(q), co, si;
  if (fabs(q - qr) < 1e-9) {
    int k = (int)fmod(fmod(qr, 4.0) + 4.0, 4.0);
    co = k == 0 ? 1.0 : (k == 2 ? -1.0 : 0.0);
    si = k == 1 ? 1.0 : (k == 3 ? -1.0 : 0.0);
  } else { co = cos(r); si = sin(r); }
  orc_mat m = cv->m;
  cv->m.a = m.a * co + m.c * si; cv->m.b = m.b * co + m.d * si;
  cv->m.c = m.c * co - m.a * si; cv->m.d = m.d * co - m.b * si;
}

static void cv_fill_rect_opaque(orc_canvas* cv, double x, double y, double w, double h, const uint8_t rgba[4]) {
  /* only the identity-CTM full-pixel case is on the path (index.js:1423-1424) */
  double xa = cv->m.a * x + cv->m.e, xb = cv->m.a * (x + w) + cv->m.e;
  double ya = cv->m.d * y + cv->m.f, yb = cv->m.d * (y + h) + cv->m.f;
  double xl = dmin(xa, xb), xh = dmax(xa, xb), yl = dmin(ya, yb), yh = dmax(ya, yb);
  double X0 = ceil(xl - 0.5), X1 = ceil(xh - 0.5), Y0 = ceil(yl - 0.5), Y1 = ceil(yh - 0.5);
  if (X0 < 0) X0 = 0; if (Y0 < cv->y0) Y0 = cv->y0;
  if (X1 > cv->w) X1 = cv->w; if (Y1 > cv->y1) Y1 = cv->y1;
  for (int Y = (int)Y0; Y < (int)Y1; Y++) {
    uint8_t* row = cv->px + (size_t)Y * cv->pitch;
    for (int X = (int)X0; X < (int)X1; X++) memcpy(row + 4 * (size_t)X, rgba, 4);
  }
}

/* resolved draw: canvas-space axis-aligned sampling map (the shared arithmetic contract) */
typedef struct {
  int swap;
  double kx, ox, ky, oy;
  int X0, X1, Y0, Y1;          /* covered canvas pixels, clipped */
  int cx0, cx1, cy0, cy1;      /* inclusive clamp bounds in the source */
  double xl, xh, yl, yh;       /* continuous canvas-space extent of the destination rectangle */
} orc_resolved;

static int orc_resolve(const orc_mat* m, int cw, int ch, int img_w, int img_h,
                       double sx, double sy, double sw, double sh,
                       double rx, double ry, double rw, double rh, int aa, orc_resolved* o) {
  int noswap = (m->b == 0.0 && m->c == 0.0 && m->a != 0.0 && m->d != 0.0);
  int swap = (m->a == 0.0 && m->d == 0.0 && m->b != 0.0 && m->c != 0.0);
  if (!noswap && !swap) return -1;
  if (!(rw > 0.0) || !(rh > 0.0) || !(sw > 0.0) || !(sh > 0.0)) return 1;     /* nothing drawn */
  double ku, eu, kv, ev;
  if (!swap) { ku = m->a; eu = m->e; kv = m->d; ev = m->f; }
  else       { ku = m->b; eu = m->f; kv = m->c; ev = m->e; }
  double gx = sw / rw, gy = sh / rh;
  o->swap = swap;
  o->kx = gx / ku; o->ox = sx - (eu / ku + rx) * gx;
  o->ky = gy / kv; o->oy = sy - (ev / kv + ry) * gy;
  double wa = ku * rx + eu, wb = ku * (rx + rw) + eu;        /* extent along the axis that drives source x */
  double za = kv * ry + ev, zb = kv * (ry + rh) + ev;        /* extent along the axis that drives source y */
  double wl = dmin(wa, wb), wh = dmax(wa, wb), zl = dmin(za, zb), zh = dmax(za, zb);
  double W0 = ceil(wl - 0.5), W1 = ceil(wh - 0.5), Z0 = ceil(zl - 0.5), Z1 = ceil(zh - 0.5);
  if (aa) { W0 = floor(wl); W1 = ceil(wh); Z0 = floor(zl); Z1 = ceil(zh); }     /* every pixel the rectangle touches */
  o->xl = swap ? zl : wl; o->xh = swap ? zh : wh; o->yl = swap ? wl : zl; o->yh = swap ? wh : zh;
  double X0 = swap ? Z0 : W0, X1 = swap ? Z1 : W1, Y0 = swap ? W0 : Z0, Y1 = swap ? W1 : Z1;
  if (X0 < 0) X0 = 0; if (Y0 < 0) Y0 = 0;
  if (X1 > cw) X1 = cw; if (Y1 > ch) Y1 = ch;
  o->X0 = (int)X0; o->X1 = (int)X1; o->Y0 = (int)Y0; o->Y1 = (int)Y1;
  double c;
  c = floor(sx); o->cx0 = c < 0 ? 0 : (int)c;
  c = ceil(sx + sw) - 1.0; o->cx1 = c > img_w - 1 ? img_w - 1 : (int)c;
  c = floor(sy); o->cy0 = c < 0 ? 0 : (int)c;
  c = ceil(sy + sh) - 1.0; o->cy1 = c > img_h - 1 ? img_h - 1 : (int)c;
  if (o->cx1 < o->cx0 || o->cy1 < o->cy0) return 1;
  return 0;
}

static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* 9-argument drawImage of a straight-alpha RGBA8 image under the current CTM */
static int cv_draw_image(orc_canvas* cv, const uint8_t* img, int img_w, int img_h, size_t img_pitch,
                         double sx, double sy, double sw, double sh, double rx, double ry, double rw, double rh) {
  orc_resolved R;
  int rc = orc_resolve(&cv->m, cv->w, cv->h, img_w, img_h, sx, sy, sw, sh, rx, ry, rw, rh, cv->aa, &R);
  if (rc != 0) return rc < 0 ? rc : 0;
  int Y0 = R.Y0 < cv->y0 ? cv->y0 : R.Y0, Y1 = R.Y1 > cv->y1 ? cv->y1 : R.Y1;
  if (Y0 >= Y1 || R.X0 >= R.X1) return 0;

  /* per-axis tables: along canvas X (cols) and canvas Y (rows) */
  int nX = R.X1 - R.X0, nY = Y1 - Y0;
  int* xi0 = (int*)malloc(sizeof(int) * 2 * (size_t)(nX + nY));
  int* xi1 = xi0 + nX; int* yi0 = xi1 + nX; int* yi1 = yi0 + nY;
  double* xt = (double*)malloc(sizeof(double) * (size_t)(nX + nY));
  double* yt = xt + nX;
  /* X drives source x (noswap) or source y (swap) */
  for (int i = 0; i < nX; i++) {
    double Xc = (double)(R.X0 + i) + 0.5;
    double s = !R.swap ? R.kx * Xc + R.ox : R.ky * Xc + R.oy;
    int lo = !R.swap ? R.cx0 : R.cy0, hi = !R.swap ? R.cx1 : R.cy1;
    if (cv->smoothing) {
      double f = s - 0.5, fl = floor(f);
      xt[i] = f - fl; xi0[i] = iclamp((int)fl, lo, hi); xi1[i] = iclamp((int)fl + 1, lo, hi);
    } else { xi0[i] = xi1[i] = iclamp((int)floor(s), lo, hi); xt[i] = 0.0; }
  }
  for (int j = 0; j < nY; j++) {
    double Yc = (double)(Y0 + j) + 0.5;
    double s = !R.swap ? R.ky * Yc + R.oy : R.kx * Yc + R.ox;
    int lo = !R.swap ? R.cy0 : R.cx0, hi = !R.swap ? R.cy1 : R.cx1;
    if (cv->smoothing) {
      double f = s - 0.5, fl = floor(f);
      yt[j] = f - fl; yi0[j] = iclamp((int)fl, lo, hi); yi1[j] = iclamp((int)fl + 1, lo, hi);
    } else { yi0[j] = yi1[j] = iclamp((int)floor(s), lo, hi); yt[j] = 0.0; }
  }
  /* area coverage of each pixel column / row by the rectangle (1 everywhere when edge AA is off) */
  double* covx = (double*)malloc(sizeof(double) * (size_t)(nX + nY));
  double* covy = covx + nX;
  for (int i = 0; i < nX; i++) {
    double X = (double)(R.X0 + i);
    double c = cv->aa ? dmin(X + 1.0, R.xh) - dmax(X, R.xl) : 1.0;
    covx[i] = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
  }
  for (int j = 0; j < nY; j++) {
    double Y = (double)(Y0 + j);
    double c = cv->aa ? dmin(Y + 1.0, R.yh) - dmax(Y, R.yl) : 1.0;
    covy[j] = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
  }
  /* ORC_AREA ("imageSmoothingQuality = 'high'" read as area averaging, index.js:1419; an OPTION, the contract's default stays
   * bilinear): per source axis the weight function is a box of width max(1, |k|) centred on the sample position; at |k| <= 1
   * that is exactly the bilinear pair, beyond it every source pixel under the output pixel contributes by its overlap.
   * Separable; taps outside the source rectangle clamp to its edge (their weight lands on the edge pixel). */
  if (cv->area && (fabs(R.kx) > 1.0 || fabs(R.ky) > 1.0)) {
    for (int j = 0; j < nY; j++) {
      uint8_t* drow = cv->px + (size_t)(Y0 + j) * cv->pitch + 4 * (size_t)R.X0;
      for (int i = 0; i < nX; i++) {
        const double cov = covx[i] * covy[j];
        if (cov <= 0.0) continue;
        /* canvas X drives source x (or source y when swapped) */
        const double Xc = (double)(R.X0 + i) + 0.5, Yc = (double)(Y0 + j) + 0.5;
        const double sxc = !R.swap ? R.kx * Xc + R.ox : R.kx * Yc + R.ox;
        const double syc = !R.swap ? R.ky * Yc + R.oy : R.ky * Xc + R.oy;
        const double wxw = fabs(R.kx) > 1.0 ? fabs(R.kx) : 1.0, wyw = fabs(R.ky) > 1.0 ? fabs(R.ky) : 1.0;
        const double xlo = sxc - 0.5 * wxw, xhi = sxc + 0.5 * wxw, ylo = syc - 0.5 * wyw, yhi = syc + 0.5 * wyw;
        const int ix0 = (int)floor(xlo), ix1 = (int)ceil(xhi) - 1, iy0 = (int)floor(ylo), iy1 = (int)ceil(yhi) - 1;
        double acc[4] = {0, 0, 0, 0};
        for (int yy = iy0; yy <= iy1; yy++) {
          const double oy = dmin((double)yy + 1.0, yhi) - dmax((double)yy, ylo);
          if (oy <= 0.0) continue;
          const uint8_t* srow = img + (size_t)iclamp(yy, R.cy0, R.cy1) * img_pitch;
          double racc[4] = {0, 0, 0, 0};
          for (int xx = ix0; xx <= ix1; xx++) {
            const double ox = dmin((double)xx + 1.0, xhi) - dmax((double)xx, xlo);
            if (ox <= 0.0) continue;
            const uint8_t* s = srow + 4 * (size_t)iclamp(xx, R.cx0, R.cx1);
            const double a = (double)s[3];
            racc[0] += ox * (s[0] * a); racc[1] += ox * (s[1] * a); racc[2] += ox * (s[2] * a); racc[3] += ox * a;
          }
          acc[0] += oy * racc[0]; acc[1] += oy * racc[1]; acc[2] += oy * racc[2]; acc[3] += oy * racc[3];
        }
        const double norm = 1.0 / (wxw * wyw);
        const double A = acc[3] * norm;
        uint8_t* d = drow + 4 * (size_t)i;
        const double keep = 1.0 - cov * (A / 255.0);
        for (int c = 0; c < 3; c++) {
          const double P = acc[c] * norm / 255.0;
          double v = floor(P * cov + d[c] * keep + 0.5);
          d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
        double va = floor(A * cov + d[3] * keep + 0.5);
        d[3] = (uint8_t)(va < 0 ? 0 : (va > 255 ? 255 : va));
      }
    }
    free(xi0); free(xt); free(covx);
    return 0;
  }
  /* identity fast path (what BASELINE's uniform configs reduce to): 1:1, no swap, integer offset */
  int identity = !R.swap && R.kx == 1.0 && R.ky == 1.0 && R.ox == floor(R.ox) && R.oy == floor(R.oy);

  /* ... whose rows are contiguous source bytes when no index was clamped: an all-opaque row is then one memcpy (what a
   * Canvas backend's identity blit does; same bytes as the per-pixel loop below, which is kept for every other row) */
  const int row_copy = identity && !cv->aa && nX > 0 && xi0[nX - 1] - xi0[0] == nX - 1;

  for (int j = 0; j < nY; j++) {
    uint8_t* drow = cv->px + (size_t)(Y0 + j) * cv->pitch + 4 * (size_t)R.X0;
    if (row_copy) {
      const uint8_t* srow = img + (size_t)yi0[j] * img_pitch + 4 * (size_t)xi0[0];
      uint32_t all = 0xFFFFFFFFu;
      for (int i = 0; i < nX; i++) { uint32_t px; memcpy(&px, srow + 4 * (size_t)i, 4); all &= px; }
      if ((all >> 24) == 255u) { memcpy(drow, srow, 4 * (size_t)nX); continue; }
    }
    if (identity || !cv->smoothing) {
      for (int i = 0; i < nX; i++) {
        int ix = !R.swap ? xi0[i] : yi0[j], iy = !R.swap ? yi0[j] : xi0[i];
        const uint8_t* s = img + (size_t)iy * img_pitch + 4 * (size_t)ix;
        uint8_t* d = drow + 4 * (size_t)i;
        unsigned a = s[3];
        const double cov = covx[i] * covy[j];
        if (cov <= 0.0) continue;
        if (cov < 1.0) {                          /* fractional edge: coverage-weighted source-over, in double */
          const double A = (double)a, keep = 1.0 - cov * (A / 255.0);
          for (int c = 0; c < 3; c++) {
            const double P = (double)(s[c] * a) / 255.0;
            double v = floor(P * cov + d[c] * keep + 0.5);
            d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
          }
          double va = floor(A * cov + d[3] * keep + 0.5);
          d[3] = (uint8_t)(va < 0 ? 0 : (va > 255 ? 255 : va));
          continue;
        }
        if (a == 255) { memcpy(d, s, 4); continue; }
        unsigned ia = 255 - a;
        d[0] = (uint8_t)((s[0] * a + d[0] * ia + 127) / 255);   /* d is premultiplied; s is straight */
        d[1] = (uint8_t)((s[1] * a + d[1] * ia + 127) / 255);
        d[2] = (uint8_t)((s[2] * a + d[2] * ia + 127) / 255);
        d[3] = (uint8_t)((255 * a + d[3] * ia + 127) / 255);
      }
    } else {
      for (int i = 0; i < nX; i++) {
        int ix0, ix1, iy0, iy1; double tx, ty;
        if (!R.swap) { ix0 = xi0[i]; ix1 = xi1[i]; tx = xt[i]; iy0 = yi0[j]; iy1 = yi1[j]; ty = yt[j]; }
        else         { ix0 = yi0[j]; ix1 = yi1[j]; tx = yt[j]; iy0 = xi0[i]; iy1 = xi1[i]; ty = xt[i]; }
        const uint8_t* p00 = img + (size_t)iy0 * img_pitch + 4 * (size_t)ix0;
        const uint8_t* p01 = img + (size_t)iy0 * img_pitch + 4 * (size_t)ix1;
        const uint8_t* p10 = img + (size_t)iy1 * img_pitch + 4 * (size_t)ix0;
        const uint8_t* p11 = img + (size_t)iy1 * img_pitch + 4 * (size_t)ix1;
        uint8_t* d = drow + 4 * (size_t)i;
        double w00 = (1.0 - tx) * (1.0 - ty), w01 = tx * (1.0 - ty), w10 = (1.0 - tx) * ty, w11 = tx * ty;
        double A = w00 * p00[3] + w01 * p01[3] + w10 * p10[3] + w11 * p11[3];
        const double cov = covx[i] * covy[j];
        if (cov <= 0.0) continue;
        double keep = 1.0 - cov * (A / 255.0);
        for (int c = 0; c < 3; c++) {
          double P = (w00 * p00[c] * p00[3] + w01 * p01[c] * p01[3] + w10 * p10[c] * p10[3] + w11 * p11[c] * p11[3]) / 255.0;
          double v = floor(P * cov + d[c] * keep + 0.5);
          d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
        double va = floor(A * cov + d[3] * keep + 0.5);
        d[3] = (uint8_t)(va < 0 ? 0 : (va > 255 ? 255 : va));
      }
    }
  }
  free(xi0); free(xt); free(covx);
  return 0;
}

/* utils/canvas.js:153-202 — the exact call sequence, against the emulator */
static int cv_draw_with_orientation(orc_canvas* cv, const uint8_t* img, int iw, int ih, size_t pitch,
                                    double sx, double sy, double sw, double sh,
                                    double dx, double dy, double dw, double dh, int orientation) {
  int rc;
  cv_save(cv);
  if (!orientation || orientation == 1) {
    rc = cv_draw_image(cv, img, iw, ih, pitch, sx, sy, sw, sh, dx, dy, dw, dh);
    cv_restore(cv);
    return rc;
  }
  const double HALF_PI = 0.5 * 3.141592653589793;
  switch (orientation) {
    case 2: cv_translate(cv, dx + dw, dy); cv_scale(cv, -1, 1);
            rc = cv_draw_image(cv, img, iw, ih, pitch, sx, sy, sw, sh, 0, 0, dw, dh); break;
    case 3: cv_translate(cv, dx + dw, dy + dh); cv_rotate(cv, 3.141592653589793);
            rc = cv_draw_image(cv, img, iw, ih, pitch, sx, sy, sw, sh, 0, 0, dw, dh); break;
    case 4: cv_translate(cv, dx, dy + dh); cv_scale(cv, 1, -1);
            rc = cv_draw_image(cv, img, iw, ih, pitch, sx, sy, sw, sh, 0, 0, dw, dh); break;
    case 5: cv_translate(cv, dx, dy); cv_rotate(cv, HALF_PI); cv_scale(cv, 1, -1);
            rc = cv_draw_image(cv, img, iw, ih, pitch, sx, sy, sw, sh, 0, 0, dh, dw); break;
    case 6: cv_translate(cv, dx + dw, dy); cv_rotate(cv, HALF_PI);
            rc = cv_draw_image(cv, img, iw, ih, pitch, sx, sy, sw, sh, 0, 0, dh, dw); break;
    case 7: cv_translate(cv, dx + dw, dy); cv_rotate(cv, HALF_PI); cv_scale(cv, -1, 1);
            rc = cv_draw_image(cv, img, iw, ih, pitch, sx, sy, sw, sh, 0, 0, dh, dw); break;
    case 8: cv_translate(cv, dx, dy + dh); cv_rotate(cv, -HALF_PI);
            rc = cv_draw_image(cv, img, iw, ih, pitch, sx, sy, sw, sh, 0, 0, dh, dw); break;
    default: rc = cv_draw_image(cv, img, iw, ih, pitch, sx, sy, sw, sh, dx, dy, dw, dh);
  }
  cv_restore(cv);
  return rc;
}

/* index.js:1423-1428 + 1439-1554 raster side: white fill, optional scale(ss), one drawWithOrientation per
 * rect, then un-premultiply (a no-op for the opaque canvas the path produces).  Renders rows [y0,y1). */
static int render_band(int cw, int ch, double ss, const orc_rect* rects, int n,
                       const uint8_t* const* src, const orc_image* imgs, const size_t* pitch,
                       int filter, uint8_t* dst, size_t dst_pitch, int y0, int y1) {
  orc_canvas cv;
  cv_init(&cv, cw, ch, dst, dst_pitch, (filter & 3) | (filter & ORC_EDGE_AA), y0, y1);
  const uint8_t white[4] = {255, 255, 255, 255};
  cv_fill_rect_opaque(&cv, 0, 0, cw, ch, white);
  if (ss != 1.0) cv_scale(&cv, ss, ss);
  for (int i = 0; i < n; i++) {
    const orc_rect* r = &rects[i];
    const orc_image* im = &imgs[r->image];
    int bw = im->bmp_w > 0 ? im->bmp_w : im->width, bh = im->bmp_h > 0 ? im->bmp_h : im->height;
    size_t p = pitch ? pitch[r->image] : (size_t)bw * 4;
    int rc = cv_draw_with_orientation(&cv, src[r->image], bw, bh, p, 0, 0, bw, bh,
                                      r->dx, r->dy, r->dw, r->dh, r->orientation);
    if (rc < 0) return rc;
  }
  return 0;
}

typedef struct {
  int cw, ch; double ss; const orc_rect* rects; int n; const uint8_t* const* src; const orc_image* imgs;
  const size_t* pitch; int filter; uint8_t* dst; size_t dst_pitch; int y0, y1; int rc;
} band_job;
static void* band_main(void* p) {
  band_job* j = (band_job*)p;
  j->rc = render_band(j->cw, j->ch, j->ss, j->rects, j->n, j->src, j->imgs, j->pitch, j->filter, j->dst, j->dst_pitch, j->y0, j->y1);
  return 0;
}

/* render a plan: dst is straight RGBA8 (alpha is 255 everywhere for this path) */
ORC_API int orc_render(int canvas_w, int canvas_h, double super_sample, const orc_rect* rects, int n_rects,
                       const orc_image* imgs, const uint8_t* const* src, const size_t* src_pitch,
                       int filter, uint8_t* dst, size_t dst_pitch, int threads) {
  if (threads < 1) threads = 1;
  if (threads > canvas_h) threads = canvas_h;
  if (threads > 256) threads = 256;
  band_job jobs[256]; pthread_t th[256];
  for (int t = 0; t < threads; t++) {
    band_job* j = &jobs[t];
    j->cw = canvas_w; j->ch = canvas_h; j->ss = super_sample; j->rects = rects; j->n = n_rects; j->src = src;
    j->imgs = imgs; j->pitch = src_pitch; j->filter = filter; j->dst = dst; j->dst_pitch = dst_pitch;
    j->y0 = (int)((int64_t)canvas_h * t / threads); j->y1 = (int)((int64_t)canvas_h * (t + 1) / threads); j->rc = 0;
  }
  if (threads == 1) band_main(&jobs[0]);
  else {
    for (int t = 0; t < threads; t++) pthread_create(&th[t], 0, band_main, &jobs[t]);
    for (int t = 0; t < threads; t++) pthread_join(th[t], 0);
  }
  for (int t = 0; t < threads; t++) if (jobs[t].rc) return jobs[t].rc;
  return 0;
}

/* plan + render in one call: the restated onStitch stages 2-5 (minus decode / PNG encode) */
ORC_API int orc_stitch(const orc_image* imgs, int n, const uint8_t* const* src, const size_t* src_pitch,
                       int direction, int mode, double gap, const orc_limits* lim, int filter,
                       orc_plan* plan, uint8_t** out_px, int threads) {
  orc_rect* rects = (orc_rect*)malloc(sizeof(orc_rect) * (size_t)(n > 0 ? n : 1));
  int rc = orc_plan_compute(imgs, n, direction, mode, gap, lim, plan, rects);
  if (rc != 0) { free(rects); return rc; }
  size_t cw = (size_t)plan->canvas_w, ch = (size_t)plan->canvas_h;
  uint8_t* px = (uint8_t*)malloc(cw * ch * 4);
  if (!px) { free(rects); return -3; }
  rc = orc_render((int)cw, (int)ch, plan->super_sample, rects, n, imgs, src, src_pitch, filter, px, cw * 4, threads);
  free(rects);
  if (rc) { free(px); return rc; }
  *out_px = px;
  return 0;
}
ORC_API void orc_free(void* p) { free(p); }

/* ---- generic op-list raster (for the Canvas-shim level parity tests) -------------------------------
 * ops: kind 0 = fillRect(opaque colour) under matrix m; kind 1 = drawImage(9 args) under matrix m. */
typedef struct {
  int32_t kind, image;
  double m[6];
  double s[4];          /* sx, sy, sw, sh */
  double d[4];          /* dx, dy, dw, dh  (fill: x, y, w, h) */
  uint8_t rgba[4]; int32_t pad;
} orc_op;

ORC_API int orc_render_ops(int canvas_w, int canvas_h, const uint8_t clear_rgba[4], const orc_op* ops, int n_ops,
                           const orc_image* imgs, const uint8_t* const* src, const size_t* src_pitch,
                           int filter, uint8_t* dst, size_t dst_pitch) {
  orc_canvas cv;
  cv_init(&cv, canvas_w, canvas_h, dst, dst_pitch, (filter & 3) | (filter & ORC_EDGE_AA), 0, canvas_h);
  /* initial canvas colour (premultiplied storage) */
  uint8_t pm[4];
  for (int c = 0; c < 3; c++) pm[c] = (uint8_t)((clear_rgba[c] * clear_rgba[3] + 127) / 255);
  pm[3] = clear_rgba[3];
  for (int y = 0; y < canvas_h; y++)
    for (int x = 0; x < canvas_w; x++) memcpy(dst + (size_t)y * dst_pitch + 4 * (size_t)x, pm, 4);
  for (int i = 0; i < n_ops; i++) {
    const orc_op* o = &ops[i];
    cv.m.a = o->m[0]; cv.m.b = o->m[1]; cv.m.c = o->m[2]; cv.m.d = o->m[3]; cv.m.e = o->m[4]; cv.m.f = o->m[5];
    if (o->kind == 0) {
      if (cv.m.b != 0.0 || cv.m.c != 0.0 || o->rgba[3] != 255) return -1;
      cv_fill_rect_opaque(&cv, o->d[0], o->d[1], o->d[2], o->d[3], o->rgba);
    } else {
      const orc_image* im = &imgs[o->image];
      int bw = im->bmp_w > 0 ? im->bmp_w : im->width, bh = im->bmp_h > 0 ? im->bmp_h : im->height;
      size_t p = src_pitch ? src_pitch[o->image] : (size_t)bw * 4;
      int rc = cv_draw_image(&cv, src[o->image], bw, bh, p, o->s[0], o->s[1], o->s[2], o->s[3], o->d[0], o->d[1], o->d[2], o->d[3]);
      if (rc < 0) return rc;
    }
  }
  /* export: un-premultiply to straight alpha */
  for (int y = 0; y < canvas_h; y++) {
    uint8_t* row = dst + (size_t)y * dst_pitch;
    for (int x = 0; x < canvas_w; x++) {
      uint8_t* d = row + 4 * (size_t)x; unsigned a = d[3];
      if (a == 255) continue;
      if (a == 0) { d[0] = d[1] = d[2] = 0; continue; }
      for (int c = 0; c < 3; c++) { unsigned v = (d[c] * 255u + a / 2) / a; d[c] = (uint8_t)(v > 255 ? 255 : v); }
    }
  }
  return 0;
}

/* expose the resolve step so tests can compare the product's resolved draws bit for bit */
ORC_API int orc_resolve_draw(const double m[6], int cw, int ch, int img_w, int img_h, const double s[4], const double d[4],
                             double out_k[4], int out_box[4], int out_clamp[4], int* out_swap) {
  orc_mat M = { m[0], m[1], m[2], m[3], m[4], m[5] };
  orc_resolved R;
  int rc = orc_resolve(&M, cw, ch, img_w, img_h, s[0], s[1], s[2], s[3], d[0], d[1], d[2], d[3], 0, &R);
  if (rc) return rc;
  out_k[0] = R.kx; out_k[1] = R.ox; out_k[2] = R.ky; out_k[3] = R.oy;
  out_box[0] = R.X0; out_box[1] = R.Y0; out_box[2] = R.X1; out_box[3] = R.Y1;
  out_clamp[0] = R.cx0; out_clamp[1] = R.cy0; out_clamp[2] = R.cx1; out_clamp[3] = R.cy1;
  *out_swap = R.swap;
  return 0;
}

/* the CTM that drawWithOrientation leaves at its drawImage call + that call's rectangle (for trace tests) */
ORC_API void orc_orientation_ctm(double ss, double dx, double dy, double dw, double dh, int orientation,
                                 double out_m[6], double out_rect[4]) {
  orc_canvas cv; uint8_t dummy[4];
  cv_init(&cv, 1, 1, dummy, 4, 0, 0, 0);
  if (ss != 1.0) cv_scale(&cv, ss, ss);
  const double HALF_PI = 0.5 * 3.141592653589793;
  double rx = dx, ry = dy, rw = dw, rh = dh;
  switch (orientation) {
    case 2: cv_translate(&cv, dx + dw, dy); cv_scale(&cv, -1, 1); rx = ry = 0; break;
    case 3: cv_translate(&cv, dx + dw, dy + dh); cv_rotate(&cv, 3.141592653589793); rx = ry = 0; break;
    case 4: cv_translate(&cv, dx, dy + dh); cv_scale(&cv, 1, -1); rx = ry = 0; break;
    case 5: cv_translate(&cv, dx, dy); cv_rotate(&cv, HALF_PI); cv_scale(&cv, 1, -1); rx = ry = 0; rw = dh; rh = dw; break;
    case 6: cv_translate(&cv, dx + dw, dy); cv_rotate(&cv, HALF_PI); rx = ry = 0; rw = dh; rh = dw; break;
    case 7: cv_translate(&cv, dx + dw, dy); cv_rotate(&cv, HALF_PI); cv_scale(&cv, -1, 1); rx = ry = 0; rw = dh; rh = dw; break;
    case 8: cv_translate(&cv, dx, dy + dh); cv_rotate(&cv, -HALF_PI); rx = ry = 0; rw = dh; rh = dw; break;
    default: break;
  }
  out_m[0] = cv.m.a; out_m[1] = cv.m.b; out_m[2] = cv.m.c; out_m[3] = cv.m.d; out_m[4] = cv.m.e; out_m[5] = cv.m.f;
  out_rect[0] = rx; out_rect[1] = ry; out_rect[2] = rw; out_rect[3] = rh;
}
