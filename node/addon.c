/*
 * addon.c — thin N-API binding of the C-ABI (include/imagestitch.h) for the TypeScript/Node host.
 *
 * The reference is JavaScript calling the platform Canvas (miniprogram-stitch/miniprogram/pages/index/index.js:1186,
 * utils/canvas.js); this addon is what a maintainer would require() in its place.  No pixel arithmetic here: every
 * function marshals arguments and calls libimagestitch.so.
 *
 *   plan(images, direction, mode, gap, limits)                         -> plan object        (pure CPU)
 *   stitch(images, direction, mode, gap, limits, filter, asPng?, devices?, split?) -> Promise<{width,height,data}>  (napi_async_work)
 *       devices: number[] (devices[0] = root) shards the stitch over several GPUs from this process (ist_stitch_rgba8_multi:
 *       RCCL gather over xGMI); split 0 = by image (round robin), 1 = by band (equal output rows, draw by draw), 2 = by rows (across all draws), 3 = auto
 *   stitchSync(...same...)                                             -> {width,height,data}
 *   render(canvasW, canvasH, clearRGBA, ops, images, filter, region, asPng?) -> Buffer (region pixels, or the PNG file)
 *   encodePng(data, width, height) -> Buffer;  stitch(..., filter, true) resolves {width,height,png}
 *   deviceCount(), lastError(), abiVersion()
 *
 * images[i] = {width, height, orientation?, fileSize?, opaque?, bmpWidth?, bmpHeight?, data?: Uint8Array|Buffer}
 * limits    = {platform: 0|1|2, maxSide, maxPixels, superSample} or null (MI355X default: caps lifted, superSample 1)
 * ops       = Float64Array, 18 doubles per op: kind, image, m[6], s[4], d[4], r|g<<8|b<<16|a<<24, reserved
 */
#define NODE_GYP_MODULE_NAME imagestitch
#include <node_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/imagestitch.h"

#define CHECK(call)                                                      \
  do {                                                                   \
    if ((call) != napi_ok) {                                             \
      napi_throw_error(env, NULL, "N-API call failed: " #call);          \
      return NULL;                                                       \
    }                                                                    \
  } while (0)

static ist_ctx* g_ctx = NULL;

static pthread_once_t g_ctx_once = PTHREAD_ONCE_INIT;
static char g_ctx_err[256];
static void make_ctx(void) {
  g_ctx = ist_ctx_create(0);
  if (!g_ctx) snprintf(g_ctx_err, sizeof g_ctx_err, "%s", ist_last_error());
}
/* one context for the process; NULL (with the reason in g_ctx_err) when there is no HIP device: no CPU fallback */
static ist_ctx* get_ctx(void) {
  pthread_once(&g_ctx_once, make_ctx);
  return g_ctx;
}

static napi_value throw_ist(napi_env env, int code) {
  char msg[512];
  const char* why = ist_last_error();
  /* same shape as the reference's toast: '拼图失败：' + message (index.js:1620) */
  strcpy(msg, "\xe6\x8b\xbc\xe5\x9b\xbe\xe5\xa4\xb1\xe8\xb4\xa5\xef\xbc\x9a");
  strncat(msg, why && *why ? why : "unknown", sizeof(msg) - strlen(msg) - 1);
  char codebuf[16];
  snprintf(codebuf, sizeof codebuf, "%d", code);
  napi_throw_error(env, codebuf, msg);
  return NULL;
}

static int get_named_i64(napi_env env, napi_value obj, const char* key, int64_t* out) {
  napi_value v; napi_valuetype t; double d;
  if (napi_get_named_property(env, obj, key, &v) != napi_ok) return 0;
  if (napi_typeof(env, v, &t) != napi_ok || t != napi_number) return 0;
  if (napi_get_value_double(env, v, &d) != napi_ok) return 0;
  *out = (int64_t)d;
  return 1;
}
static int get_named_f64(napi_env env, napi_value obj, const char* key, double* out) {
  napi_value v; napi_valuetype t;
  if (napi_get_named_property(env, obj, key, &v) != napi_ok) return 0;
  if (napi_typeof(env, v, &t) != napi_ok || t != napi_number) return 0;
  return napi_get_value_double(env, v, out) == napi_ok;
}
static int get_named_bool(napi_env env, napi_value obj, const char* key) {
  napi_value v; napi_valuetype t; bool b = false;
  if (napi_get_named_property(env, obj, key, &v) != napi_ok) return 0;
  if (napi_typeof(env, v, &t) != napi_ok) return 0;
  if (t == napi_boolean) { napi_get_value_bool(env, v, &b); return b; }
  if (t == napi_number) { double d = 0; napi_get_value_double(env, v, &d); return d != 0; }
  return 0;
}

typedef struct {
  int n;
  ist_image_desc* descs;
  const uint8_t** data;
  size_t* pitch;
  napi_ref* refs;          /* keeps the JS buffers alive while async work runs */
} images_t;

static void images_free(napi_env env, images_t* im) {
  if (im->refs) for (int i = 0; i < im->n; i++) if (im->refs[i]) napi_delete_reference(env, im->refs[i]);
  free(im->descs); free(im->data); free(im->pitch); free(im->refs);
  memset(im, 0, sizeof *im);
}

/* returns 0 on failure (exception pending) */
static int images_parse(napi_env env, napi_value arr, images_t* im, int want_refs) {
  bool is_arr = false; uint32_t n = 0;
  memset(im, 0, sizeof *im);
  if (napi_is_array(env, arr, &is_arr) != napi_ok || !is_arr) { napi_throw_type_error(env, NULL, "images must be an array"); return 0; }
  napi_get_array_length(env, arr, &n);
  im->n = (int)n;
  im->descs = (ist_image_desc*)calloc(n ? n : 1, sizeof(ist_image_desc));
  im->data = (const uint8_t**)calloc(n ? n : 1, sizeof(uint8_t*));
  im->pitch = (size_t*)calloc(n ? n : 1, sizeof(size_t));
  im->refs = want_refs ? (napi_ref*)calloc(n ? n : 1, sizeof(napi_ref)) : NULL;
  for (uint32_t i = 0; i < n; i++) {
    napi_value e, d; napi_valuetype t; int64_t v;
    napi_get_element(env, arr, i, &e);
    if (napi_typeof(env, e, &t) != napi_ok || t != napi_object) { napi_throw_type_error(env, NULL, "images[i] must be an object"); return 0; }
    ist_image_desc* D = &im->descs[i];
    if (get_named_i64(env, e, "width", &v)) D->width = (int32_t)v;
    if (get_named_i64(env, e, "height", &v)) D->height = (int32_t)v;
    D->orientation = 1;
    if (get_named_i64(env, e, "orientation", &v)) D->orientation = (int32_t)v;
    if (get_named_i64(env, e, "bmpWidth", &v)) D->bmp_width = (int32_t)v;
    if (get_named_i64(env, e, "bmpHeight", &v)) D->bmp_height = (int32_t)v;
    if (get_named_i64(env, e, "fileSize", &v)) D->file_size = v;
    D->opaque = get_named_bool(env, e, "opaque");
    if (napi_get_named_property(env, e, "data", &d) == napi_ok) {
      bool is_ta = false, is_buf = false;
      napi_is_typedarray(env, d, &is_ta);
      napi_is_buffer(env, d, &is_buf);
      void* p = NULL; size_t len = 0;
      if (is_buf) napi_get_buffer_info(env, d, &p, &len);
      else if (is_ta) {
        napi_typedarray_type tt; napi_value ab; size_t off;
        napi_get_typedarray_info(env, d, &tt, &len, &p, &ab, &off);
        if (tt != napi_uint8_array && tt != napi_uint8_clamped_array) { napi_throw_type_error(env, NULL, "image data must be a Uint8Array / Buffer"); return 0; }
      }
      if (p) {
        const int64_t bw = D->bmp_width > 0 ? D->bmp_width : D->width, bh = D->bmp_height > 0 ? D->bmp_height : D->height;
        if (bw > 0 && bh > 0 && (int64_t)len < bw * bh * 4) { napi_throw_range_error(env, NULL, "image data is smaller than width*height*4"); return 0; }
        im->data[i] = (const uint8_t*)p;
        im->pitch[i] = (size_t)bw * 4;
        if (want_refs) napi_create_reference(env, d, 1, &im->refs[i]);
      }
    }
  }
  return 1;
}

static void limits_parse(napi_env env, napi_value v, ist_limits* lim) {
  napi_valuetype t = napi_undefined;
  napi_typeof(env, v, &t);
  ist_limits_unlimited(lim);
  if (t != napi_object) return;
  int64_t plat = -1; double d;
  if (get_named_i64(env, v, "platform", &plat) && plat >= 0) ist_limits_default((int)plat, lim);
  if (get_named_f64(env, v, "maxSide", &d)) lim->max_side = d;
  if (get_named_f64(env, v, "maxPixels", &d)) lim->max_pixels = d;
  if (get_named_f64(env, v, "superSample", &d)) lim->max_super_sample = d;
}

static void set_num(napi_env env, napi_value obj, const char* key, double v) {
  napi_value n; napi_create_double(env, v, &n); napi_set_named_property(env, obj, key, n);
}

static napi_value plan_to_js(napi_env env, const ist_plan* p) {
  napi_value o, rects;
  napi_create_object(env, &o);
  set_num(env, o, "outW", p->out_w); set_num(env, o, "outH", p->out_h);
  set_num(env, o, "scaleDown", p->scale_down); set_num(env, o, "superSample", p->super_sample);
  set_num(env, o, "canvasW", (double)p->canvas_w); set_num(env, o, "canvasH", (double)p->canvas_h);
  napi_value b; napi_get_boolean(env, p->big_task != 0, &b); napi_set_named_property(env, o, "bigTask", b);
  napi_create_array_with_length(env, (size_t)p->n_rects, &rects);
  for (int i = 0; i < p->n_rects; i++) {
    napi_value r; napi_create_object(env, &r);
    set_num(env, r, "image", p->rects[i].image); set_num(env, r, "orientation", p->rects[i].orientation);
    set_num(env, r, "dx", p->rects[i].dx); set_num(env, r, "dy", p->rects[i].dy);
    set_num(env, r, "dw", p->rects[i].dw); set_num(env, r, "dh", p->rects[i].dh);
    napi_set_element(env, rects, (uint32_t)i, r);
  }
  napi_set_named_property(env, o, "rects", rects);
  return o;
}

/* plan(images, direction, mode, gap, limits) */
static napi_value js_plan(napi_env env, napi_callback_info info) {
  size_t argc = 5; napi_value argv[5];
  CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 4) { napi_throw_type_error(env, NULL, "plan(images, direction, mode, gap, limits)"); return NULL; }
  images_t im;
  if (!images_parse(env, argv[0], &im, 0)) { images_free(env, &im); return NULL; }
  int32_t direction = 0, mode = 0; double gap = 0; ist_limits lim;
  napi_get_value_int32(env, argv[1], &direction);
  napi_get_value_int32(env, argv[2], &mode);
  napi_get_value_double(env, argv[3], &gap);
  limits_parse(env, argc > 4 ? argv[4] : argv[3], &lim);
  if (argc <= 4) ist_limits_unlimited(&lim);
  ist_plan p;
  const int rc = ist_plan_compute(im.descs, im.n, direction, mode, gap, &lim, &p);
  images_free(env, &im);
  if (rc < 0) return throw_ist(env, rc);
  if (rc == IST_NOTHING_TO_DO) { napi_value u; napi_get_null(env, &u); return u; }
  napi_value out = plan_to_js(env, &p);
  ist_plan_free(&p);
  return out;
}

typedef struct {
  images_t im;
  int direction, mode, filter; double gap; ist_limits lim;
  int want_png; int64_t png_len;          /* stitchPng: `pixels` holds the PNG file bytes */
  int devices[64]; int ndev, split;       /* opts.devices / opts.split: ndev 0 = the process-wide single context */
  ist_plan plan; uint8_t* pixels; int rc; char err[256];
  napi_deferred deferred; napi_async_work work;
} stitch_job;

static void free_pixels(napi_env env, void* data, void* hint) { (void)env; (void)hint; ist_free(data); }

static napi_value stitch_result(napi_env env, stitch_job* j) {
  napi_value o, buf;
  napi_create_object(env, &o);
  const size_t bytes = j->want_png ? (size_t)j->png_len : (size_t)j->plan.canvas_w * (size_t)j->plan.canvas_h * 4;
  if (napi_create_external_buffer(env, bytes, j->pixels, free_pixels, NULL, &buf) != napi_ok) { ist_free(j->pixels); return NULL; }
  set_num(env, o, "width", (double)j->plan.canvas_w);
  set_num(env, o, "height", (double)j->plan.canvas_h);
  napi_set_named_property(env, o, j->want_png ? "png" : "data", buf);
  napi_set_named_property(env, o, "plan", plan_to_js(env, &j->plan));
  ist_plan_free(&j->plan);
  return o;
}

static void stitch_execute(napi_env env, void* data) {
  (void)env;
  stitch_job* j = (stitch_job*)data;
  ist_ctx* ctx = get_ctx();
  if (!ctx) { j->rc = IST_E_NO_DEVICE; snprintf(j->err, sizeof j->err, "%s", g_ctx_err); return; }
  for (int i = 0; i < j->im.n; i++)
    if (!j->im.data[i]) { j->rc = IST_E_DECODE; snprintf(j->err, sizeof j->err, "\xe5\x9b\xbe\xe7\x89\x87%d\xe8\xa7\xa3\xe7\xa0\x81\xe5\xbc\x82\xe5\xb8\xb8", i); return; }
  if (j->ndev > 0 && !j->want_png)
    j->rc = ist_stitch_rgba8_multi(j->devices, j->ndev, j->im.descs, j->im.data, j->im.pitch, j->im.n, j->direction, j->mode, j->gap, &j->lim,
                                   j->filter, j->split, &j->plan, &j->pixels);
  else if (j->want_png)
    j->rc = ist_stitch_png(ctx, j->im.descs, j->im.data, j->im.pitch, j->im.n, j->direction, j->mode, j->gap, &j->lim,
                           j->filter, &j->plan, &j->pixels, &j->png_len);
  else
    j->rc = ist_stitch_rgba8(ctx, j->im.descs, j->im.data, j->im.pitch, j->im.n, j->direction, j->mode, j->gap, &j->lim,
                             j->filter, &j->plan, &j->pixels);
  if (j->rc < 0) snprintf(j->err, sizeof j->err, "%s", ist_last_error());
}

static napi_value make_error(napi_env env, int code, const char* why) {
  char msg[512], codebuf[16];
  strcpy(msg, "\xe6\x8b\xbc\xe5\x9b\xbe\xe5\xa4\xb1\xe8\xb4\xa5\xef\xbc\x9a");
  strncat(msg, why && *why ? why : "unknown", sizeof(msg) - strlen(msg) - 1);
  snprintf(codebuf, sizeof codebuf, "%d", code);
  napi_value m, c, e;
  napi_create_string_utf8(env, msg, NAPI_AUTO_LENGTH, &m);
  napi_create_string_utf8(env, codebuf, NAPI_AUTO_LENGTH, &c);
  napi_create_error(env, c, m, &e);
  return e;
}

static void stitch_complete(napi_env env, napi_status status, void* data) {
  stitch_job* j = (stitch_job*)data;
  (void)status;
  if (j->rc < 0) napi_reject_deferred(env, j->deferred, make_error(env, j->rc, j->err));
  else if (j->rc == IST_NOTHING_TO_DO) { napi_value u; napi_get_null(env, &u); napi_resolve_deferred(env, j->deferred, u); }
  else {
    napi_value r = stitch_result(env, j);
    if (r) napi_resolve_deferred(env, j->deferred, r);
    else napi_reject_deferred(env, j->deferred, make_error(env, IST_E_NOMEM, "could not wrap the output buffer"));
  }
  napi_delete_async_work(env, j->work);
  images_free(env, &j->im);
  free(j);
}

static stitch_job* stitch_parse(napi_env env, napi_callback_info info, int want_refs) {
  size_t argc = 9; napi_value argv[9];
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < 6) {
    napi_throw_type_error(env, NULL, "stitch(images, direction, mode, gap, limits, filter)");
    return NULL;
  }
  stitch_job* j = (stitch_job*)calloc(1, sizeof *j);
  if (!images_parse(env, argv[0], &j->im, want_refs)) { images_free(env, &j->im); free(j); return NULL; }
  int32_t v = 0;
  napi_get_value_int32(env, argv[1], &v); j->direction = v;
  napi_get_value_int32(env, argv[2], &v); j->mode = v;
  napi_get_value_double(env, argv[3], &j->gap);
  limits_parse(env, argv[4], &j->lim);
  napi_get_value_int32(env, argv[5], &v); j->filter = v;
  if (argc > 6) { bool b = false; napi_get_value_bool(env, argv[6], &b); j->want_png = b ? 1 : 0; }
  if (argc > 7) {                                       /* devices: number[] */
    bool is_arr = false; uint32_t n = 0;
    if (napi_is_array(env, argv[7], &is_arr) == napi_ok && is_arr) {
      napi_get_array_length(env, argv[7], &n);
      if (n > 64) { napi_throw_range_error(env, NULL, "devices: at most 64 entries"); images_free(env, &j->im); free(j); return NULL; }
      for (uint32_t i = 0; i < n; i++) { napi_value e; int32_t d = -1; napi_get_element(env, argv[7], i, &e); napi_get_value_int32(env, e, &d); j->devices[i] = d; }
      j->ndev = (int)n;
    }
  }
  if (argc > 8) { napi_get_value_int32(env, argv[8], &v); j->split = v; }
  return j;
}

/* stitch(...) -> Promise: runs on the libuv pool, the event loop keeps turning (the reference yields between images
 * with `await _sleep(0)`, index.js:1567) */
static napi_value js_stitch(napi_env env, napi_callback_info info) {
  stitch_job* j = stitch_parse(env, info, 1);
  if (!j) return NULL;
  napi_value promise, name;
  CHECK(napi_create_promise(env, &j->deferred, &promise));
  napi_create_string_utf8(env, "imagestitch.stitch", NAPI_AUTO_LENGTH, &name);
  CHECK(napi_create_async_work(env, NULL, name, stitch_execute, stitch_complete, j, &j->work));
  CHECK(napi_queue_async_work(env, j->work));
  return promise;
}

/* stitchFiles(files: Buffer[], direction, mode, gap, limits, filter) -> Promise<{width,height,png,plan}>
 * the device-resident pipeline (ist_stitch_files_png): only file bytes go in, only PNG bytes come out */
typedef struct {
  int n; const uint8_t** files; int64_t* lens; napi_ref* refs;
  int direction, mode, filter; double gap; ist_limits lim;
  ist_plan plan; uint8_t* png; int64_t png_len; int rc; char err[256];
  napi_deferred deferred; napi_async_work work;
} files_job;

static void files_execute(napi_env env, void* data) {
  (void)env;
  files_job* j = (files_job*)data;
  ist_ctx* ctx = get_ctx();
  if (!ctx) { j->rc = IST_E_NO_DEVICE; snprintf(j->err, sizeof j->err, "%s", g_ctx_err); return; }
  j->rc = ist_stitch_files_png(ctx, j->files, j->lens, j->n, j->direction, j->mode, j->gap, &j->lim, j->filter, &j->plan, &j->png, &j->png_len);
  if (j->rc < 0) snprintf(j->err, sizeof j->err, "%s", ist_last_error());
}

static void files_complete(napi_env env, napi_status status, void* data) {
  files_job* j = (files_job*)data;
  (void)status;
  if (j->rc < 0) napi_reject_deferred(env, j->deferred, make_error(env, j->rc, j->err));
  else if (j->rc == IST_NOTHING_TO_DO) { napi_value u; napi_get_null(env, &u); napi_resolve_deferred(env, j->deferred, u); }
  else {
    napi_value o, buf;
    napi_create_object(env, &o);
    if (napi_create_external_buffer(env, (size_t)j->png_len, j->png, free_pixels, NULL, &buf) != napi_ok) {
      ist_free(j->png); napi_reject_deferred(env, j->deferred, make_error(env, IST_E_NOMEM, "could not wrap the PNG buffer"));
    } else {
      set_num(env, o, "width", (double)j->plan.canvas_w); set_num(env, o, "height", (double)j->plan.canvas_h);
      napi_set_named_property(env, o, "png", buf);
      napi_set_named_property(env, o, "plan", plan_to_js(env, &j->plan));
      napi_resolve_deferred(env, j->deferred, o);
    }
    ist_plan_free(&j->plan);
  }
  napi_delete_async_work(env, j->work);
  for (int i = 0; i < j->n; i++) if (j->refs[i]) napi_delete_reference(env, j->refs[i]);
  free(j->files); free(j->lens); free(j->refs); free(j);
}

static napi_value js_stitch_files(napi_env env, napi_callback_info info) {
  size_t argc = 6; napi_value argv[6];
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < 6) { napi_throw_type_error(env, NULL, "stitchFiles(files, direction, mode, gap, limits, filter)"); return NULL; }
  bool is_arr = false; uint32_t n = 0;
  if (napi_is_array(env, argv[0], &is_arr) != napi_ok || !is_arr) { napi_throw_type_error(env, NULL, "files must be an array of Buffers"); return NULL; }
  napi_get_array_length(env, argv[0], &n);
  files_job* j = (files_job*)calloc(1, sizeof *j);
  j->n = (int)n;
  j->files = (const uint8_t**)calloc(n ? n : 1, sizeof(uint8_t*)); j->lens = (int64_t*)calloc(n ? n : 1, sizeof(int64_t)); j->refs = (napi_ref*)calloc(n ? n : 1, sizeof(napi_ref));
  for (uint32_t i = 0; i < n; i++) {
    napi_value e; void* p = NULL; size_t len = 0; bool isbuf = false, ta = false;
    napi_get_element(env, argv[0], i, &e);
    napi_is_buffer(env, e, &isbuf); napi_is_typedarray(env, e, &ta);
    if (isbuf) napi_get_buffer_info(env, e, &p, &len);
    else if (ta) { napi_typedarray_type tt; napi_value ab; size_t off; napi_get_typedarray_info(env, e, &tt, &len, &p, &ab, &off); }
    if (!p) { napi_throw_type_error(env, NULL, "files[i] must be a Buffer / Uint8Array"); for (uint32_t k = 0; k < i; k++) napi_delete_reference(env, j->refs[k]); free(j->files); free(j->lens); free(j->refs); free(j); return NULL; }
    j->files[i] = (const uint8_t*)p; j->lens[i] = (int64_t)len;
    napi_create_reference(env, e, 1, &j->refs[i]);
  }
  int32_t v = 0;
  napi_get_value_int32(env, argv[1], &v); j->direction = v;
  napi_get_value_int32(env, argv[2], &v); j->mode = v;
  napi_get_value_double(env, argv[3], &j->gap);
  limits_parse(env, argv[4], &j->lim);
  napi_get_value_int32(env, argv[5], &v); j->filter = v;
  napi_value promise, name;
  CHECK(napi_create_promise(env, &j->deferred, &promise));
  napi_create_string_utf8(env, "imagestitch.stitchFiles", NAPI_AUTO_LENGTH, &name);
  CHECK(napi_create_async_work(env, NULL, name, files_execute, files_complete, j, &j->work));
  CHECK(napi_queue_async_work(env, j->work));
  return promise;
}

static napi_value js_stitch_sync(napi_env env, napi_callback_info info) {
  stitch_job* j = stitch_parse(env, info, 0);
  if (!j) return NULL;
  stitch_execute(env, j);
  napi_value out = NULL;
  if (j->rc < 0) { napi_throw(env, make_error(env, j->rc, j->err)); }
  else if (j->rc == IST_NOTHING_TO_DO) napi_get_null(env, &out);
  else out = stitch_result(env, j);
  images_free(env, &j->im);
  free(j);
  return out;
}

/* render(canvasW, canvasH, clearRGBA(Uint8Array 4), ops(Float64Array 18/op), images, filter, region|null) -> Buffer */
static napi_value js_render(napi_env env, napi_callback_info info) {
  size_t argc = 8; napi_value argv[8];
  CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 6) { napi_throw_type_error(env, NULL, "render(canvasW, canvasH, clear, ops, images, filter, region)"); return NULL; }
  double cw = 0, ch = 0; int32_t filter = 1;
  napi_get_value_double(env, argv[0], &cw);
  napi_get_value_double(env, argv[1], &ch);
  uint8_t clear[4] = {0, 0, 0, 0};
  { bool ta = false; napi_is_typedarray(env, argv[2], &ta);
    if (ta) { napi_typedarray_type tt; size_t len; void* p; napi_value ab; size_t off;
      napi_get_typedarray_info(env, argv[2], &tt, &len, &p, &ab, &off);
      if (len >= 4 && p) memcpy(clear, p, 4); } }
  napi_typedarray_type tt; size_t len = 0; void* p = NULL; napi_value ab; size_t off;
  bool ta = false; napi_is_typedarray(env, argv[3], &ta);
  if (!ta) { napi_throw_type_error(env, NULL, "ops must be a Float64Array"); return NULL; }
  napi_get_typedarray_info(env, argv[3], &tt, &len, &p, &ab, &off);
  if (tt != napi_float64_array || len % 18) { napi_throw_type_error(env, NULL, "ops must be a Float64Array with 18 doubles per op"); return NULL; }
  const int n_ops = (int)(len / 18);
  ist_op* ops = (ist_op*)calloc(n_ops ? n_ops : 1, sizeof(ist_op));
  const double* q = (const double*)p;
  for (int i = 0; i < n_ops; i++, q += 18) {
    ops[i].kind = (int32_t)q[0]; ops[i].image = (int32_t)q[1];
    memcpy(ops[i].m, q + 2, 6 * sizeof(double));
    memcpy(ops[i].s, q + 8, 4 * sizeof(double));
    memcpy(ops[i].d, q + 12, 4 * sizeof(double));
    const uint32_t c = (uint32_t)q[16];
    ops[i].rgba[0] = c & 255; ops[i].rgba[1] = (c >> 8) & 255; ops[i].rgba[2] = (c >> 16) & 255; ops[i].rgba[3] = (c >> 24) & 255;
  }
  images_t im;
  if (!images_parse(env, argv[4], &im, 0)) { images_free(env, &im); free(ops); return NULL; }
  napi_get_value_int32(env, argv[5], &filter);
  ist_region reg = {0, 0, (int32_t)cw, (int32_t)ch};
  int have_region = 0;
  if (argc > 6) {
    napi_valuetype t; napi_typeof(env, argv[6], &t);
    if (t == napi_object) {
      int64_t v;
      if (get_named_i64(env, argv[6], "x", &v)) reg.x = (int32_t)v;
      if (get_named_i64(env, argv[6], "y", &v)) reg.y = (int32_t)v;
      if (get_named_i64(env, argv[6], "w", &v)) reg.w = (int32_t)v;
      if (get_named_i64(env, argv[6], "h", &v)) reg.h = (int32_t)v;
      have_region = 1;
    }
  }
  if (reg.w < 1 || reg.h < 1) { images_free(env, &im); free(ops); napi_throw_range_error(env, NULL, "empty region"); return NULL; }
  if (argc > 7) {                 /* asPng: wx.canvasToTempFilePath({fileType:'png'}) - the canvas stays on the device */
    bool as_png = false; napi_get_value_bool(env, argv[7], &as_png);
    if (as_png) {
      uint8_t* png = NULL; int64_t len = 0;
      ist_ctx* c2 = get_ctx();
      int rc2 = c2 ? ist_render_png(c2, (int64_t)cw, (int64_t)ch, clear, ops, n_ops, im.descs, im.data, im.pitch, im.n, filter, &png, &len)
                   : IST_E_NO_DEVICE;
      images_free(env, &im); free(ops);
      if (rc2 == IST_E_NO_DEVICE && !c2) { napi_throw(env, make_error(env, rc2, g_ctx_err)); return NULL; }
      if (rc2 < 0) return throw_ist(env, rc2);
      napi_value buf;
      if (napi_create_external_buffer(env, (size_t)len, png, free_pixels, NULL, &buf) != napi_ok) { ist_free(png); napi_throw_error(env, NULL, "out of memory"); return NULL; }
      return buf;
    }
  }
  const size_t bytes = (size_t)reg.w * (size_t)reg.h * 4;
  void* out_data = NULL; napi_value out;
  if (napi_create_buffer(env, bytes, &out_data, &out) != napi_ok) { images_free(env, &im); free(ops); napi_throw_error(env, NULL, "out of memory"); return NULL; }
  ist_ctx* ctx = get_ctx();
  int rc = ctx ? ist_render_rgba8(ctx, (int64_t)cw, (int64_t)ch, clear, ops, n_ops, im.descs, im.data, im.pitch, im.n, filter,
                                  have_region ? &reg : NULL, (uint8_t*)out_data, (size_t)reg.w * 4)
               : IST_E_NO_DEVICE;
  images_free(env, &im); free(ops);
  if (rc == IST_E_NO_DEVICE && !ctx) { napi_throw(env, make_error(env, rc, g_ctx_err)); return NULL; }
  if (rc < 0) return throw_ist(env, rc);
  return out;
}

/* encodePng(data: Uint8Array RGBA, width, height) -> Buffer (PNG file bytes) */
static napi_value js_encode_png(napi_env env, napi_callback_info info) {
  size_t argc = 3; napi_value argv[3];
  CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  if (argc < 3) { napi_throw_type_error(env, NULL, "encodePng(data, width, height)"); return NULL; }
  bool ta = false, isbuf = false; void* p = NULL; size_t len = 0;
  napi_is_buffer(env, argv[0], &isbuf); napi_is_typedarray(env, argv[0], &ta);
  if (isbuf) napi_get_buffer_info(env, argv[0], &p, &len);
  else if (ta) { napi_typedarray_type tt; napi_value ab; size_t off; napi_get_typedarray_info(env, argv[0], &tt, &len, &p, &ab, &off); }
  double w = 0, h = 0;
  napi_get_value_double(env, argv[1], &w); napi_get_value_double(env, argv[2], &h);
  if (!p || w < 1 || h < 1 || (double)len < w * h * 4) { napi_throw_range_error(env, NULL, "data is smaller than width*height*4"); return NULL; }
  ist_ctx* ctx = get_ctx();
  if (!ctx) { napi_throw(env, make_error(env, IST_E_NO_DEVICE, g_ctx_err)); return NULL; }
  uint8_t* png = NULL; int64_t n = 0;
  const int rc = ist_png_encode_rgba8(ctx, (const uint8_t*)p, (size_t)w * 4, (int64_t)w, (int64_t)h, &png, &n);
  if (rc < 0) return throw_ist(env, rc);
  napi_value buf;
  if (napi_create_external_buffer(env, (size_t)n, png, free_pixels, NULL, &buf) != napi_ok) { ist_free(png); napi_throw_error(env, NULL, "out of memory"); return NULL; }
  return buf;
}

/* decodeImage(file: Buffer) -> {width, height, orientation, data}: PNG (host) or JPEG (host Huffman + GPU reconstruction) */
static napi_value js_decode_image(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  bool ta = false, isbuf = false; void* p = NULL; size_t len = 0;
  if (argc < 1) { napi_throw_type_error(env, NULL, "decodeImage(buffer)"); return NULL; }
  napi_is_buffer(env, argv[0], &isbuf); napi_is_typedarray(env, argv[0], &ta);
  if (isbuf) napi_get_buffer_info(env, argv[0], &p, &len);
  else if (ta) { napi_typedarray_type tt; napi_value ab; size_t off; napi_get_typedarray_info(env, argv[0], &tt, &len, &p, &ab, &off); }
  if (!p) { napi_throw_type_error(env, NULL, "decodeImage expects a Buffer / Uint8Array"); return NULL; }
  int32_t w = 0, h = 0, orient = 0;
  int rc = ist_image_info((const uint8_t*)p, (int64_t)len, &w, &h, &orient);
  if (rc < 0) return throw_ist(env, rc);
  const int is_jpeg = len >= 2 && ((const uint8_t*)p)[0] == 0xFF && ((const uint8_t*)p)[1] == 0xD8;
  ist_ctx* ctx = is_jpeg ? get_ctx() : NULL;
  if (is_jpeg && !ctx) { napi_throw(env, make_error(env, IST_E_NO_DEVICE, g_ctx_err)); return NULL; }
  void* out_data = NULL; napi_value buf, o;
  if (napi_create_buffer(env, (size_t)w * (size_t)h * 4, &out_data, &buf) != napi_ok) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
  rc = ist_image_decode_rgba8(ctx, (const uint8_t*)p, (int64_t)len, (uint8_t*)out_data, (size_t)w * 4, (int64_t)h);
  if (rc < 0) return throw_ist(env, rc);
  napi_create_object(env, &o);
  set_num(env, o, "width", w); set_num(env, o, "height", h); set_num(env, o, "orientation", orient ? orient : 1);
  napi_value op; napi_get_boolean(env, is_jpeg != 0, &op); napi_set_named_property(env, o, "opaque", op);
  napi_set_named_property(env, o, "data", buf);
  return o;
}

/* decodePng(file: Buffer) -> {width, height, data: Buffer RGBA8}   (host decode, no GPU needed) */
static napi_value js_decode_png(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  bool ta = false, isbuf = false; void* p = NULL; size_t len = 0;
  if (argc < 1) { napi_throw_type_error(env, NULL, "decodePng(buffer)"); return NULL; }
  napi_is_buffer(env, argv[0], &isbuf); napi_is_typedarray(env, argv[0], &ta);
  if (isbuf) napi_get_buffer_info(env, argv[0], &p, &len);
  else if (ta) { napi_typedarray_type tt; napi_value ab; size_t off; napi_get_typedarray_info(env, argv[0], &tt, &len, &p, &ab, &off); }
  if (!p) { napi_throw_type_error(env, NULL, "decodePng expects a Buffer / Uint8Array"); return NULL; }
  int32_t w = 0, h = 0;
  int rc = ist_png_info((const uint8_t*)p, (int64_t)len, &w, &h);
  if (rc < 0) return throw_ist(env, rc);
  void* out_data = NULL; napi_value buf, o;
  if (napi_create_buffer(env, (size_t)w * (size_t)h * 4, &out_data, &buf) != napi_ok) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
  rc = ist_png_decode_rgba8((const uint8_t*)p, (int64_t)len, (uint8_t*)out_data, (size_t)w * 4, (int64_t)h);
  if (rc < 0) return throw_ist(env, rc);
  napi_create_object(env, &o);
  set_num(env, o, "width", w); set_num(env, o, "height", h);
  napi_set_named_property(env, o, "data", buf);
  return o;
}

static napi_value js_device_count(napi_env env, napi_callback_info info) {
  (void)info; napi_value v; napi_create_int32(env, ist_device_count(), &v); return v;
}
static napi_value js_last_error(napi_env env, napi_callback_info info) {
  (void)info; napi_value v; napi_create_string_utf8(env, ist_last_error(), NAPI_AUTO_LENGTH, &v); return v;
}
static napi_value js_abi_version(napi_env env, napi_callback_info info) {
  (void)info; napi_value v; napi_create_int32(env, ist_abi_version(), &v); return v;
}

/* setPngLevel(level): 0 stored deflate blocks, 1 compressed on the GPU (ist_ctx_set_png_level) - applies to every PNG
 * this process exports afterwards */
static napi_value js_set_png_level(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  int32_t level = 0;
  if (argc < 1 || napi_get_value_int32(env, argv[0], &level) != napi_ok) { napi_throw_type_error(env, NULL, "setPngLevel(level)"); return NULL; }
  ist_ctx* ctx = get_ctx();
  if (!ctx) { napi_throw(env, make_error(env, IST_E_NO_DEVICE, g_ctx_err)); return NULL; }
  const int rc = ist_ctx_set_png_level(ctx, level);
  if (rc < 0) return throw_ist(env, rc);
  napi_value v; napi_get_undefined(env, &v); return v;
}

/* environment teardown: nothing of the library may still be in flight when the HIP runtime shuts down (include/imagestitch.h,
 * ist_ctx_sync); the idle pinned result blocks go back to the system */
static void on_env_cleanup(void* arg) {
  (void)arg;
  if (g_ctx) (void)ist_ctx_sync(g_ctx);
  ist_pool_trim();
}

static napi_value init(napi_env env, napi_value exports) {
  napi_add_env_cleanup_hook(env, on_env_cleanup, NULL);
  napi_property_descriptor props[] = {
      {"plan", NULL, js_plan, NULL, NULL, NULL, napi_default, NULL},
      {"stitch", NULL, js_stitch, NULL, NULL, NULL, napi_default, NULL},
      {"stitchSync", NULL, js_stitch_sync, NULL, NULL, NULL, napi_default, NULL},
      {"stitchFiles", NULL, js_stitch_files, NULL, NULL, NULL, napi_default, NULL},
      {"render", NULL, js_render, NULL, NULL, NULL, napi_default, NULL},
      {"encodePng", NULL, js_encode_png, NULL, NULL, NULL, napi_default, NULL},
      {"setPngLevel", NULL, js_set_png_level, NULL, NULL, NULL, napi_default, NULL},
      {"decodePng", NULL, js_decode_png, NULL, NULL, NULL, napi_default, NULL},
      {"decodeImage", NULL, js_decode_image, NULL, NULL, NULL, napi_default, NULL},
      {"deviceCount", NULL, js_device_count, NULL, NULL, NULL, napi_default, NULL},
      {"lastError", NULL, js_last_error, NULL, NULL, NULL, napi_default, NULL},
      {"abiVersion", NULL, js_abi_version, NULL, NULL, NULL, napi_default, NULL},
  };
  napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
