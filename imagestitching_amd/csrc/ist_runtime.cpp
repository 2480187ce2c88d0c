// ist_runtime.cpp — device context, compiled jobs and the host-buffer convenience path of the C-ABI.
//
// Reference anchors (miniprogram-stitch/miniprogram/): the context stands for the canvas node obtained at
// pages/index/index.js:1196-1204; a job for the offscreen canvas + recorded draws (utils/canvas.js:131-150,
// index.js:1391-1428, 1532-1551); launch for the raster flush the export forces (utils/canvas.js:205-242).
// There is deliberately no CPU fallback: without a HIP device every rendering entry point fails.
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <cerrno>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "ist_ctx.h"
#include "ist_internal.h"
#include "ist_jpeg.h"
#include "ist_launch.h"
#include "ist_webp.h"

using namespace ist;

#define IST_HIP(expr)                                                                                       \
  do {                                                                                                      \
    const hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) return fail(IST_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
  } while (0)

namespace ist {
int ctx_png_level(const ist_ctx* ctx) { return ctx ? ctx->png_level : 0; }
int ctx_png_scratch(ist_ctx* ctx, size_t need, void** p) {
  const int rc = grow_device(&ctx->scratch_png, &ctx->scratch_png_bytes, need);
  *p = ctx->scratch_png;
  return rc;
}

static std::atomic<int64_t> g_dev_allocs{0};
static std::atomic<int64_t> g_gpu_entropy_files{0};
static std::atomic<int64_t> g_direct_images{0};
int dev_malloc(void** p, size_t bytes) {
  g_dev_allocs.fetch_add(1, std::memory_order_relaxed);
  return static_cast<int>(hipMalloc(p, bytes));
}
void dev_free(void* p) { if (p) (void)hipFree(p); }

int grow_device(void** p, size_t* have, size_t need) {
  if (*have >= need) return IST_OK;
  if (*p) { dev_free(*p); *p = nullptr; *have = 0; }
  if (dev_malloc(p, need) != 0) { (void)hipGetLastError(); return fail(IST_E_NOMEM, "out of device memory (" + std::to_string(need >> 20) + " MiB)"); }
  *have = need;
  return IST_OK;
}
}  // namespace ist

namespace {

Stager& stager_of(ist_ctx* ctx) {
  if (!ctx->stager) ctx->stager.reset(new Stager(ctx->device));
  return *ctx->stager;
}

// A buffer the library hands to the caller (freed with ist_free): a pinned block from the pool, filled by ONE linear DMA
// from device memory, no host copy.  Synchronises `stream`.
int read_back_pooled(const void* dev, size_t bytes, hipStream_t stream, uint8_t** out) {
  uint8_t* host = static_cast<uint8_t*>(pool_take(bytes));
  if (!host) return fail(IST_E_NOMEM, "out of pinned host memory for the result");
  if (hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
    (void)hipGetLastError();
    pool_give(host);
    return fail(IST_E_HIP, "result readback failed");
  }
  *out = host;
  return IST_OK;
}

// the context's second stream (high priority: its small kernels should not queue behind thousands of workgroups of the first)
int ensure_aux(ist_ctx* ctx) {
  if (ctx->aux) return IST_OK;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (tuning_mode() && std::getenv("IST_AUX_PRIORITY")) hi = std::atoi(std::getenv("IST_AUX_PRIORITY")) ? hi : lo;     // A/B knob: 0 = lowest
  if (hipStreamCreateWithPriority(&ctx->aux, hipStreamNonBlocking, hi) != hipSuccess) { (void)hipGetLastError(); ctx->aux = nullptr; return fail(IST_E_HIP, "hipStreamCreate failed"); }
  return IST_OK;
}

// The PNG file of a canvas in device memory -> a pooled pinned block (freed with ist_free).  `dfile` = device scratch of
// at least ist_png_bound bytes (nullptr: the context's own).  The compressing encoder hands the file over slab by slab
// while it is still compressing (png_encode_device_deflate); the stored form is encoded whole and copied once.
// Caller holds ctx->mu.  Synchronises ctx->stream.
int png_to_host(ist_ctx* ctx, const void* canvas, size_t pitch, int64_t w, int64_t h, void* dfile, uint8_t** out_png, int64_t* out_len,
                const std::function<int(int64_t, void*)>& need_rows = nullptr, int64_t slab_rows_hint = 0) {
  const int64_t cap = ist_png_bound(w, h);
  if (!dfile) {
    const int rc = grow_device(&ctx->scratch_file, &ctx->scratch_file_bytes, static_cast<size_t>(cap));
    if (rc) return rc;
    dfile = ctx->scratch_file;
  }
  int64_t len = 0;
  if (ctx->png_level > 0) {
    { const int rc = ensure_aux(ctx); if (rc) return rc; }
    uint8_t* host = static_cast<uint8_t*>(pool_take(static_cast<size_t>(cap)));
    if (!host) return fail(IST_E_NOMEM, "out of pinned host memory for the result");
    if (!ctx->png2 && hipStreamCreateWithFlags(&ctx->png2, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->png2 = nullptr; }   // (without it the slabs share one stream)
    const int rc = png_encode_device_deflate(ctx, canvas, pitch, w, h, dfile, cap, &len, ctx->stream, host, ctx->aux, need_rows, slab_rows_hint, ctx->png2);
    if (rc) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamSynchronize(ctx->stream); if (ctx->png2) (void)hipStreamSynchronize(ctx->png2); pool_give(host); return rc; }
    *out_png = host; *out_len = len;
    return IST_OK;
  }
  int rc = need_rows ? need_rows(h, ctx->stream) : IST_OK;   // (the stored form reads the whole canvas in one pass)
  if (rc) return rc;
  rc = ist_png_encode_device(ctx, canvas, pitch, w, h, dfile, cap, &len, ctx->stream);
  if (rc) return rc;
  uint8_t* host = nullptr;
  rc = read_back_pooled(dfile, static_cast<size_t>(len), ctx->stream, &host);
  if (rc) return rc;
  *out_png = host; *out_len = len;
  return IST_OK;
}

}  // namespace

extern "C" {

int64_t ist_debug_device_allocs(void) { return g_dev_allocs.load(std::memory_order_relaxed); }
int64_t ist_debug_gpu_entropy_files(void) { return g_gpu_entropy_files.load(std::memory_order_relaxed); }
int64_t ist_debug_direct_images(void) { return g_direct_images.load(std::memory_order_relaxed); }

int ist_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

ist_ctx* ist_ctx_create(int device) {
  const int n = ist_device_count();
  if (n <= 0) { fail(IST_E_NO_DEVICE, "no HIP device: the stitch path has no CPU fallback"); return nullptr; }
  if (device < 0 || device >= n) { fail(IST_E_INVALID, "device index out of range"); return nullptr; }
  DeviceGuard g(device);
  if (!g.ok) { fail(IST_E_NO_DEVICE, "hipSetDevice failed"); return nullptr; }
  std::unique_ptr<ist_ctx> c(new ist_ctx);
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    fail(IST_E_HIP, "hipStreamCreate failed");
    return nullptr;
  }
  return c.release();
}

int ist_ctx_set_png_level(ist_ctx* ctx, int level) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  if (level < 0 || level > 1) return fail(IST_E_INVALID, "PNG level must be 0 (stored) or 1 (compressed)");
  ctx->png_level = level;
  return IST_OK;
}

int ist_ctx_sync(ist_ctx* ctx) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  DeviceGuard g(ctx->device);
  bool ok = hipStreamSynchronize(ctx->stream) == hipSuccess;
  if (ctx->aux) ok = (hipStreamSynchronize(ctx->aux) == hipSuccess) && ok;
  if (ctx->render) ok = (hipStreamSynchronize(ctx->render) == hipSuccess) && ok;
  if (ctx->png2) ok = (hipStreamSynchronize(ctx->png2) == hipSuccess) && ok;
  if (ctx->stager) ok = (ctx->stager->sync() == IST_OK) && ok;
  if (!ok) { (void)hipGetLastError(); return fail(IST_E_HIP, "hipStreamSynchronize failed"); }
  return IST_OK;
}

void ist_ctx_destroy(ist_ctx* ctx) {
  if (!ctx) return;
  DeviceGuard g(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);
  dev_free(ctx->scratch_src);
  dev_free(ctx->scratch_dst);
  dev_free(ctx->scratch_dec);
  dev_free(ctx->scratch_huff);
  dev_free(ctx->scratch_png);
  dev_free(ctx->scratch_file);
  dev_free(ctx->scratch_arena);
  dev_free(ctx->scratch_ent);
  for (hipStream_t st : ctx->img_stream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  for (hipEvent_t ev : ctx->img_event) if (ev) (void)hipEventDestroy(ev);
  for (void* q : ctx->img_huff) dev_free(q);
  if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
  if (ctx->render) { (void)hipStreamSynchronize(ctx->render); (void)hipStreamDestroy(ctx->render); }
  if (ctx->png2) { (void)hipStreamSynchronize(ctx->png2); (void)hipStreamDestroy(ctx->png2); }
  if (ctx->render_done) (void)hipEventDestroy(ctx->render_done);
  for (const ist_ctx::TableBlock& b : ctx->table_pool) dev_free(b.p);
  ctx->workers.reset();
  ctx->stager.reset();
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

static std::atomic<int64_t> g_flat_launches{0};
static std::atomic<int64_t> g_duplex_stitches{0};

int64_t ist_debug_flat_launches(void) { return g_flat_launches.load(); }

ist_job* ist_job_create(ist_ctx* ctx, int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4],
                        const ist_op* ops, int n_ops, const ist_image_desc* images, int n_images,
                        int filter, const ist_region* clip) {
  if (!ctx) { fail(IST_E_NO_CONTEXT, "无法获取绘图上下文"); return nullptr; }
  if (n_images > kMaxImages) { fail(IST_E_UNSUPPORTED, "more than 128 images in one launch"); return nullptr; }
  static const uint8_t transparent[4] = {0, 0, 0, 0};
  std::unique_ptr<ist_job> job(new ist_job);
  job->ctx = ctx;
  if (compile_ops(canvas_w, canvas_h, clear_rgba ? clear_rgba : transparent, ops, n_ops, images, n_images, filter,
                  clip, &job->host) != IST_OK)
    return nullptr;
  for (const DevOp& o : job->host.ops) job->max_image = std::max(job->max_image, o.image);
  job->flat = compile_flat_twin(canvas_w, canvas_h, clear_rgba ? clear_rgba : transparent, ops, n_ops, images, n_images, filter, job->host);
  DeviceGuard g(ctx->device);
  // the tables (the job's five and, when it has one, its flat twin's five) travel as ONE allocation and ONE copy (256-byte aligned sections)
  const Compiled* hs[2] = {&job->host, job->flat ? &job->flat->host : nullptr};
  DevTables* dts[2] = {&job->dt, job->flat ? &job->flat_dt : nullptr};
  size_t bytes[2][5], at[2][5], total = 0;
  const void* from[2][5];
  for (int t = 0; t < 2; ++t) {
    if (!hs[t]) continue;
    const Compiled& h = *hs[t];
    const size_t b[5] = {h.ops.size() * sizeof(DevOp), h.cells.size() * sizeof(DevCell), h.bands.size() * sizeof(DevBand),
                         h.stacks.size() * sizeof(int32_t), h.tiles.size() * sizeof(DevTile)};
    const void* f[5] = {h.ops.data(), h.cells.data(), h.bands.data(), h.stacks.data(), h.tiles.data()};
    for (int k = 0; k < 5; ++k) { bytes[t][k] = b[k]; from[t][k] = f[k]; at[t][k] = total; total += (b[k] + 255) & ~static_cast<size_t>(255); }
  }
  if (total) {
    std::vector<uint8_t> blob(total, 0);
    for (int t = 0; t < 2; ++t)
      for (int k = 0; k < 5 && hs[t]; ++k) if (bytes[t][k]) std::memcpy(blob.data() + at[t][k], from[t][k], bytes[t][k]);
    {                                       // a block of an earlier job of this context, if one is large enough
      std::lock_guard<std::mutex> lk(ctx->table_mu);
      for (size_t k = 0; k < ctx->table_pool.size(); ++k)
        if (ctx->table_pool[k].bytes >= total && ctx->table_pool[k].bytes <= 4 * total + (1u << 20)) {
          job->d_tables = ctx->table_pool[k].p; job->d_tables_bytes = ctx->table_pool[k].bytes;
          ctx->table_pool.erase(ctx->table_pool.begin() + static_cast<std::ptrdiff_t>(k));
          break;
        }
    }
    if (!job->d_tables && dev_malloc(reinterpret_cast<void**>(&job->d_tables), total) == hipSuccess) job->d_tables_bytes = total;
    if (!job->d_tables ||
        hipMemcpy(job->d_tables, blob.data(), total, hipMemcpyHostToDevice) != hipSuccess) {      // blocking: the tables are in place when this returns
      (void)hipGetLastError();
      fail(IST_E_HIP, "uploading the op tables failed");
      ist_job_destroy(job.release());
      return nullptr;
    }
    for (int t = 0; t < 2; ++t) {
      if (!hs[t]) continue;
      dts[t]->ops = bytes[t][0] ? reinterpret_cast<DevOp*>(job->d_tables + at[t][0]) : nullptr;
      dts[t]->cells = bytes[t][1] ? reinterpret_cast<DevCell*>(job->d_tables + at[t][1]) : nullptr;
      dts[t]->bands = bytes[t][2] ? reinterpret_cast<DevBand*>(job->d_tables + at[t][2]) : nullptr;
      dts[t]->stacks = bytes[t][3] ? reinterpret_cast<int32_t*>(job->d_tables + at[t][3]) : nullptr;
      dts[t]->tiles = bytes[t][4] ? reinterpret_cast<DevTile*>(job->d_tables + at[t][4]) : nullptr;
    }
  }
  return job.release();
}

int ist_job_info_get(const ist_job* job, ist_job_info* out) {
  if (!job || !out) return fail(IST_E_INVALID, "ist_job_info_get: NULL argument");
  *out = job->host.info;
  return IST_OK;
}

size_t ist_job_preferred_dst_pitch(const ist_job* job) {
  if (!job) return 0;
  const size_t row = static_cast<size_t>(job->host.rx1 - job->host.rx0) * 4;      // (a clipped job renders into a buffer as wide as its region)
  return job->flat ? row : (row + 4095) & ~static_cast<size_t>(4095);
}

int ist_job_launch(ist_job* job, const void* const* src, const size_t* src_pitch, int n_images, void* dst,
                   size_t dst_pitch, void* stream) {
  if (!job || !dst) return fail(IST_E_INVALID, "ist_job_launch: NULL argument");
  if (n_images <= job->max_image) return fail(IST_E_DECODE, "图片" + std::to_string(job->max_image) + "解码异常: source table too short");
  if (n_images > kMaxImages) return fail(IST_E_UNSUPPORTED, "more than 128 images in one launch");
  const Compiled& h = job->host;
  // only the rendered region is ever addressed, so a compact band buffer (pitch = region width) is legal when the
  // caller biases dst by -(ry0*pitch + rx0*4)
  if (dst_pitch < static_cast<size_t>(h.rx1 - h.rx0) * 4 || (dst_pitch & 3)) return fail(IST_E_INVALID, "dst_pitch too small or not a multiple of 4");
  LaunchArgs a;
  std::memset(&a, 0, sizeof(a));
  a.dst = static_cast<uint8_t*>(dst);
  a.dst_pitch = dst_pitch;
  bool dense = job->flat && dst_pitch == static_cast<size_t>(h.canvas_w) * 4;      // (a flat twin exists only for draws as wide as the canvas)
  for (const DevOp& o : h.ops) {
    if (o.image < 0) continue;
    const int i = o.image;
    if (!src || !src[i]) return fail(IST_E_DECODE, "图片" + std::to_string(i) + "解码异常");
    const size_t p = src_pitch ? src_pitch[i] : static_cast<size_t>(h.img_w[i]) * 4;
    if (p < static_cast<size_t>(h.img_w[i]) * 4 || (p & 3)) return fail(IST_E_INVALID, "src_pitch too small or not a multiple of 4");
    if ((reinterpret_cast<uintptr_t>(src[i]) & 3) != 0) return fail(IST_E_INVALID, "source pixels must be 4-byte aligned");
    a.src[i] = static_cast<const uint8_t*>(src[i]);
    a.pitch[i] = p;
    dense = dense && p == dst_pitch;
  }
  if ((reinterpret_cast<uintptr_t>(dst) & 3) != 0) return fail(IST_E_INVALID, "dst must be 4-byte aligned");
  // dense rows on both sides: the job's flat twin moves the same bytes as rows of kFlatPitch (validated above on the caller's own table)
  const Compiled& run = dense ? job->flat->host : h;
  const DevTables& dt = dense ? job->flat_dt : job->dt;
  if (dense) {
    LaunchArgs f;
    std::memset(&f, 0, sizeof(f));
    f.dst = a.dst + job->flat->dst_delta;
    f.dst_pitch = kFlatPitch;
    for (size_t k = 0; k < job->flat->src.size(); ++k) {
      f.src[k] = a.src[job->flat->src[k].image] + job->flat->src[k].delta;
      f.pitch[k] = kFlatPitch;
    }
    a = f;
    g_flat_launches.fetch_add(1, std::memory_order_relaxed);
  }
  a.ops = dt.ops; a.cells = dt.cells; a.bands = dt.bands; a.stacks = dt.stacks;
  a.tiles = run.tiles.empty() ? nullptr : dt.tiles;
  a.n_bands = static_cast<int32_t>(run.bands.size());
  a.n_cells = static_cast<int32_t>(run.cells.size());
  a.filter = run.filter;
  a.lds_words = run.lds_words;
  a.lds_half = run.lds_half;
  a.pad_ = 0;
  DeviceGuard g(job->ctx->device);
  if (!g.ok) return fail(IST_E_NO_DEVICE, "hipSetDevice failed");
  {
    std::lock_guard<std::mutex> lk(job->launch_mu);
    job->launched = true;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int k = 0;
    while (k < job->n_launched_on && job->launched_on[k] != st) ++k;
    if (k == job->n_launched_on) {
      if (k < ist_job::kStreams) job->launched_on[job->n_launched_on++] = st;
      else job->launched_many = true;
    }
  }
  return launch_stitch(a, run.info.n_tiles, run.kernel_kind, stream);
}

void ist_job_destroy(ist_job* job) {
  if (!job) return;
  if (job->ctx) {
    DeviceGuard g(job->ctx->device);
    // the tables go to the context's pool for the next job: every launch must have read them first.  (What hipFree did
    // implicitly.  NOT an event per launch: recorded behind every kernel it cost back-to-back launches 3 % — 135 -> 140 us,
    // measured.)  Only the streams the job ran on are waited for: a host that shares the device (torch, the device group's
    // other streams) is not stalled by the death of one job.  One-shot jobs of the host-path entry points arrive here with
    // an idle stream.  A stream the caller has destroyed since makes the wait fail: then, and only then, the device is waited for.
    bool idle = true;
    if (job->launched) {
      bool per_stream = !job->launched_many;
      for (int k = 0; k < job->n_launched_on && per_stream; ++k)
        if (hipStreamSynchronize(job->launched_on[k]) != hipSuccess) { (void)hipGetLastError(); per_stream = false; }
      if (!per_stream) { idle = hipDeviceSynchronize() == hipSuccess; if (!idle) (void)hipGetLastError(); }
    }
    if (job->d_tables) {
      bool kept = false;
      if (idle) {
        std::lock_guard<std::mutex> lk(job->ctx->table_mu);
        if (static_cast<int>(job->ctx->table_pool.size()) < ist_ctx::kTablePool && job->d_tables_bytes <= (64u << 20)) {
          job->ctx->table_pool.push_back(ist_ctx::TableBlock{job->d_tables, job->d_tables_bytes});
          kept = true;
        }
      }
      if (!kept) dev_free(job->d_tables);
    }
  }
  delete job;
}

// host sources -> device scratch -> fused launch into ctx->scratch_dst (left on the device, stream NOT synchronised).
// The scratch holds exactly the rendered region, rows contiguous (the launch addresses it as if it were the canvas: dst is
// biased by the region's origin), so every readback is one linear copy.  Caller holds ctx->mu.
static int render_to_scratch(ist_ctx* ctx, int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4],
                             const ist_op* ops, int n_ops, const ist_image_desc* images, const uint8_t* const* src,
                             const size_t* src_pitch, int n_images, int filter, const ist_region* region,
                             int64_t* out_w, int64_t* out_h) {
  ist_job* job = ist_job_create(ctx, canvas_w, canvas_h, clear_rgba, ops, n_ops, images, n_images, filter, region);
  if (!job) return g_last_code ? g_last_code : IST_E_INVALID;
  struct JobFree { ist_job* j; ~JobFree() { ist_job_destroy(j); } } jf{job};

  // stage the sources that the job actually samples
  std::vector<size_t> off(static_cast<size_t>(n_images), 0);
  std::vector<char> used(static_cast<size_t>(n_images), 0);
  for (const DevOp& o : job->host.ops) if (o.image >= 0) used[o.image] = 1;
  size_t total = 0;
  for (int i = 0; i < n_images; ++i) {
    if (!used[i]) continue;
    if (!src || !src[i]) return fail(IST_E_DECODE, "图片" + std::to_string(i) + "解码异常");
    off[i] = total;
    total += (static_cast<size_t>(job->host.img_w[i]) * 4 * job->host.img_h[i] + 255) & ~static_cast<size_t>(255);
  }
  int rc = grow_device(&ctx->scratch_src, &ctx->scratch_src_bytes, total ? total : 256);
  if (rc) return rc;
  const int64_t rw = job->host.rx1 - job->host.rx0, rh = job->host.ry1 - job->host.ry0;
  const size_t pitch = static_cast<size_t>(rw) * 4;
  rc = grow_device(&ctx->scratch_dst, &ctx->scratch_dst_bytes, pitch * static_cast<size_t>(rh));
  if (rc) return rc;
  std::vector<const void*> dsrc(static_cast<size_t>(n_images), nullptr);
  std::vector<size_t> dpitch(static_cast<size_t>(n_images), 0);
  std::vector<RowsCopy> up;
  for (int i = 0; i < n_images; ++i) {
    if (!used[i]) continue;
    const size_t row = static_cast<size_t>(job->host.img_w[i]) * 4;
    const size_t hp = src_pitch ? src_pitch[i] : row;
    if (hp < row) return fail(IST_E_INVALID, "src_pitch too small");
    uint8_t* d = static_cast<uint8_t*>(ctx->scratch_src) + off[i];
    up.push_back(RowsCopy{d, src[i], nullptr, hp, row, static_cast<size_t>(job->host.img_h[i])});
    dsrc[i] = d; dpitch[i] = row;
  }
  rc = stager_of(ctx).upload(up, ctx->stream);
  if (rc) return rc;
  const uintptr_t biased = reinterpret_cast<uintptr_t>(ctx->scratch_dst) - (static_cast<uintptr_t>(job->host.ry0) * pitch + static_cast<uintptr_t>(job->host.rx0) * 4);
  rc = ist_job_launch(job, dsrc.data(), dpitch.data(), n_images, reinterpret_cast<void*>(biased), pitch, ctx->stream);
  if (rc) return rc;
  // the job's device tables are freed when `jf` goes out of scope: the launch must have consumed them
  IST_HIP(hipStreamSynchronize(ctx->stream));
  if (out_w) *out_w = rw;
  if (out_h) *out_h = rh;
  return IST_OK;
}

int ist_render_rgba8(ist_ctx* ctx, int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4],
                     const ist_op* ops, int n_ops, const ist_image_desc* images, const uint8_t* const* src,
                     const size_t* src_pitch, int n_images, int filter, const ist_region* region, uint8_t* dst,
                     size_t dst_pitch) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  if (!dst) return fail(IST_E_INVALID, "ist_render_rgba8: dst is NULL");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  // the region that will be read back (same-size export, index.js:1577-1579; or getImageData, 1564): check the caller's
  // pitch before any work is queued
  int64_t rw = canvas_w, rh = canvas_h;
  if (region) {
    const int64_t rx = std::max<int64_t>(0, region->x), ry = std::max<int64_t>(0, region->y);
    rw = std::min<int64_t>(canvas_w, static_cast<int64_t>(region->x) + region->w) - rx;
    rh = std::min<int64_t>(canvas_h, static_cast<int64_t>(region->y) + region->h) - ry;
  }
  if (rw > 0 && dst_pitch < static_cast<size_t>(rw) * 4) return fail(IST_E_INVALID, "dst_pitch too small");
  int rc = render_to_scratch(ctx, canvas_w, canvas_h, clear_rgba, ops, n_ops, images, src, src_pitch, n_images, filter, region, &rw, &rh);
  if (rc) return rc;
  std::vector<RowsCopy> down{RowsCopy{ctx->scratch_dst, nullptr, dst, dst_pitch, static_cast<size_t>(rw) * 4, static_cast<size_t>(rh)}};
  return stager_of(ctx).download(down, ctx->stream);
}

// (below, behind RowBands) the same band by band: band b + 1's source rows go up while the encoder compresses band b and sends its slabs down
static int render_png_banded(ist_ctx* ctx, int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4], const ist_op* ops, int n_ops,
                             const ist_image_desc* images, const uint8_t* const* src, const size_t* src_pitch, int n_images, int filter,
                             uint8_t** out_png, int64_t* out_len);

// PNG of a rendered op list: the canvas never leaves the device, only the PNG bytes cross PCIe
int ist_render_png(ist_ctx* ctx, int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4], const ist_op* ops,
                   int n_ops, const ist_image_desc* images, const uint8_t* const* src, const size_t* src_pitch,
                   int n_images, int filter, uint8_t** out_png, int64_t* out_len) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  if (!out_png || !out_len) return fail(IST_E_INVALID, "ist_render_png: NULL output");
  *out_png = nullptr; *out_len = 0;
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  int rc = render_png_banded(ctx, canvas_w, canvas_h, clear_rgba, ops, n_ops, images, src, src_pitch, n_images, filter, out_png, out_len);
  if (rc != 1) return rc;                      // (1: not applicable, nothing queued)
  rc = render_to_scratch(ctx, canvas_w, canvas_h, clear_rgba, ops, n_ops, images, src, src_pitch, n_images, filter, nullptr, nullptr, nullptr);
  if (rc) return rc;
  return png_to_host(ctx, ctx->scratch_dst, static_cast<size_t>(canvas_w) * 4, canvas_w, canvas_h, nullptr, out_png, out_len);
}

// plan + render + PNG: onStitch stages 2-5 including the export (index.js:1251-1581), decode excluded
int ist_stitch_png(ist_ctx* ctx, const ist_image_desc* images, const uint8_t* const* src, const size_t* src_pitch,
                   int n_images, int direction, int mode, double gap, const ist_limits* limits, int filter,
                   ist_plan* out_plan, uint8_t** out_png, int64_t* out_len) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  if (!out_plan || !out_png || !out_len) return fail(IST_E_INVALID, "ist_stitch_png: NULL output");
  ist_limits lim;
  if (limits) lim = *limits; else ist_limits_unlimited(&lim);
  int rc = ist_plan_compute(images, n_images, direction, mode, gap, &lim, out_plan);
  if (rc != IST_OK) return rc;
  std::vector<ist_op> ops(static_cast<size_t>(out_plan->n_rects) + 1);
  int n_ops = 0;
  rc = ist_plan_ops(out_plan, images, n_images, ops.data(), &n_ops);
  static const uint8_t transparent[4] = {0, 0, 0, 0};
  if (rc == IST_OK)
    rc = ist_render_png(ctx, out_plan->canvas_w, out_plan->canvas_h, transparent, ops.data(), n_ops, images, src, src_pitch,
                        n_images, filter, out_png, out_len);
  if (rc != IST_OK) ist_plan_free(out_plan);
  return rc;
}

// ---- JPEG decode: entropy decoding on the host, reconstruction on the GPU (ist_jpeg.cpp / ist_jpeg_kernels.hip) ----------
extern "C++" {
namespace {
// where one image's JPEG stages live on the device: offsets into TWO arenas - `main` (coefficient planes, quantisation
// tables, sample planes: sized from the frame header alone, so it can be laid out before any file is entropy-decoded) and
// `ent` (the sparse entries of a host-decoded sequential file: sized by the decode)
struct JpegDevLayout { size_t coef[3], q[3], plane[3], ent[3], start[3], cnt[3]; };

void jpeg_layout(const JpegImage& J, size_t* off, JpegDevLayout* L) {
  auto take = [&](size_t bytes) { const size_t at = *off; *off += (bytes + 255) & ~static_cast<size_t>(255); return at; };
  std::memset(L, 0, sizeof(*L));
  for (int c = 0; c < J.ncomp; ++c) {
    const JpegComp& C = J.comp[c];
    const size_t nblk = static_cast<size_t>(C.blocks_x) * C.blocks_y;
    L->coef[c] = take(nblk * 128);
    L->q[c] = take(128);
    L->plane[c] = take(nblk * 64);
  }
}
// the `ent` arena part of a host-decoded image (components in sparse form)
void jpeg_layout_sparse(const JpegImage& J, size_t* off, JpegDevLayout* L) {
  auto take = [&](size_t bytes) { const size_t at = *off; *off += (bytes + 255) & ~static_cast<size_t>(255); return at; };
  for (int c = 0; c < J.ncomp; ++c) {
    const JpegComp& C = J.comp[c];
    if (!C.sparse) continue;
    const size_t nblk = static_cast<size_t>(C.blocks_x) * C.blocks_y;
    L->ent[c] = take(C.ent.size() * 4 + 4); L->start[c] = take(nblk * 4); L->cnt[c] = take(nblk);
  }
}

// H2D of the coefficients (sparse entries are scattered into a zeroed plane on the GPU) + the reconstruction launches
// (the quantisation tables go to the kernels by value: no upload)
// what the reconstruction kernels need to know of an image whose coefficient planes are (or will be) in the arena at L
JpegDeviceJob jpeg_device_job(const JpegImage& J, uint8_t* d, const JpegDevLayout& L, uint8_t* d_out, size_t out_pitch) {
  JpegDeviceJob job;
  job.width = J.width; job.height = J.height; job.ncomp = J.ncomp; job.hmax = J.hmax; job.vmax = J.vmax;
  for (int c = 0; c < 3; ++c) { job.d_coef[c] = nullptr; job.q_host[c] = nullptr; job.d_plane[c] = nullptr; job.h[c] = job.v[c] = 1; job.blocks_x[c] = job.blocks_y[c] = 0; }
  for (int c = 0; c < J.ncomp; ++c) {
    const JpegComp& C = J.comp[c];
    job.d_coef[c] = reinterpret_cast<int16_t*>(d + L.coef[c]);
    job.q_host[c] = C.q;
    job.d_plane[c] = d + L.plane[c];
    job.h[c] = C.h; job.v[c] = C.v; job.blocks_x[c] = C.blocks_x; job.blocks_y[c] = C.blocks_y;
  }
  job.out = d_out; job.out_pitch = out_pitch;
  return job;
}

int jpeg_enqueue(const JpegImage& J, uint8_t* d, uint8_t* d_ent, const JpegDevLayout& L, uint8_t* d_out, size_t out_pitch, hipStream_t stream, bool coef_on_device = false,
                 bool chroma_done = false) {
  JpegDeviceJob job = jpeg_device_job(J, d, L, d_out, out_pitch);
  job.chroma_done = chroma_done;
  for (int c = 0; c < J.ncomp; ++c) {
    const JpegComp& C = J.comp[c];
    const size_t nblk = static_cast<size_t>(C.blocks_x) * C.blocks_y;
    int16_t* d_coef = reinterpret_cast<int16_t*>(d + L.coef[c]);
    if (coef_on_device) {
      // the GPU entropy decoder already filled the plane
    } else if (C.sparse) {
      IST_HIP(hipMemsetAsync(d_coef, 0, nblk * 128, stream));
      if (!d_ent) return fail(IST_E_INVALID, "JPEG sparse coefficients without a device arena");
      if (!C.ent.empty()) IST_HIP(hipMemcpyAsync(d_ent + L.ent[c], C.ent.data(), C.ent.size() * 4, hipMemcpyHostToDevice, stream));
      IST_HIP(hipMemcpyAsync(d_ent + L.start[c], C.start.data(), nblk * 4, hipMemcpyHostToDevice, stream));
      IST_HIP(hipMemcpyAsync(d_ent + L.cnt[c], C.cnt.data(), nblk, hipMemcpyHostToDevice, stream));
      const int rc = jpeg_launch_scatter(reinterpret_cast<const uint32_t*>(d_ent + L.ent[c]), reinterpret_cast<const uint32_t*>(d_ent + L.start[c]), d_ent + L.cnt[c], d_coef, static_cast<int>(nblk), stream);
      if (rc) return rc;
    } else {
      if (C.coef.size() != nblk * 64) return fail(IST_E_DECODE, "JPEG component without coefficients");
      IST_HIP(hipMemcpyAsync(d_coef, C.coef.data(), nblk * 128, hipMemcpyHostToDevice, stream));
    }
  }
  return jpeg_launch_reconstruct(job, stream);
}
}  // namespace
}  // extern "C++"

int ist_jpeg_info(const uint8_t* file, int64_t len, int32_t* width, int32_t* height, int32_t* orientation) {
  JpegImage J;
  const int rc = jpeg_parse_and_entropy_decode(file, len, &J, true);
  if (rc) return rc;
  if (width) *width = J.width;
  if (height) *height = J.height;
  if (orientation) *orientation = J.orientation;
  return IST_OK;
}

int ist_jpeg_decode_rgba8(ist_ctx* ctx, const uint8_t* file, int64_t len, uint8_t* out, size_t out_pitch, int64_t out_rows) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  JpegImage J;
  int rc = jpeg_parse_and_entropy_decode(file, len, &J, false);
  if (rc) return rc;
  if (!out || out_pitch < static_cast<size_t>(J.width) * 4 || out_rows < J.height) return fail(IST_E_INVALID, "ist_jpeg_decode_rgba8: output buffer too small");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  // one device allocation: coefficients + tables + sample planes + RGBA
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~static_cast<size_t>(255); return at; };
  JpegDevLayout L;
  jpeg_layout(J, &off, &L);
  jpeg_layout_sparse(J, &off, &L);                      // (one arena holds both parts here)
  const size_t row = static_cast<size_t>(J.width) * 4;
  const size_t o_out = take(row * J.height);
  uint8_t* d = nullptr;
  rc = grow_device(&ctx->scratch_arena, &ctx->scratch_arena_bytes, off);
  if (rc) return rc;
  d = static_cast<uint8_t*>(ctx->scratch_arena);
  rc = jpeg_enqueue(J, d, d, L, d + o_out, row, ctx->stream);
  if (rc) return rc;
  std::vector<RowsCopy> down{RowsCopy{d + o_out, nullptr, out, out_pitch, row, static_cast<size_t>(J.height)}};
  rc = stager_of(ctx).download(down, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);              // nothing of this call may still read the arena when the next call reuses it
  return rc;
}

// format-agnostic front door: PNG (host decode) or JPEG (host entropy decode + GPU reconstruction)
extern "C" int ist_misc_info(const uint8_t* file, int64_t len, int32_t* w, int32_t* h);
extern "C" int ist_misc_decode_rgba8(const uint8_t* file, int64_t len, uint8_t* out, size_t pitch, int64_t out_rows);
static bool is_jpeg(const uint8_t* f, int64_t n) { return f && n >= 2 && f[0] == 0xFF && f[1] == 0xD8; }
static bool is_misc(const uint8_t* f, int64_t n) { return f && n >= 4 && ((f[0] == 'B' && f[1] == 'M') || !std::memcmp(f, "GIF8", 4)); }

int ist_image_info(const uint8_t* file, int64_t len, int32_t* width, int32_t* height, int32_t* orientation) {
  if (is_jpeg(file, len)) return ist_jpeg_info(file, len, width, height, orientation);
  if (is_webp(file, len)) return webp_info(file, len, width, height, orientation);      // EXIF chunk of the container
  if (orientation) *orientation = 0;
  if (is_misc(file, len)) return ist_misc_info(file, len, width, height);
  return ist_png_info(file, len, width, height);
}

int ist_image_decode_rgba8(ist_ctx* ctx, const uint8_t* file, int64_t len, uint8_t* out, size_t out_pitch, int64_t out_rows) {
  if (is_jpeg(file, len)) return ist_jpeg_decode_rgba8(ctx, file, len, out, out_pitch, out_rows);
  if (is_misc(file, len)) return ist_misc_decode_rgba8(file, len, out, out_pitch, out_rows);
  if (is_webp(file, len)) return webp_decode_rgba8(file, len, out, out_pitch, out_rows);
  return ist_png_decode_rgba8(file, len, out, out_pitch, out_rows);
}

// ---- files -> bitmaps in HBM: the decode stage shared by ist_stitch_files_png and ist_decode_files_device ------------
// (index.js:1441-1520 decodes image after image; :1559-1571 flushes and releases each one.)  Every image has a host thread:
// container parse + de-stuffing, and - baseline JPEG - the upload of its scan on a stream of its own, so that the uploads
// run while other images are still being parsed.  The Huffman passes of ALL eligible images then run as ONE batch on the
// consumer's stream: the decoder is latency-bound per workgroup (a 12 MP photo is 58 workgroups), so nine images in one
// launch take as long as one, whereas one chain per image on nine streams took 2x longer than the batch (measured: the
// runtime multiplexes streams onto four hardware queues, three chains per queue ran back to back).  Behind the batch the
// images are reconstructed one by one as the consumer asks for them, so (ist_stitch_files_png) band k of the canvas is
// rendered and exported while the images behind it are still being reconstructed.  Files the GPU entropy decoder does not
// take (progressive, non-interleaved scans, more than 2048 restart intervals, PNG / BMP / GIF / WebP) are decoded on their thread and uploaded when the consumer
// asks for the image.  With phase timing on, the same steps run with a stream sync between them.
extern "C++" {
namespace {

struct Dec { int rc = 0; std::string err; bool jpeg = false; JpegImage J; JpegGpuScan G; int w = 0, h = 0, orient = 0; std::vector<uint8_t> px; };

// phase clock: stderr lines under IST_TIMING=1, numbers for ist_ctx_last_timing when the context asked for them.  Phases
// end with a stream synchronisation only while one of the two is on.
struct Phases {
  ist_ctx* ctx; bool print, on;
  std::chrono::steady_clock::time_point t_prev;
  explicit Phases(ist_ctx* c) : ctx(c) {
    static const bool env = std::getenv("IST_TIMING") != nullptr;
    print = env; on = env || c->timing_on;
    if (c->timing_on) for (double& v : c->last_ms) v = 0.0;
    t_prev = std::chrono::steady_clock::now();
  }
  void lap(int phase, const char* what, hipStream_t st) {
    if (!on) return;
    if (st) (void)hipStreamSynchronize(st);
    const auto t = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(t - t_prev).count();
    if (print) std::fprintf(stderr, "[ist timing] %-28s %8.2f ms\n", what, ms);
    if (ctx->timing_on && phase >= 0 && phase < IST_PHASE_COUNT) ctx->last_ms[phase] += ms;
    t_prev = t;
  }
};

constexpr int kImgStreams = 8;           // image i runs on stream i mod kImgStreams

int ensure_image_lanes(ist_ctx* ctx, int n) {
  const size_t want = static_cast<size_t>(std::min(n, kImgStreams));
  while (ctx->img_stream.size() < want) {
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return fail(IST_E_HIP, "hipStreamCreate failed"); }
    ctx->img_stream.push_back(st);
  }
  while (ctx->img_event.size() < static_cast<size_t>(n)) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return fail(IST_E_HIP, "hipEventCreate failed"); }
    ctx->img_event.push_back(ev);
  }
  if (ctx->img_huff.size() < static_cast<size_t>(n)) { ctx->img_huff.resize(static_cast<size_t>(n), nullptr); ctx->img_huff_bytes.resize(static_cast<size_t>(n), 0); }
  if (ctx->scan_bufs.size() < static_cast<size_t>(n)) ctx->scan_bufs.resize(static_cast<size_t>(n));
  return IST_OK;
}

// One call's decode work.  Lifetime: construct -> headers() -> (caller lays out its arena) -> start() -> take(i) for every
// image the caller consumes, in any order -> finish().  The destructor joins whatever still runs.
class FileDecoder {
 public:
  FileDecoder(ist_ctx* ctx, const uint8_t* const* files, const int64_t* lens, int n, Phases* ph)
      : ctx_(ctx), files_(files), lens_(lens), n_(n), ph_(ph), dec_(static_cast<size_t>(n)),
        on_gpu_(static_cast<size_t>(n), 0), taken_(static_cast<size_t>(n), 0), uploaded_(static_cast<size_t>(n), 0), started_(static_cast<size_t>(n), 0),
        chroma_done_(static_cast<size_t>(n), 0), jo_(static_cast<size_t>(n)) {}
  ~FileDecoder() {
    join_all();
    for (int i = 0; i < n_; ++i) if (started_[static_cast<size_t>(i)]) (void)hipStreamSynchronize(stream_of(i));
    for (int i = 0; i < n_ && static_cast<size_t>(i) < ctx_->scan_bufs.size(); ++i) {      // keep the scans' memory for the next call (at most 8 MiB per image)
      ScanBuf& mine = dec_[static_cast<size_t>(i)].G.stream;
      if (mine.capacity() > ctx_->scan_bufs[static_cast<size_t>(i)].capacity() && mine.capacity() <= (8u << 20)) ctx_->scan_bufs[static_cast<size_t>(i)].swap(mine);
    }
  }

  // 1. frame headers only (microseconds per file): sizes, sampling, EXIF orientation - what the planner and the arena need
  int headers() {
    static const bool gpu_huffman = std::getenv("IST_JPEG_HOST_HUFFMAN") == nullptr;
    gpu_huffman_ = gpu_huffman;
    for (int i = 0; i < n_; ++i) {
      Dec& D = dec_[static_cast<size_t>(i)];
      const uint8_t* f = files_[i]; const int64_t len = lens_[i];
      D.jpeg = f && len >= 2 && f[0] == 0xFF && f[1] == 0xD8;
      int rc;
      if (D.jpeg) {
        rc = jpeg_parse_and_entropy_decode(f, len, &D.J, true);
        D.w = D.J.width; D.h = D.J.height; D.orient = D.J.orientation;
      } else {
        int32_t w = 0, h = 0, o = 0;
        rc = ist_image_info(f, len, &w, &h, &o);
        D.w = w; D.h = h; D.orient = o;                    // WebP carries EXIF in its container
      }
      if (rc != IST_OK) return fail(rc, "图片" + std::to_string(i) + "解码异常: " + g_last_error);   // index.js:1512-1514
    }
    return IST_OK;
  }
  const Dec& dec(int i) const { return dec_[static_cast<size_t>(i)]; }
  // device bytes of the JPEG stages (coefficient planes, tables, sample planes), carved from *off of the caller's arena
  void layout(size_t* off) { for (int i = 0; i < n_; ++i) if (dec_[static_cast<size_t>(i)].jpeg) jpeg_layout(dec_[static_cast<size_t>(i)].J, off, &jo_[static_cast<size_t>(i)]); }

  // 2. the workers.  arena: what layout() was sized for; img[i] / pitch[i]: where bitmap i goes (device memory)
  int start(uint8_t* arena, uint8_t* const* img, const size_t* pitch) {
    arena_ = arena; img_ = img; pitch_ = pitch;
    const int rc = ensure_image_lanes(ctx_, n_);
    if (rc) return rc;
    if (!ctx_->workers) ctx_->workers.reset(new WorkerPool());
    ctx_->workers->run(n_, [this](int i) { worker(i); });
    running_ = true;
    if (!ph_->on) return IST_OK;
    // phase timing: the steps one after the other
    join_all();
    int rc2 = first_error(); if (rc2) return rc2;
    for (int i = 0; i < n_; ++i) if (uploaded_[static_cast<size_t>(i)]) (void)hipStreamSynchronize(stream_of(i));
    ph_->lap(IST_PHASE_HOST_DECODE, "decode on host threads (+ scan uploads)", nullptr);
    rc2 = huffman_all(ctx_->stream); if (rc2) return rc2;
    ph_->lap(IST_PHASE_ENTROPY_GPU, "entropy decode (GPU)", ctx_->stream);
    for (int i = 0; i < n_; ++i) { rc2 = take(i, ctx_->stream); if (rc2) return rc2; }
    ph_->lap(IST_PHASE_RECONSTRUCT, "H2D + JPEG reconstruct (GPU)", ctx_->stream);
    return IST_OK;
  }

  // 3. bitmap i is needed by work that will be submitted to `consumer` next.  The first call waits for every image's HOST
  // side and runs the Huffman batch on `consumer`; then image i is reconstructed (or, a file the GPU path did not take,
  // uploaded / reconstructed from host coefficients) on `consumer`.  Idempotent per image; one consumer stream per call.
  int take(int i, hipStream_t consumer) {
    const size_t k = static_cast<size_t>(i);
    if (taken_[k]) return IST_OK;
    int rc = huffman_all(consumer);
    if (rc) return rc;
    rc = chroma_all(consumer);
    if (rc) return rc;
    Dec& D = dec_[k];
    taken_[k] = 1;
    if (on_gpu_[k]) return jpeg_enqueue(D.J, arena_, nullptr, jo_[k], img_[i], pitch_[i], consumer, true, chroma_done_[k] != 0);
    const size_t row = static_cast<size_t>(D.w) * 4;
    if (!D.jpeg) {                                  // PNG / BMP / GIF / WebP: decoded on the thread, uploaded here
      std::vector<RowsCopy> up;
      if (pitch_[i] != row) for (int y = 0; y < D.h; ++y) up.push_back(RowsCopy{img_[i] + static_cast<size_t>(y) * pitch_[i], D.px.data() + static_cast<size_t>(y) * row, nullptr, row, row, 1});
      else up.push_back(RowsCopy{img_[i], D.px.data(), nullptr, row, row, static_cast<size_t>(D.h)});
      return stager_of(ctx_).upload(up, consumer);
    }
    // a JPEG whose coefficients are on the host (progressive, non-interleaved scans, thousands of restart intervals, or a file that
    // failed the GPU decoder's validation and is decoded again by the host decoder)
    if (D.G.eligible) {
      D.G.eligible = false;
      JpegImage host;
      rc = jpeg_parse_and_entropy_decode(files_[i], lens_[i], &host, false, nullptr);
      if (rc) return fail(rc, "图片" + std::to_string(i) + "解码异常: " + g_last_error);
      D.J = std::move(host);
    }
    size_t need = 0;
    JpegDevLayout L = jo_[k];
    jpeg_layout_sparse(D.J, &need, &L);
    if (need) {                                      // one block serves the host-decoded images in turn
      (void)hipStreamSynchronize(consumer);
      if (need > ctx_->scratch_ent_bytes) { rc = grow_device(&ctx_->scratch_ent, &ctx_->scratch_ent_bytes, need + need / 2); if (rc) return rc; }
    }
    return jpeg_enqueue(D.J, arena_, static_cast<uint8_t*>(ctx_->scratch_ent), L, img_[i], pitch_[i], consumer, false);
  }

  int finish(hipStream_t consumer) {
    for (int i = 0; i < n_; ++i) { const int rc = take(i, consumer); if (rc) { join_all(); return rc; } }
    return IST_OK;
  }
  int gpu_decoded() const { int g = 0; for (char v : on_gpu_) g += v ? 1 : 0; return g; }

 private:
  hipStream_t stream_of(int i) const { return ctx_->img_stream[static_cast<size_t>(i % kImgStreams) % ctx_->img_stream.size()]; }
  void join_all() { if (running_) { ctx_->workers->wait(); running_ = false; } }
  int first_error() {
    for (int i = 0; i < n_; ++i) if (dec_[static_cast<size_t>(i)].rc != IST_OK) return fail(dec_[static_cast<size_t>(i)].rc, "图片" + std::to_string(i) + "解码异常: " + dec_[static_cast<size_t>(i)].err);
    return IST_OK;
  }
  // every worker has returned; ONE Huffman batch over the eligible images on `consumer`, behind their scan uploads
  int huffman_all(hipStream_t consumer) {
    if (huff_done_) return IST_OK;
    tl_mark("decoder: waiting for the per-image host work");
    join_all();
    tl_mark("decoder: host work of every image done");
    int rc = first_error();
    if (rc) return rc;
    huff_done_ = true;
    std::vector<JpegGpuItem> items; std::vector<int> who;
    for (int i = 0; i < n_; ++i) {
      const size_t k = static_cast<size_t>(i);
      Dec& D = dec_[k];
      if (!D.jpeg || !D.G.eligible) continue;
      JpegGpuItem it; it.J = &D.J; it.S = &D.G;
      for (int c = 0; c < 3; ++c) it.d_coef[c] = c < D.J.ncomp ? reinterpret_cast<int16_t*>(arena_ + jo_[k].coef[c]) : nullptr;
      if (uploaded_[k]) {
        it.d_stream = static_cast<const uint8_t*>(ctx_->img_huff[k]);
        if (hipStreamWaitEvent(consumer, ctx_->img_event[k], 0) != hipSuccess) { (void)hipGetLastError(); return fail(IST_E_HIP, "hipStreamWaitEvent failed"); }
      }
      items.push_back(it); who.push_back(i);
    }
    std::vector<uint8_t> okv;
    rc = jpeg_gpu_entropy_decode(items, &okv, consumer, &ctx_->scratch_huff, &ctx_->scratch_huff_bytes);
    if (rc) return rc;
    for (size_t q = 0; q < who.size(); ++q) {
      on_gpu_[static_cast<size_t>(who[q])] = okv[q] ? 1 : 0;
      if (okv[q]) g_gpu_entropy_files.fetch_add(1, std::memory_order_relaxed);
    }
    return IST_OK;
  }
  // the chroma planes of every image the GPU decoded, in ONE launch behind the batch (the first take() runs it: part of the
  // reconstruction phase): each image then costs one fused launch
  int chroma_all(hipStream_t consumer) {
    if (chroma_batch_done_) return IST_OK;
    chroma_batch_done_ = true;
    std::vector<JpegDeviceJob> chroma;
    for (int i = 0; i < n_; ++i) {
      const size_t k = static_cast<size_t>(i);
      if (!on_gpu_[k] || dec_[k].J.ncomp != 3) continue;
      chroma.push_back(jpeg_device_job(dec_[k].J, arena_, jo_[k], nullptr, 0));
      chroma_done_[k] = 1;
    }
    if (!chroma.empty()) return jpeg_launch_chroma_idct(chroma.data(), static_cast<int>(chroma.size()), consumer);
    return IST_OK;
  }
  // container + host entropy stage of image i; a baseline JPEG's de-stuffed scan goes up on the image's own stream
  void worker(int i) {
    const size_t k = static_cast<size_t>(i);
    Dec& D = dec_[k];
    DeviceGuard dg(ctx_->device);
    const uint8_t* f = files_[i]; const int64_t len = lens_[i];
    auto failed = [&](int rc) { D.rc = rc; D.err = g_last_error; };     // (thread-local message: carry it out)
    if (!D.jpeg) {
      D.px.resize(static_cast<size_t>(D.w) * D.h * 4);
      const int rc = ist_image_decode_rgba8(nullptr, f, len, D.px.data(), static_cast<size_t>(D.w) * 4, D.h);
      if (rc) failed(rc);
      return;
    }
    JpegImage full;
    // The scan is de-stuffed (SSE2, 16 bytes a step) into a heap block the context keeps from call to call and goes to the
    // device in ONE copy.  (measured, 120 calls each, twice: from a page-locked block of the context instead - a true DMA, no
    // bounce buffer - the call's median was the same, 2.35 vs 2.34-2.37 ms, so the simpler path stayed; in 256 KB pieces sent
    // while the rest was still being de-stuffed, half of the calls took 8 ms.)
    if (k < ctx_->scan_bufs.size()) D.G.stream.swap(ctx_->scan_bufs[k]);            // (a recycled block: capacity, no contents)
    // (IST_TUNING=1 IST_JPEG_SECOND_READ_444=1, tests only: the second read sees the luma sampling factors as 1x1 - what a
    // caller's buffer rewritten between the two parses would look like)
    static const bool flip = tuning_mode() && std::getenv("IST_JPEG_SECOND_READ_444") != nullptr;
    std::vector<uint8_t> flipped;
    if (flip) {
      flipped.assign(f, f + len);
      for (int64_t q = 2; q + 12 < len; ++q) if (flipped[static_cast<size_t>(q)] == 0xFF && flipped[static_cast<size_t>(q) + 1] == 0xC0) { flipped[static_cast<size_t>(q) + 11] = 0x11; break; }
      f = flipped.data();
    }
    const int rc = jpeg_parse_and_entropy_decode(f, len, &full, false, gpu_huffman_ ? &D.G : nullptr);
    if (rc) { failed(rc); return; }
    // The arena (coefficient + sample planes, layout()) was sized from the header-only parse: every input of that layout must
    // be the same on this second read, or the Huffman write kernel, the IDCT and the scatter would run past their planes
    // (4:2:0 turning 4:4:4 doubles blocks_x * blocks_y).  The bytes may be a caller's buffer another thread is still writing.
    bool same = full.width == D.w && full.height == D.h && full.ncomp == D.J.ncomp && full.hmax == D.J.hmax && full.vmax == D.J.vmax &&
                full.mcus_x == D.J.mcus_x && full.mcus_y == D.J.mcus_y;
    for (int c = 0; same && c < full.ncomp; ++c)
      same = full.comp[c].h == D.J.comp[c].h && full.comp[c].v == D.J.comp[c].v && full.comp[c].blocks_x == D.J.comp[c].blocks_x &&
             full.comp[c].blocks_y == D.J.comp[c].blocks_y;
    if (!same) { g_last_error = "JPEG frame header changed between two reads"; failed(IST_E_DECODE); return; }
    D.J = std::move(full);
    if (!D.G.eligible) return;
    const size_t bytes = D.G.stream.size();
    if (ctx_->img_huff_bytes[k] < bytes) {
      dev_free(ctx_->img_huff[k]); ctx_->img_huff[k] = nullptr; ctx_->img_huff_bytes[k] = 0;
      if (dev_malloc(&ctx_->img_huff[k], bytes + bytes / 4) != 0) { (void)hipGetLastError(); return; }
      ctx_->img_huff_bytes[k] = bytes + bytes / 4;
    }
    hipStream_t st = stream_of(i);
    started_[k] = 1;
    if (hipMemcpyAsync(ctx_->img_huff[k], D.G.stream.data(), bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipEventRecord(ctx_->img_event[k], st) != hipSuccess) { (void)hipGetLastError(); return; }
    uploaded_[k] = 1;
  }

  ist_ctx* ctx_; const uint8_t* const* files_; const int64_t* lens_; int n_; Phases* ph_;
  std::vector<Dec> dec_;
  bool running_ = false;                                       // the context's worker pool is on this call's files
  std::vector<char> on_gpu_, taken_, uploaded_, started_, chroma_done_;      // started_: the image's stream carries uploads of this call; chroma_done_: its chroma planes were made behind the Huffman batch
  std::vector<JpegDevLayout> jo_;
  uint8_t* arena_ = nullptr; uint8_t* const* img_ = nullptr; const size_t* pitch_ = nullptr;
  bool gpu_huffman_ = true, huff_done_ = false, chroma_batch_done_ = false;
};

// One stitch cut into a background launch + one launch per draw (the same cut the device group uses, ist_shard_parts with a
// slot per image): band k can be rendered - and exported - as soon as image k is decoded.  ok = false when the draws overlap
// (or there is nothing to cut): the caller then renders the canvas with ONE launch once every image is there.
struct BandedJobs {
  bool ok = false;
  ist_job* bg = nullptr;
  std::vector<ist_job*> band;            // per part
  std::vector<ist_part> parts;           // sorted by Y0
  ~BandedJobs() { if (bg) ist_job_destroy(bg); for (ist_job* j : band) if (j) ist_job_destroy(j); }
};

int compile_banded(ist_ctx* ctx, int64_t cw, int64_t ch, const uint8_t clear[4], const ist_op* ops, int n_ops, const ist_image_desc* images,
                   int n_images, int filter, BandedJobs* out) {
  std::vector<ist_part> cut(static_cast<size_t>(n_ops) + 8);
  int n_parts = 0;
  if (ist_shard_parts(ops, n_ops, cw, ch, images, n_images, filter, std::max(1, n_images), IST_SPLIT_IMAGE, cut.data(), static_cast<int>(cut.size()), &n_parts) != IST_OK || n_parts < 2)
    return IST_OK;                       // overlapping draws (or a single image): not banded, not an error
  // the per-band events are the context's per-IMAGE events (ensure_image_lanes): a cut with more parts than images - an image
  // drawn twice, a draw split in two - is rendered whole instead
  if (n_parts > n_images) return IST_OK;
  cut.resize(static_cast<size_t>(n_parts));
  std::stable_sort(cut.begin(), cut.end(), [](const ist_part& a, const ist_part& b) { return a.Y0 < b.Y0; });
  std::vector<ist_op> bg_ops;
  int fill_at = -1;
  for (int k = 0; k < n_ops; ++k) { if (ops[k].kind != IST_OP_DRAW) bg_ops.push_back(ops[k]); if (fill_at < 0 && ops[k].kind == IST_OP_FILL) fill_at = k; }
  for (const ist_part& p : cut) {
    ist_op hole;
    std::memset(&hole, 0, sizeof hole);
    hole.kind = IST_OP_HOLE; hole.image = -1; hole.m[0] = 1.0; hole.m[3] = 1.0;
    hole.d[0] = p.X0; hole.d[1] = p.Y0; hole.d[2] = p.X1 - p.X0; hole.d[3] = p.Y1 - p.Y0;
    bg_ops.push_back(hole);
  }
  out->bg = ist_job_create(ctx, cw, ch, clear, bg_ops.data(), static_cast<int>(bg_ops.size()), images, n_images, filter, nullptr);
  if (!out->bg) return g_last_code ? g_last_code : IST_E_INVALID;
  for (const ist_part& p : cut) {
    ist_op two[2]; int n2 = 0;
    if (fill_at >= 0 && fill_at < p.op) two[n2++] = ops[fill_at];
    two[n2++] = ops[p.op];
    const ist_region clip{p.X0, p.Y0, p.X1 - p.X0, p.Y1 - p.Y0};
    ist_job* j = ist_job_create(ctx, cw, ch, clear, two, n2, images, n_images, filter, &clip);
    if (!j) return g_last_code ? g_last_code : IST_E_INVALID;
    out->band.push_back(j);
  }
  out->parts = cut;
  out->ok = true;
  return IST_OK;
}

int stitch_files_png_locked(ist_ctx* ctx, const uint8_t* const* files, const int64_t* lens, int n_images, int direction, int mode,
                            double gap, const ist_limits* limits, int filter, ist_plan* out_plan, uint8_t** out_png, int64_t* out_len);

}  // namespace
}  // extern "C++"

int ist_ctx_set_timing(ist_ctx* ctx, int on) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  ctx->timing_on = on != 0;
  return IST_OK;
}

int ist_ctx_last_timing(ist_ctx* ctx, double* ms, int n) {
  if (!ctx || !ms || n < 0) return fail(IST_E_INVALID, "ist_ctx_last_timing: bad argument");
  for (int k = 0; k < n; ++k) ms[k] = k < IST_PHASE_COUNT ? ctx->last_ms[k] : 0.0;
  return IST_OK;
}

// files -> decoded bitmaps in caller-owned device memory (the Image.src step, utils/canvas.js:27-121, ending in HBM)
int ist_decode_files_device(ist_ctx* ctx, const uint8_t* const* files, const int64_t* lens, int n_images, void* const* dst,
                            const size_t* dst_pitch, const int64_t* dst_rows, ist_image_desc* out_descs) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  if (n_images <= 0) return IST_NOTHING_TO_DO;
  if (!files || !lens || !dst || !dst_pitch || !dst_rows) return fail(IST_E_INVALID, "ist_decode_files_device: NULL argument");
  if (n_images > kMaxImages) return fail(IST_E_UNSUPPORTED, "more than 128 images in one call");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  Phases ph(ctx);
  FileDecoder fd(ctx, files, lens, n_images, &ph);
  int rc = fd.headers();
  if (rc) return rc;
  std::vector<uint8_t*> img(static_cast<size_t>(n_images));
  for (int i = 0; i < n_images; ++i) {
    const Dec& D = fd.dec(i);
    // the file's own header is untrusted: the caller states what its buffer holds
    if (!dst[i] || dst_pitch[i] < static_cast<size_t>(D.w) * 4 || (dst_pitch[i] & 3) || dst_rows[i] < D.h || (reinterpret_cast<uintptr_t>(dst[i]) & 3))
      return fail(IST_E_INVALID, "ist_decode_files_device: the buffer of image " + std::to_string(i) + " is too small for " + std::to_string(D.w) + "x" + std::to_string(D.h));
    img[static_cast<size_t>(i)] = static_cast<uint8_t*>(dst[i]);
    if (out_descs) {
      ist_image_desc& d = out_descs[i];
      std::memset(&d, 0, sizeof d);
      d.width = D.w; d.height = D.h; d.orientation = D.orient ? D.orient : 1; d.opaque = D.jpeg ? 1 : 0; d.file_size = lens[i];
    }
  }
  size_t off = 0;
  fd.layout(&off);
  rc = grow_device(&ctx->scratch_dec, &ctx->scratch_dec_bytes, off ? off : 256);
  if (rc) return rc;
  ph.lap(IST_PHASE_PLAN_ARENA, "device arena", nullptr);
  rc = fd.start(static_cast<uint8_t*>(ctx->scratch_dec), img.data(), dst_pitch);
  if (rc) return rc;
  rc = fd.finish(ctx->stream);
  if (rc) return rc;
  IST_HIP(hipStreamSynchronize(ctx->stream));      // the bitmaps are complete; the host coefficients in flight may go
  return IST_OK;
}

// ---- the whole onStitch for files, device-resident: only file bytes go in and only PNG bytes come out over PCIe -------
// decode (index.js:1441-1520) -> plan (1251-1386) -> resample+blit (1532-1551) -> PNG export (1577-1579), PIPELINED: the
// planner needs only the frame headers, so the canvas is laid out first; then every image decodes on its own thread and
// stream (FileDecoder), band k of the canvas is rendered as soon as image k is there, and the PNG encoder - which works slab
// by slab, each slab crossing PCIe while the next one compresses - asks for canvas rows as it reaches them.  The first
// slabs of the file are on their way to the host while the last images are still in the Huffman decoder.
int ist_stitch_files_png(ist_ctx* ctx, const uint8_t* const* files, const int64_t* lens, int n_images, int direction, int mode,
                         double gap, const ist_limits* limits, int filter, ist_plan* out_plan, uint8_t** out_png,
                         int64_t* out_len) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  if (!out_plan || !out_png || !out_len) return fail(IST_E_INVALID, "ist_stitch_files_png: NULL output");
  *out_png = nullptr; *out_len = 0;
  std::memset(out_plan, 0, sizeof(*out_plan));
  if (n_images <= 0) return IST_NOTHING_TO_DO;
  if (!files || !lens) return fail(IST_E_INVALID, "ist_stitch_files_png: NULL input");
  if (n_images > kMaxImages) return fail(IST_E_UNSUPPORTED, "more than 128 images in one launch");
  std::lock_guard<std::mutex> lock(ctx->mu);
  return stitch_files_png_locked(ctx, files, lens, n_images, direction, mode, gap, limits, filter, out_plan, out_png, out_len);
}

extern "C++" {
namespace {
// (the caller holds ctx->mu and has checked the arguments)
int stitch_files_png_locked(ist_ctx* ctx, const uint8_t* const* files, const int64_t* lens, int n_images, int direction, int mode,
                            double gap, const ist_limits* limits, int filter, ist_plan* out_plan, uint8_t** out_png, int64_t* out_len) {
  const int n = n_images;
  DeviceGuard g(ctx->device);
  Phases ph(ctx);
  // (IST_TUNING=1 IST_TIMELINE=1: host-side marks of one call on stderr, microseconds from its start)
  const bool tl_outer = tl_active();                      // (ist_stitch_paths_png started the call's clock: its file reads are part of the call)
  if (!tl_outer) tl_begin();
  struct TlEnd { bool mine; ~TlEnd() { if (mine) tl_end("ist_stitch_files_png"); } } tl_end_guard{!tl_outer};
  auto mark = [&](const char* what) { tl_mark(what); };
  FileDecoder fd(ctx, files, lens, n, &ph);
  int rc = fd.headers();
  if (rc) return rc;
  mark("headers parsed");
  // plan (orientation from the file, like getImageInfo -> index.js:734)
  std::vector<ist_image_desc> descs(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) {
    const Dec& D = fd.dec(i);
    ist_image_desc& d = descs[static_cast<size_t>(i)];
    std::memset(&d, 0, sizeof d);
    d.width = D.w; d.height = D.h; d.orientation = D.orient ? D.orient : 1; d.opaque = D.jpeg ? 1 : 0; d.file_size = lens[i];
  }
  ist_limits lim;
  if (limits) lim = *limits; else ist_limits_unlimited(&lim);
  rc = ist_plan_compute(descs.data(), n, direction, mode, gap, &lim, out_plan);
  if (rc != IST_OK) return rc;
  struct PlanGuard { ist_plan* p; bool keep = false; ~PlanGuard() { if (!keep) ist_plan_free(p); } } pg{out_plan};
  std::vector<ist_op> ops(static_cast<size_t>(out_plan->n_rects) + 1);
  int n_ops = 0;
  rc = ist_plan_ops(out_plan, descs.data(), n, ops.data(), &n_ops);
  if (rc != IST_OK) return rc;

  // one device arena (the context's, grow-only): bitmaps, JPEG coefficient planes + sample planes, canvas, PNG
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~static_cast<size_t>(255); return at; };
  std::vector<size_t> o_img(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) o_img[static_cast<size_t>(i)] = take(static_cast<size_t>(fd.dec(i).w) * 4 * fd.dec(i).h + 16);
  fd.layout(&off);
  const size_t canvas_pitch = static_cast<size_t>(out_plan->canvas_w) * 4;
  const size_t o_canvas = take(canvas_pitch * static_cast<size_t>(out_plan->canvas_h));
  const int64_t png_cap = ist_png_bound(out_plan->canvas_w, out_plan->canvas_h);
  const size_t o_png = take(static_cast<size_t>(png_cap));
  rc = grow_device(&ctx->scratch_arena, &ctx->scratch_arena_bytes, off);
  if (rc) return rc;
  uint8_t* d = static_cast<uint8_t*>(ctx->scratch_arena);
  ph.lap(IST_PHASE_PLAN_ARENA, "plan + device arena", nullptr);
  std::vector<uint8_t*> img(static_cast<size_t>(n));
  std::vector<const void*> dsrc(static_cast<size_t>(n));
  std::vector<size_t> dpitch(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) {
    img[static_cast<size_t>(i)] = d + o_img[static_cast<size_t>(i)];
    dsrc[static_cast<size_t>(i)] = img[static_cast<size_t>(i)];
    dpitch[static_cast<size_t>(i)] = static_cast<size_t>(fd.dec(i).w) * 4;
  }
  mark("plan + arena");
  rc = fd.start(d, img.data(), dpitch.data());          // the images decode from here on
  if (rc) return rc;
  mark("workers started");
  // Two streams: RENDER (Huffman batch, then per image: reconstruction + its band of the canvas) and ctx->stream (the PNG
  // encoder, which waits for band k's event before it compresses the slabs that read it).  On one stream the reconstruction
  // of image k+1 sat between the slabs of band k and band k+1 and cost its full time; on its own stream it runs beside them.
  rc = ensure_aux(ctx);
  if (rc) return rc;
  if (!ctx->render) {                                  // (high priority: its short kernels should not queue behind the encoder's thousands of workgroups)
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&ctx->render, hipStreamNonBlocking, hi) != hipSuccess) { (void)hipGetLastError(); ctx->render = nullptr; return fail(IST_E_HIP, "hipStreamCreate failed"); }
  }
  hipStream_t render = ph.on ? ctx->stream : ctx->render;
  // the stitch, cut per image (compiled while the workers parse): background now, band k when image k is there
  static const uint8_t transparent[4] = {0, 0, 0, 0};
  BandedJobs bj;
  rc = compile_banded(ctx, out_plan->canvas_w, out_plan->canvas_h, transparent, ops.data(), n_ops, descs.data(), n, filter, &bj);
  if (rc) return rc;
  mark("banded jobs compiled");
  ist_job* whole = nullptr;
  struct JobFree { ist_job** j; ~JobFree() { if (*j) ist_job_destroy(*j); } } jf{&whole};
  if (!bj.ok) {
    whole = ist_job_create(ctx, out_plan->canvas_w, out_plan->canvas_h, transparent, ops.data(), n_ops, descs.data(), n, filter, nullptr);
    if (!whole) return g_last_code ? g_last_code : IST_E_INVALID;
  } else {
    rc = ist_job_launch(bj.bg, dsrc.data(), dpitch.data(), n, d + o_canvas, canvas_pitch, render);
    if (rc) return rc;
  }
  // A draw that only MOVES its image - no scaling, no turn, whole pixels, nothing clipped, an opaque source - needs no bitmap
  // of its own and no launch: the image is reconstructed straight into its box of the canvas (the colour kernel writes with the
  // canvas pitch).  That is every draw of a same-width vertical strip (BASELINE configs[1]): per 12 MP photo 48 MB less to
  // write, 96 MB less to read and write again, and one launch less.  (Pipelined mode only: the phase-timed run keeps the
  // stitch a step of its own; both make the same canvas.)
  std::vector<char> direct(bj.parts.size(), 0);
  if (bj.ok && !ph.on) {
    for (size_t k = 0; k < bj.parts.size(); ++k) {
      const ist_part& p = bj.parts[k];
      const Dec& D = fd.dec(p.image);
      const ist_op& o = ops[static_cast<size_t>(p.op)];
      const double X = o.m[4] + o.d[0], Y = o.m[5] + o.d[1];
      ist_job_info info;
      if (!D.jpeg || descs[static_cast<size_t>(p.image)].orientation != 1 || o.kind != IST_OP_DRAW || o.image != p.image) continue;
      if (o.m[0] != 1.0 || o.m[1] != 0.0 || o.m[2] != 0.0 || o.m[3] != 1.0) continue;
      if (o.s[0] != 0.0 || o.s[1] != 0.0 || o.s[2] != D.w || o.s[3] != D.h || o.d[2] != D.w || o.d[3] != D.h) continue;
      if (X != std::floor(X) || Y != std::floor(Y) || X < 0 || Y < 0 || X + D.w > out_plan->canvas_w || Y + D.h > out_plan->canvas_h) continue;
      if (p.X0 != static_cast<int32_t>(X) || p.Y0 != static_cast<int32_t>(Y) || p.X1 - p.X0 != D.w || p.Y1 - p.Y0 != D.h) continue;
      if (ist_job_info_get(bj.band[k], &info) != IST_OK || info.tiles_copy != info.n_tiles || info.n_tiles == 0) continue;      // (the compiler agrees: copy tiles only)
      bool shared = false;                               // (an image drawn twice keeps its bitmap)
      for (size_t q = 0; q < bj.parts.size(); ++q) if (q != k && bj.parts[q].image == p.image) shared = true;
      if (shared) continue;
      direct[k] = 1;
      g_direct_images.fetch_add(1, std::memory_order_relaxed);
      img[static_cast<size_t>(p.image)] = d + o_canvas + static_cast<size_t>(p.Y0) * canvas_pitch + static_cast<size_t>(p.X0) * 4;
      dpitch[static_cast<size_t>(p.image)] = canvas_pitch;
      dsrc[static_cast<size_t>(p.image)] = img[static_cast<size_t>(p.image)];
    }
  }
  bool rendered_whole = false;
  // the export is about to read canvas rows [0, y_end) on `reader` (one of the encoder's two streams): order it behind the
  // render of those rows
  auto ordered = [&](hipStream_t reader, hipEvent_t ev) -> int {
    if (render == reader) return IST_OK;
    if (hipStreamWaitEvent(reader, ev, 0) != hipSuccess) { (void)hipGetLastError(); return fail(IST_E_HIP, "ordering the export behind the render failed"); }
    return IST_OK;
  };
  if (!ctx->render_done && hipEventCreateWithFlags(&ctx->render_done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->render_done = nullptr; return fail(IST_E_HIP, "hipEventCreate failed"); }
  // The FIRST request renders everything: behind the Huffman batch every image is reconstructed and its band rendered on the
  // render stream, an event per band (the images' upload events are free again by then).  History (rocprofv3 timelines,
  // profiles/r03_file_pipeline_kernels.txt is the last of them): (1) pulling band k+1 only when slab k+1 was about to be
  // submitted put its reconstruction in competition with slab k's compression, which fills every CU - the 40 us colour
  // kernel took 250 us and the slabs ran one after the other at half speed; (2) a render THREAD that submitted the bands while
  // the encoder's thread submitted slabs was slower still (7.2-7.4 ms against 6.6-7.0): the render kernels then trickled in
  // between the slabs' workgroups for 4.4 ms instead of 1.0.  The compressing kernel owns the chip while it runs; the only work
  // that really hides behind it is the file's trip over PCIe.
  size_t next_part = 0;
  auto need_rows = [&](int64_t y_end, void* reader_) -> int {
    hipStream_t reader = static_cast<hipStream_t>(reader_);
    if (!bj.ok) {
      if (!rendered_whole) {
        int rc2 = fd.finish(render);
        if (rc2) return rc2;
        rendered_whole = true;
        rc2 = ist_job_launch(whole, dsrc.data(), dpitch.data(), n, d + o_canvas, canvas_pitch, render);
        if (rc2) return rc2;
        if (hipEventRecord(ctx->render_done, render) != hipSuccess) { (void)hipGetLastError(); return fail(IST_E_HIP, "hipEventRecord failed"); }
      }
      return ordered(reader, ctx->render_done);
    }
    // first request (the short first slab): only the bands it reads, so that the file's first bytes are on their way while the
    // rest is rendered; every later request: everything that is left
    const bool first_request = next_part == 0;
    if (next_part < bj.parts.size()) mark(first_request ? "first rows requested" : "rest of the rows requested");
    while (next_part < bj.parts.size() && (!first_request || bj.parts[next_part].Y0 < y_end)) {
      const ist_part& p = bj.parts[next_part];
      int rc2 = fd.take(p.image, render);
      if (rc2) return rc2;
      if (!direct[next_part]) rc2 = ist_job_launch(bj.band[next_part], dsrc.data(), dpitch.data(), n, d + o_canvas, canvas_pitch, render);
      if (rc2) return rc2;
      if (hipEventRecord(ctx->img_event[next_part], render) != hipSuccess) { (void)hipGetLastError(); return fail(IST_E_HIP, "hipEventRecord failed"); }
      ++next_part;
    }
    if (first_request) mark("first band submitted (Huffman batch done)");
    // the last band that rows [0, y_end) touch (bands are sorted by Y0; the background launch precedes them all on the render
    // stream).  Rows that no band touches (a gap at the top) are ordered behind the background launch alone.
    size_t last = bj.parts.size();
    for (size_t k = 0; k < bj.parts.size(); ++k) if (bj.parts[k].Y0 < y_end) last = k;
    if (last < next_part) return ordered(reader, ctx->img_event[last]);
    if (hipEventRecord(ctx->render_done, render) != hipSuccess) { (void)hipGetLastError(); return fail(IST_E_HIP, "hipEventRecord failed"); }
    return ordered(reader, ctx->render_done);
  };
  if (ph.on) {                                            // phase timing: the whole canvas first, then the export
    rc = need_rows(out_plan->canvas_h, ctx->stream);
    if (rc) return rc;
    ph.lap(IST_PHASE_STITCH, "compile + stitch launches", ctx->stream);
  }
  // PNG export on the device; the file's slabs cross PCIe (the only D2H of the call) while later slabs compress and -
  // not timing - while later images decode
  int64_t len = 0;
  uint8_t* host = nullptr;
  int64_t hint_rows = 0;
  for (const ist_part& p : bj.parts) hint_rows = std::max<int64_t>(hint_rows, p.Y1 - p.Y0);
  mark("encoder entered");
  rc = png_to_host(ctx, d + o_canvas, canvas_pitch, out_plan->canvas_w, out_plan->canvas_h, d + o_png, &host, &len, need_rows, hint_rows);
  mark("encoder returned (file in host memory)");
  if (rc == IST_OK) rc = fd.finish(render);               // (images whose draw is clipped away entirely)
  (void)hipStreamSynchronize(render);                     // nothing of this call runs on when the arena is handed to the next
  mark("render stream idle");
  if (rc) { if (host) pool_give(host); (void)hipStreamSynchronize(ctx->stream); return rc; }
  ph.lap(IST_PHASE_PNG, "PNG encode (GPU) + D2H, overlapped", ctx->stream);
  ph.lap(IST_PHASE_D2H, "(D2H: inside the PNG phase)", nullptr);
  if (ph.print) std::fprintf(stderr, "[ist timing] %d of %d images decoded by the GPU entropy decoder\n", fd.gpu_decoded(), n);
  *out_png = host; *out_len = len;
  pg.keep = true;
  return IST_OK;
}
}  // namespace
}  // extern "C++"

int ist_stitch_paths_png(ist_ctx* ctx, const char* const* paths, int n_images, int direction, int mode, double gap, const ist_limits* limits,
                         int filter, ist_plan* out_plan, uint8_t** out_png, int64_t* out_len) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  if (!out_plan || !out_png || !out_len) return fail(IST_E_INVALID, "ist_stitch_paths_png: NULL output");
  *out_png = nullptr; *out_len = 0;
  std::memset(out_plan, 0, sizeof(*out_plan));
  if (n_images <= 0) return IST_NOTHING_TO_DO;
  if (!paths) return fail(IST_E_INVALID, "ist_stitch_paths_png: NULL input");
  if (n_images > kMaxImages) return fail(IST_E_UNSUPPORTED, "more than 128 images in one launch");
  // The files are READ into blocks the context keeps from call to call (grow-only up to kKeepFileBytes each), one parked
  // worker per file - not mapped: the decoders parse a file twice (headers for the arena layout, then the scan) and walk it
  // with plain loads, so a file that another process rewrites or truncates while it is mapped would change under them
  // (a different layout on the second read) or raise SIGBUS in a worker thread and take the host process down.
  // A short read - the file shrank between fstat and read - is an error of that image.
  tl_begin();
  struct TlEnd { ~TlEnd() { tl_end("ist_stitch_paths_png"); } } tl_end_guard;
  std::lock_guard<std::mutex> lock(ctx->mu);
  tl_mark("paths: context locked");
  const size_t n = static_cast<size_t>(n_images);
  constexpr size_t kKeepFileBytes = size_t{64} << 20;
  if (ctx->file_bufs.size() < n) ctx->file_bufs.resize(n);
  std::vector<int> fds(n, -1);
  std::vector<int64_t> lens(n, 0);
  std::vector<const uint8_t*> ptr(n, nullptr);
  std::vector<int> bad(n, 0);                             // 1: out of memory, 2: short read / read error
  struct Close {
    std::vector<int>& f; ist_ctx* c; size_t n;
    ~Close() {
      for (int d : f) if (d >= 0) (void)close(d);
      for (size_t i = 0; i < n && i < c->file_bufs.size(); ++i) if (c->file_bufs[i].capacity() > kKeepFileBytes) { ScanBuf none; c->file_bufs[i].swap(none); }
    }
  } closer{fds, ctx, n};
  for (size_t i = 0; i < n; ++i) {
    fds[i] = paths[i] ? open(paths[i], O_RDONLY | O_CLOEXEC) : -1;
    struct stat st;
    if (fds[i] < 0 || fstat(fds[i], &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0)
      return fail(IST_E_DECODE, "图片" + std::to_string(i) + "解码异常: " + (fds[i] < 0 ? "cannot open the file" : "empty file, or not a regular file"));
    // (an image file of more than 1 GiB is no photo: refuse it before a block of that size is allocated for it)
    if (st.st_size > (off_t{1} << 30)) return fail(IST_E_DECODE, "图片" + std::to_string(i) + "解码异常: the file is larger than 1 GiB");
    lens[i] = static_cast<int64_t>(st.st_size);
  }
  auto read_one = [&](int k) {
    const size_t i = static_cast<size_t>(k), want = static_cast<size_t>(lens[i]);
    ScanBuf& b = ctx->file_bufs[i];
    if (!b.reserve(want + want / 8 + 64)) { bad[i] = 1; return; }
    size_t got = 0;
    while (got < want) {
      const ssize_t r = pread(fds[i], b.data() + got, want - got, static_cast<off_t>(got));
      if (r < 0 && errno == EINTR) continue;
      if (r <= 0) { bad[i] = 2; return; }
      got += static_cast<size_t>(r);
    }
    b.set_size(want);
    ptr[i] = b.data();
  };
  if (n_images == 1) read_one(0);
  else {
    if (!ctx->workers) ctx->workers.reset(new WorkerPool());
    tl_mark("paths: files opened");
    ctx->workers->run(n_images, read_one);
    tl_mark("paths: read tasks handed out");
    ctx->workers->wait();
    tl_mark("paths: files read");
  }
  for (size_t i = 0; i < n; ++i) {
    if (bad[i] == 1) return fail(IST_E_NOMEM, "out of memory for the bytes of image " + std::to_string(i));
    if (bad[i]) return fail(IST_E_DECODE, "图片" + std::to_string(i) + "解码异常: the file changed while it was read");
  }
  return stitch_files_png_locked(ctx, ptr.data(), lens.data(), n_images, direction, mode, gap, limits, filter, out_plan, out_png, out_len);
}

// PNG of host pixels (H2D, encode, D2H)
int ist_png_encode_rgba8(ist_ctx* ctx, const uint8_t* pixels, size_t pitch, int64_t w, int64_t h, uint8_t** out_png,
                         int64_t* out_len) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  if (!pixels || !out_png || !out_len || w < 1 || h < 1 || pitch < static_cast<size_t>(w) * 4) return fail(IST_E_INVALID, "ist_png_encode_rgba8: bad argument");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard g(ctx->device);
  const size_t row = static_cast<size_t>(w) * 4;
  int rc = grow_device(&ctx->scratch_dst, &ctx->scratch_dst_bytes, row * static_cast<size_t>(h));
  if (rc) return rc;
  std::vector<RowsCopy> up{RowsCopy{ctx->scratch_dst, pixels, nullptr, pitch, row, static_cast<size_t>(h)}};
  rc = stager_of(ctx).upload(up, ctx->stream);
  if (rc) return rc;
  return png_to_host(ctx, ctx->scratch_dst, row, w, h, nullptr, out_png, out_len);
}

// The host paths with both directions of PCIe busy (round 4).  The canvas is cut into row bands (ist_shard_row_cuts: ~40 MB each, cuts on
// multiples of 8 rows); band b is the whole op list clipped to its rows, and ist_shard_parts (IST_SPLIT_ROWS) names the source rows it
// samples.  Band by band: the rows not yet on the device go up in 32 MiB pieces on the staging stream (Stager::upload_big), the band is
// launched behind them, and - ist_stitch_rgba8 - its rows go down into the pinned result on the aux stream while the next band's sources go
// up, or - ist_render_png / ist_stitch_png - the PNG encoder compresses it and sends its slabs down meanwhile.  Any layout shards this way: a
// vertical strip (index.js:1522-1538) sends image after image, a horizontal one (1540-1553) a slice of every image per band.  Upload-all,
// launch, download-all costs 8.1 + 7.5 ms for nine 12 MP images; overlapped the two directions hold 48 GB/s each (tools/exp/duplex2.cpp).
// An earlier banded attempt (round 2) sent the uploads as 4 MiB chunks on four streams, which collapses to 12.7 GB/s each way as soon as
// downloads are in flight (tools/exp/duplex.cpp) - the piece size was the problem, not the idea.
extern "C++" {
namespace {
struct RowBands {
  ist_ctx* ctx = nullptr;
  bool ok = false;                        // false after prepare(): not applicable (a small canvas, an op list the row cut refuses); nothing was queued
  int nb = 0, n_images = 0;
  int64_t cw = 0, ch = 0;
  size_t row = 0, total = 0;
  const ist_image_desc* images = nullptr;
  const uint8_t* const* src = nullptr;
  const size_t* src_pitch = nullptr;
  std::vector<int32_t> cuts;
  std::vector<ist_part> parts;
  std::vector<ist_job*> jobs;
  std::vector<const void*> dsrc;
  std::vector<size_t> dpitch;
  std::vector<int64_t> lo, hi;            // rows of image i already sent: [lo, hi)
  std::vector<RowsCopy> items;
  uint8_t* canvas = nullptr;
  ~RowBands() { for (ist_job* q : jobs) if (q) ist_job_destroy(q); }
  size_t bw(int i) const { return static_cast<size_t>(images[i].bmp_width > 0 ? images[i].bmp_width : images[i].width); }
  int64_t bh(int i) const { return static_cast<int64_t>(images[i].bmp_height > 0 ? images[i].bmp_height : images[i].height); }
  int64_t y0(int b) const { return cuts[static_cast<size_t>(b)]; }
  int64_t y1(int b) const { return cuts[static_cast<size_t>(b) + 1]; }

  int prepare(ist_ctx* c, int64_t canvas_w, int64_t canvas_h, const uint8_t clear[4], const ist_op* ops, int n_ops, const ist_image_desc* imgs,
              const uint8_t* const* s, const size_t* sp, int n, int filter) {
    static const bool off = tuning_mode() && std::getenv("IST_HOST_DUPLEX") && std::atoi(std::getenv("IST_HOST_DUPLEX")) == 0;
    ctx = c; cw = canvas_w; ch = canvas_h; images = imgs; src = s; src_pitch = sp; n_images = n;
    row = static_cast<size_t>(cw) * 4; total = row * static_cast<size_t>(ch);
    if (off || total < (32u << 20) || n_images < 1) return IST_OK;
    nb = static_cast<int>(std::min<size_t>(16, std::max<size_t>(2, total / (40u << 20))));
    cuts.assign(static_cast<size_t>(nb) + 1, 0);
    parts.resize(static_cast<size_t>(std::max(1, n_ops)) * static_cast<size_t>(nb) + 8);
    int n_parts = 0;
    {
      const std::string keep_msg = g_last_error;
      const int keep_code = g_last_code;
      if (ist_shard_row_cuts(ch, nb, cuts.data()) != IST_OK ||
          ist_shard_parts(ops, n_ops, cw, ch, images, n_images, filter, nb, IST_SPLIT_ROWS, parts.data(), static_cast<int>(parts.size()), &n_parts) != IST_OK) {
        g_last_error = keep_msg; g_last_code = keep_code;          // not an error of the call: the one-shot path takes it
        return IST_OK;
      }
    }
    parts.resize(static_cast<size_t>(n_parts));
    jobs.assign(static_cast<size_t>(nb), nullptr);
    for (int b = 0; b < nb; ++b) {
      if (y0(b) >= y1(b)) continue;
      const ist_region clip{0, static_cast<int32_t>(y0(b)), static_cast<int32_t>(cw), static_cast<int32_t>(y1(b) - y0(b))};
      jobs[static_cast<size_t>(b)] = ist_job_create(ctx, cw, ch, clear, ops, n_ops, images, n_images, filter, &clip);
      if (!jobs[static_cast<size_t>(b)]) return g_last_code ? g_last_code : IST_E_INVALID;
    }
    // device scratch: the images the bands draw (whole allocations, filled row range by row range), and the canvas
    std::vector<size_t> at(static_cast<size_t>(n_images), 0);
    std::vector<char> used(static_cast<size_t>(n_images), 0);
    lo.assign(static_cast<size_t>(n_images), -1); hi.assign(static_cast<size_t>(n_images), -1);
    for (const ist_part& p : parts) if (p.image >= 0 && p.image < n_images) used[static_cast<size_t>(p.image)] = 1;
    size_t src_bytes = 0;
    for (int i = 0; i < n_images; ++i) {
      if (!used[static_cast<size_t>(i)]) continue;
      if (!src || !src[i] || bw(i) < 1 || bh(i) < 1) return fail(IST_E_DECODE, "图片" + std::to_string(i) + "解码异常");
      if (src_pitch && src_pitch[i] < bw(i) * 4) return fail(IST_E_INVALID, "src_pitch too small");
      at[static_cast<size_t>(i)] = src_bytes;
      src_bytes += (bw(i) * 4 * static_cast<size_t>(bh(i)) + 255 + 256) & ~static_cast<size_t>(255);       // (+ a vector load's reach past the last row sent)
    }
    int rc = grow_device(&ctx->scratch_src, &ctx->scratch_src_bytes, src_bytes ? src_bytes : 256);
    if (rc) return rc;
    rc = grow_device(&ctx->scratch_dst, &ctx->scratch_dst_bytes, total);
    if (rc) return rc;
    if (!ctx->workers) ctx->workers.reset(new WorkerPool());
    dsrc.assign(static_cast<size_t>(n_images), nullptr);
    dpitch.assign(static_cast<size_t>(n_images), 0);
    for (int i = 0; i < n_images; ++i)
      if (used[static_cast<size_t>(i)]) { dsrc[static_cast<size_t>(i)] = static_cast<uint8_t*>(ctx->scratch_src) + at[static_cast<size_t>(i)]; dpitch[static_cast<size_t>(i)] = bw(i) * 4; }
    canvas = static_cast<uint8_t*>(ctx->scratch_dst);
    ok = true;
    return IST_OK;
  }

  // sends what band b still needs and launches it, all ordered on R
  int submit(int b, hipStream_t R) {
    items.clear();
    auto send = [&](int i, int64_t r0, int64_t r1) {            // rows [r0, r1) of image i
      if (r1 <= r0) return;
      const size_t hp = src_pitch ? src_pitch[i] : bw(i) * 4;
      items.push_back(RowsCopy{static_cast<uint8_t*>(const_cast<void*>(dsrc[static_cast<size_t>(i)])) + static_cast<size_t>(r0) * bw(i) * 4,
                               src[i] + static_cast<size_t>(r0) * hp, nullptr, hp, bw(i) * 4, static_cast<size_t>(r1 - r0)});
    };
    for (const ist_part& p : parts) {
      if (p.slot != b || p.image < 0) continue;
      const int i = p.image;
      const int64_t a = std::max<int64_t>(0, p.sy0), e = std::min<int64_t>(bh(i), p.sy1);
      if (e <= a) continue;
      int64_t& L0 = lo[static_cast<size_t>(i)]; int64_t& H0 = hi[static_cast<size_t>(i)];
      if (L0 < 0) { send(i, a, e); L0 = a; H0 = e; }
      else {                                                     // keep ONE interval per image: a band further down extends it (rows between are sent too)
        if (a < L0) { send(i, a, L0); L0 = a; }
        if (e > H0) { send(i, H0, e); H0 = e; }
      }
    }
    if (!items.empty()) { const int rc = stager_of(ctx).upload_big(items, R, ctx->workers.get()); if (rc) return rc; }
    return ist_job_launch(jobs[static_cast<size_t>(b)], dsrc.data(), dpitch.data(), n_images, canvas, row, R);
  }
};
}  // namespace
}  // extern "C++"

// *done = false: not applicable, nothing was queued and the caller takes the one-shot path.
static int stitch_banded_duplex(ist_ctx* ctx, const ist_plan* plan, const ist_op* ops, int n_ops, const ist_image_desc* images,
                                const uint8_t* const* src, const size_t* src_pitch, int n_images, int filter, uint8_t** out_pixels, bool* done) {
  *done = false;
  static const uint8_t transparent[4] = {0, 0, 0, 0};
  static const bool print = std::getenv("IST_TIMING") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) { if (print) std::fprintf(stderr, "[ist timing] host stitch: %-34s at %7.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count()); };
  RowBands rb;
  int rc = rb.prepare(ctx, plan->canvas_w, plan->canvas_h, transparent, ops, n_ops, images, src, src_pitch, n_images, filter);
  if (rc) return rc;
  if (!rb.ok) return IST_OK;
  rc = ensure_aux(ctx);
  if (rc) return rc;
  lap("band jobs compiled, scratch");
  uint8_t* host = static_cast<uint8_t*>(pool_take(rb.total));
  if (!host) return fail(IST_E_NOMEM, "out of pinned host memory for the result");
  std::vector<hipEvent_t> ev(static_cast<size_t>(rb.nb), nullptr);
  hipStream_t R = ctx->stream, D = ctx->aux;
  // (whatever happens below, the streams are idle before the pinned block or the jobs' tables are given back)
  auto finish = [&](int code) {
    (void)hipStreamSynchronize(R); (void)hipStreamSynchronize(D); (void)stager_of(ctx).sync();
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    if (code != IST_OK) pool_give(host);
    return code;
  };
  bool first = true;
  for (int b = 0; b < rb.nb; ++b) {
    const int64_t y0 = rb.y0(b), y1 = rb.y1(b);
    if (y0 >= y1) continue;
    rc = rb.submit(b, R);
    if (rc) return finish(rc);
    hipEvent_t& e = ev[static_cast<size_t>(b)];
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, R) != hipSuccess || hipStreamWaitEvent(D, e, 0) != hipSuccess ||
        hipMemcpyAsync(host + static_cast<size_t>(y0) * rb.row, rb.canvas + static_cast<size_t>(y0) * rb.row, static_cast<size_t>(y1 - y0) * rb.row, hipMemcpyDeviceToHost, D) != hipSuccess) {
      (void)hipGetLastError();
      return finish(fail(IST_E_HIP, "queueing a band's readback failed"));
    }
    if (first) { lap("first band queued"); first = false; }
  }
  lap("last band queued");
  if (hipStreamSynchronize(D) != hipSuccess || hipStreamSynchronize(R) != hipSuccess) { (void)hipGetLastError(); return finish(fail(IST_E_HIP, "result readback failed")); }
  lap("last band in host memory");
  g_duplex_stitches.fetch_add(1, std::memory_order_relaxed);
  *out_pixels = host;
  *done = true;
  return finish(IST_OK);
}

static int render_png_banded(ist_ctx* ctx, int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4], const ist_op* ops, int n_ops,
                             const ist_image_desc* images, const uint8_t* const* src, const size_t* src_pitch, int n_images, int filter,
                             uint8_t** out_png, int64_t* out_len) {
  static const uint8_t transparent[4] = {0, 0, 0, 0};
  RowBands rb;
  int rc = rb.prepare(ctx, canvas_w, canvas_h, clear_rgba ? clear_rgba : transparent, ops, n_ops, images, src, src_pitch, n_images, filter);
  if (rc) return rc;
  if (!rb.ok) return 1;
  if (!ctx->render) {                                  // (high priority: its short kernels should not queue behind the encoder's thousands of workgroups)
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&ctx->render, hipStreamNonBlocking, hi) != hipSuccess) { (void)hipGetLastError(); ctx->render = nullptr; return fail(IST_E_HIP, "hipStreamCreate failed"); }
  }
  hipStream_t R = ctx->render;
  std::vector<hipEvent_t> ev(static_cast<size_t>(rb.nb), nullptr);
  int next = 0, failed = IST_OK;
  // the encoder is about to read canvas rows [0, y_end) on `reader`: submit the bands they lie in, order the reader behind the last of them
  auto need_rows = [&](int64_t y_end, void* reader_) -> int {
    hipStream_t reader = static_cast<hipStream_t>(reader_);
    int last = -1;
    for (int b = 0; b < rb.nb; ++b) {
      if (rb.y0(b) >= rb.y1(b)) continue;
      if (rb.y0(b) >= y_end) break;
      if (b >= next) {
        const int rc2 = rb.submit(b, R);
        if (rc2) { failed = rc2; return rc2; }
        if (hipEventCreateWithFlags(&ev[static_cast<size_t>(b)], hipEventDisableTiming) != hipSuccess || hipEventRecord(ev[static_cast<size_t>(b)], R) != hipSuccess) {
          (void)hipGetLastError(); failed = IST_E_HIP; return fail(IST_E_HIP, "hipEventRecord failed");
        }
        next = b + 1;
      }
      last = b;
    }
    if (last >= 0 && hipStreamWaitEvent(reader, ev[static_cast<size_t>(last)], 0) != hipSuccess) { (void)hipGetLastError(); failed = IST_E_HIP; return fail(IST_E_HIP, "ordering the export behind the render failed"); }
    return IST_OK;
  };
  int64_t hint = 0;
  for (int b = 0; b < rb.nb; ++b) hint = std::max<int64_t>(hint, rb.y1(b) - rb.y0(b));
  rc = png_to_host(ctx, rb.canvas, rb.row, canvas_w, canvas_h, nullptr, out_png, out_len, need_rows, hint);
  (void)hipStreamSynchronize(R); (void)stager_of(ctx).sync(); (void)hipStreamSynchronize(ctx->stream);
  for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
  if (rc == IST_OK) g_duplex_stitches.fetch_add(1, std::memory_order_relaxed);
  (void)failed;
  return rc;
}

int64_t ist_debug_duplex_stitches(void) { return g_duplex_stitches.load(); }

int ist_stitch_rgba8(ist_ctx* ctx, const ist_image_desc* images, const uint8_t* const* src, const size_t* src_pitch,
                     int n_images, int direction, int mode, double gap, const ist_limits* limits, int filter,
                     ist_plan* out_plan, uint8_t** out_pixels) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  if (!out_plan || !out_pixels) return fail(IST_E_INVALID, "ist_stitch_rgba8: NULL output");
  *out_pixels = nullptr;
  ist_limits lim;
  if (limits) lim = *limits; else ist_limits_unlimited(&lim);
  int rc = ist_plan_compute(images, n_images, direction, mode, gap, &lim, out_plan);
  if (rc != IST_OK) return rc;
  std::vector<ist_op> ops(static_cast<size_t>(out_plan->n_rects) + 1);
  int n_ops = 0;
  rc = ist_plan_ops(out_plan, images, n_images, ops.data(), &n_ops);
  if (rc != IST_OK) { ist_plan_free(out_plan); return rc; }
  static const uint8_t transparent[4] = {0, 0, 0, 0};
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    DeviceGuard g(ctx->device);
    bool done = false;
    rc = stitch_banded_duplex(ctx, out_plan, ops.data(), n_ops, images, src, src_pitch, n_images, filter, out_pixels, &done);
    if (rc == IST_OK && !done) {
      rc = render_to_scratch(ctx, out_plan->canvas_w, out_plan->canvas_h, transparent, ops.data(), n_ops, images, src, src_pitch,
                             n_images, filter, nullptr, nullptr, nullptr);
      // the export (index.js:1577-1579): the whole canvas in one DMA into a pinned block of the pool
      if (rc == IST_OK)
        rc = read_back_pooled(ctx->scratch_dst, static_cast<size_t>(out_plan->canvas_w) * 4 * static_cast<size_t>(out_plan->canvas_h), ctx->stream, out_pixels);
    }
  }
  if (rc != IST_OK) { ist_plan_free(out_plan); return rc; }
  return IST_OK;
}

// buffers handed out by the library: pinned blocks go back to the pool, anything else was malloc'ed
void ist_free(void* p) {
  if (!p) return;
  if (!pool_give(p)) std::free(p);
}

void ist_pool_trim(void) { pool_trim(); }

}  // extern "C"
