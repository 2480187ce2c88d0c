// ist_mgpu.cpp — one stitch on a GROUP of GPUs from ONE process (the N-API host's way in): parts rendered on their
// devices, finished bands gathered into the root's canvas with one grouped RCCL send/recv batch over xGMI.
//
// Reference anchor: the per-image loop of onStitch (pages/index/index.js:1439-1554) — independent iterations; the seam
// the mini-program would call is onStitchVertical/Horizontal -> onStitch (index.js:771-788 -> :1186).  BASELINE.json
// north_star: "partitioned across the 8 GPUs of one node by assigning disjoint input-image subsets per GPU with a single
// RCCL gather over xGMI".  The one-process-per-GPU variant of the same layout is imagestitching_amd/dist.py (torchrun);
// both cut the job with ist_shard_parts.
//
// Layout of a launch (slot 0 = the root; a device may serve several slots):
//   * parts of slot 0 are cells of the root's own fused launch;
//   * every other part has a band job (white fill + that draw, clipped to the part's box).  On the root's DEVICE it
//     renders straight into the canvas; on another device into a compact band, which ncclSend delivers to the root:
//     full-width boxes are contiguous canvas bytes and are received IN PLACE (the root launch carries a HOLE there), other
//     boxes go to a staging band that a 1:1 placement launch copies in behind its receive (same stream);
//   * one ncclGroupStart/ncclGroupEnd holds every send and receive of the step (RCCL has no gatherv).  xGMI is point to
//     point: each sender has its own link to the root, so there is no ring and no tree to build.
// RCCL is dlopen'ed on first use (librccl.so.1): hosts that stay on one GPU never load it.
//
// UNITS.  What a slot renders into one buffer and delivers in one piece is a UNIT: under IST_SPLIT_IMAGE / IST_SPLIT_BAND a unit
// is one part (a box of one draw); under IST_SPLIT_ROWS it is the slot's whole band of canvas rows - full width whatever the
// layout (horizontal strips, index.js:1540-1553; centred rects) - rendered from the whole op list clipped to those rows, so
// every unit is received in place: no staging band, no placement launch, and the host sink below is always available.
//
// HOST SINK (ist_group_stitch_rgba8 on a strip of full-width bands): the export of the reference is host-destined
// (index.js:1577-1581, utils/canvas.js:205-242), so nothing has to meet on one GPU: every device uploads only the rows its
// parts sample over its own PCIe link, renders them into compact bands and DMAs each finished band STRAIGHT into its byte
// range of the pooled pinned result (a full-width band is contiguous there) - no xGMI gather and no 439 MB readback over
// the root's single link.  The root's launch supplies everything that is not a band (background rows, gaps).  Per-draw cuts
// of strips whose boxes are not full-width (horizontal, centred) keep the gather + root readback; IST_SPLIT_ROWS - what
// IST_SPLIT_AUTO picks for them - does not.
//
// Nothing is allocated per call after the first: band / staging buffers are grow-only arenas of the group (jobs hold
// offsets; streams are in order, so jobs may share them), and the one-call host path keeps its compiled group jobs in a
// small LRU keyed by (canvas, ops, images, filter, split).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>

#include "ist_ctx.h"
#include "ist_internal.h"

using namespace ist;

namespace {

// ------------------------------------------------------------------------------------------------ RCCL, late bound
typedef struct ncclComm* ncclComm_t;
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string error;
};
std::atomic<int64_t> g_host_sink_stitches{0};
constexpr int kNcclUint8 = 1;        // ncclDataType_t::ncclUint8 (rccl.h)

Rccl* rccl() {
  static Rccl* R = [] {
    Rccl* r = new Rccl;
    // (IST_TUNING=1 IST_RCCL_UNAVAILABLE=1, tests only: behave as on a host where librccl cannot be loaded)
    if (tuning_mode() && std::getenv("IST_RCCL_UNAVAILABLE")) { r->error = "librccl.so.1 could not be loaded: disabled by IST_RCCL_UNAVAILABLE"; return r; }
    // the copy already in the process first (a PyTorch host carries its own librccl.so bound to ITS HIP runtime)
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) if (!r->lib) r->lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char* paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : paths) if (!r->lib) r->lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!r->lib) { r->error = std::string("librccl.so.1 could not be loaded: ") + (dlerror() ? dlerror() : "?"); return r; }
    auto sym = [&](const char* n) { void* p = dlsym(r->lib, n); if (!p && r->error.empty()) r->error = std::string("librccl: missing symbol ") + n; return p; };
    r->CommInitAll = reinterpret_cast<decltype(r->CommInitAll)>(sym("ncclCommInitAll"));
    r->CommDestroy = reinterpret_cast<decltype(r->CommDestroy)>(sym("ncclCommDestroy"));
    r->GroupStart = reinterpret_cast<decltype(r->GroupStart)>(sym("ncclGroupStart"));
    r->GroupEnd = reinterpret_cast<decltype(r->GroupEnd)>(sym("ncclGroupEnd"));
    r->Send = reinterpret_cast<decltype(r->Send)>(sym("ncclSend"));
    r->Recv = reinterpret_cast<decltype(r->Recv)>(sym("ncclRecv"));
    r->GetErrorString = reinterpret_cast<decltype(r->GetErrorString)>(sym("ncclGetErrorString"));
    return r;
  }();
  return R;
}

int nccl_fail(const char* what, int rc) {
  Rccl* R = rccl();
  return fail(IST_E_HIP, std::string(what) + ": " + (R->GetErrorString ? R->GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}

}  // namespace

// ------------------------------------------------------------------------------------------------ handles
struct ist_group {
  std::vector<int> slot_dev;              // device of every slot, as the caller listed them
  std::vector<int> devs;                  // distinct devices, devs[0] = the root's
  std::vector<int> slot_rank;             // slot -> index into devs (= RCCL rank)
  std::vector<ist_ctx*> ctx;              // one context per distinct device (its stream carries the renders and sends)
  hipStream_t recv_stream = nullptr;      // root device: receives + placement launches, beside the root's own launch
  std::vector<ncclComm_t> comm;           // one communicator handle per distinct device; empty while the group is one device
  bool self_send = false;                 // test knob (IST_TUNING=1 IST_GROUP_SELF_SEND=1): same-device bands also travel through RCCL
  std::mutex mu;
  // grow-only arenas (jobs hold offsets into them): compact bands per device, staging bands on the root
  std::vector<void*> band_arena; std::vector<size_t> band_bytes;
  void* staging = nullptr; size_t staging_bytes = 0;
  // compiled jobs of the one-call host path, most recently used last
  struct Cached { std::string key; ist_group_job* job; };
  std::vector<Cached> cache;
  static constexpr size_t kCacheJobs = 4;
};

struct ist_group_job {
  ist_group* g = nullptr;
  int64_t cw = 0, ch = 0;
  int n_images = 0;
  struct PartRt {
    ist_part part;
    int rank = 0;                         // device index of the owner
  };
  std::vector<PartRt> parts;
  // what a non-root slot renders into ONE buffer and delivers in one piece: a part (IMAGE / BAND) or the slot's band (ROWS)
  struct Unit {
    int slot = 0, rank = 0;
    int32_t X0 = 0, Y0 = 0, X1 = 0, Y1 = 0;   // canvas box
    bool in_place = false;                // full canvas width: a contiguous byte range of the canvas
    bool local = false;                   // device sink: rendered on the root's device straight into the canvas
    std::vector<size_t> part_idx;         // the parts whose source pointers the unit's job reads
    ist_job* band_job = nullptr;
    size_t band_off = 0;                  // compact band in the owner's arena
    ist_job* place_job = nullptr;         // root: staged band -> canvas
    size_t staging_off = 0;               // root arena (units that are not full-width)
  };
  std::vector<Unit> units;
  int split = IST_SPLIT_IMAGE;            // the effective cut (AUTO resolved)
  ist_job* root_job = nullptr;
  std::vector<size_t> band_need;          // per device: bytes of band arena this job addresses
  size_t staging_need = 0;
  bool host_sink_ok = false;              // every band is full-width: bands can be DMA'ed straight into a host canvas
  std::vector<std::pair<int64_t, int64_t>> root_rows;   // host sink: canvas row ranges the root's launch delivers
};

namespace {

void group_job_free(ist_group_job* j) {
  if (!j) return;
  for (auto& u : j->units) {
    if (u.band_job) ist_job_destroy(u.band_job);
    if (u.place_job) ist_job_destroy(u.place_job);
  }
  if (j->root_job) ist_job_destroy(j->root_job);
  delete j;
}

inline size_t unit_bytes(const ist_group_job::Unit& u) { return static_cast<size_t>(u.X1 - u.X0) * 4 * static_cast<size_t>(u.Y1 - u.Y0); }
inline size_t round256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

int group_sync_locked(ist_group* g) {
  for (size_t r = 0; r < g->devs.size(); ++r) {
    DeviceGuard dg(g->devs[r]);
    if (hipStreamSynchronize(g->ctx[r]->stream) != hipSuccess) return fail(IST_E_HIP, "hipStreamSynchronize failed");
  }
  DeviceGuard dg(g->devs[0]);
  if (hipStreamSynchronize(g->recv_stream) != hipSuccess) return fail(IST_E_HIP, "hipStreamSynchronize failed");
  return IST_OK;
}

// the arenas cover what `job` addresses; growing one waits for the group's streams first (earlier launches may still use it)
int ensure_arenas(ist_group* g, const ist_group_job* job) {
  bool grow = job->staging_need > g->staging_bytes;
  for (size_t r = 0; r < g->devs.size(); ++r) grow = grow || job->band_need[r] > g->band_bytes[r];
  if (!grow) return IST_OK;
  int rc = group_sync_locked(g);
  if (rc) return rc;
  for (size_t r = 0; r < g->devs.size(); ++r) {
    if (job->band_need[r] <= g->band_bytes[r]) continue;
    DeviceGuard dg(g->devs[r]);
    rc = grow_device(&g->band_arena[r], &g->band_bytes[r], job->band_need[r]);
    if (rc) return fail(IST_E_NOMEM, "out of device memory for the bands of device " + std::to_string(g->devs[r]));
  }
  if (job->staging_need > g->staging_bytes) {
    DeviceGuard dg(g->devs[0]);
    rc = grow_device(&g->staging, &g->staging_bytes, job->staging_need);
    if (rc) return fail(IST_E_NOMEM, "out of device memory for the staging bands");
  }
  return IST_OK;
}

}  // namespace

extern "C" {

ist_group* ist_group_create(const int* devices, int ndev) {
  if (!devices || ndev < 1 || ndev > 64) { fail(IST_E_INVALID, "ist_group_create: device list must hold 1..64 entries"); return nullptr; }
  const int have = ist_device_count();
  if (have <= 0) { fail(IST_E_NO_DEVICE, "no HIP device: the stitch path has no CPU fallback"); return nullptr; }
  std::unique_ptr<ist_group> g(new ist_group);
  for (int s = 0; s < ndev; ++s) {
    if (devices[s] < 0 || devices[s] >= have) { fail(IST_E_INVALID, "ist_group_create: device " + std::to_string(devices[s]) + " does not exist (" + std::to_string(have) + " visible)"); return nullptr; }
    g->slot_dev.push_back(devices[s]);
    size_t r = 0;
    while (r < g->devs.size() && g->devs[r] != devices[s]) ++r;
    if (r == g->devs.size()) g->devs.push_back(devices[s]);
    g->slot_rank.push_back(static_cast<int>(r));
  }
  struct Undo { ist_group* g; bool keep = false; ~Undo() { if (!keep) for (ist_ctx* c : g->ctx) ist_ctx_destroy(c); } } undo{g.get()};
  for (int d : g->devs) {
    ist_ctx* c = ist_ctx_create(d);
    if (!c) return nullptr;
    g->ctx.push_back(c);
  }
  g->band_arena.assign(g->devs.size(), nullptr);
  g->band_bytes.assign(g->devs.size(), 0);
  {
    DeviceGuard dg(g->devs[0]);
    if (hipStreamCreateWithFlags(&g->recv_stream, hipStreamNonBlocking) != hipSuccess) { fail(IST_E_HIP, "hipStreamCreate failed"); return nullptr; }
  }
  g->self_send = tuning_mode() && std::getenv("IST_GROUP_SELF_SEND") != nullptr;
  // (the RCCL communicators are made on the first launch that gathers over xGMI - ensure_rccl: the host-sink path never
  // exchanges anything between GPUs, and must not depend on librccl being loadable)
  undo.keep = true;
  return g.release();
}

void ist_group_destroy(ist_group* g) {
  if (!g) return;
  (void)group_sync_locked(g);
  for (auto& c : g->cache) group_job_free(c.job);
  g->cache.clear();
  for (ncclComm_t c : g->comm) if (c) (void)rccl()->CommDestroy(c);
  for (size_t r = 0; r < g->devs.size(); ++r) if (g->band_arena[r]) { DeviceGuard dg(g->devs[r]); dev_free(g->band_arena[r]); }
  if (g->staging) { DeviceGuard dg(g->devs[0]); dev_free(g->staging); }
  if (g->recv_stream) { DeviceGuard dg(g->devs[0]); (void)hipStreamDestroy(g->recv_stream); }
  for (ist_ctx* c : g->ctx) ist_ctx_destroy(c);
  delete g;
}

int64_t ist_debug_host_sink_stitches(void) { return g_host_sink_stitches.load(std::memory_order_relaxed); }

int ist_group_slots(const ist_group* g) { return g ? static_cast<int>(g->slot_dev.size()) : 0; }
int ist_group_device(const ist_group* g, int slot) {
  if (!g || slot < 0 || slot >= static_cast<int>(g->slot_dev.size())) return -1;
  return g->slot_dev[static_cast<size_t>(slot)];
}

ist_group_job* ist_group_job_create(ist_group* g, int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4], const ist_op* ops,
                                    int n_ops, const ist_image_desc* images, int n_images, int filter, int split) {
  if (!g) { fail(IST_E_NO_CONTEXT, "无法获取绘图上下文"); return nullptr; }
  if (!ops || n_ops < 1 || !images || n_images < 1) { fail(IST_E_INVALID, "ist_group_job_create: empty op list"); return nullptr; }
  const int n_slots = static_cast<int>(g->slot_dev.size());
  split = ist_shard_resolve(ops, n_ops, canvas_w, canvas_h, images, n_images, filter, split);
  if (split < 0) return nullptr;
  const bool by_rows = split == IST_SPLIT_ROWS;
  std::vector<ist_part> cut(by_rows ? static_cast<size_t>(n_ops) * static_cast<size_t>(n_slots) + 8 : static_cast<size_t>(n_ops) + static_cast<size_t>(n_slots) + 8);
  int n_parts = 0;
  if (ist_shard_parts(ops, n_ops, canvas_w, canvas_h, images, n_images, filter, n_slots, split, cut.data(), static_cast<int>(cut.size()), &n_parts) != IST_OK)
    return nullptr;
  std::unique_ptr<ist_group_job, void (*)(ist_group_job*)> job(new ist_group_job, group_job_free);
  job->g = g; job->cw = canvas_w; job->ch = canvas_h; job->n_images = n_images; job->split = split;
  job->band_need.assign(g->devs.size(), 0);
  static const uint8_t transparent[4] = {0, 0, 0, 0};
  const uint8_t* clear = clear_rgba ? clear_rgba : transparent;
  for (int k = 0; k < n_parts; ++k) {
    ist_group_job::PartRt rt;
    rt.part = cut[k];
    rt.rank = g->slot_rank[static_cast<size_t>(cut[k].slot)];
    job->parts.push_back(rt);
  }
  // units of the non-root slots
  auto new_unit = [&](int slot, int32_t X0, int32_t Y0, int32_t X1, int32_t Y1) -> ist_group_job::Unit& {
    ist_group_job::Unit u;
    u.slot = slot; u.rank = g->slot_rank[static_cast<size_t>(slot)];
    u.X0 = X0; u.Y0 = Y0; u.X1 = X1; u.Y1 = Y1;
    u.in_place = X0 == 0 && static_cast<int64_t>(X1) == canvas_w;
    u.local = u.rank == 0 && !g->self_send;
    job->units.push_back(u);
    return job->units.back();
  };
  if (by_rows) {
    std::vector<int32_t> cuts(static_cast<size_t>(n_slots) + 1);
    if (ist_shard_row_cuts(canvas_h, n_slots, cuts.data()) != IST_OK) return nullptr;
    for (int sl = 1; sl < n_slots; ++sl) {
      if (cuts[static_cast<size_t>(sl) + 1] <= cuts[static_cast<size_t>(sl)]) continue;
      ist_group_job::Unit& u = new_unit(sl, 0, cuts[static_cast<size_t>(sl)], static_cast<int32_t>(canvas_w), cuts[static_cast<size_t>(sl) + 1]);
      for (size_t k = 0; k < job->parts.size(); ++k) if (job->parts[k].part.slot == sl) u.part_idx.push_back(k);
    }
  } else {
    for (size_t k = 0; k < job->parts.size(); ++k) {
      const ist_part& p = job->parts[k].part;
      if (p.slot == 0) continue;
      new_unit(p.slot, p.X0, p.Y0, p.X1, p.Y1).part_idx.push_back(k);
    }
  }
  // the root's own launch: every op that is not a sharded draw (fills), the draws slot 0 owns a part of, and - listed
  // last, so that nothing lies on top of them - a HOLE over every unit someone else writes
  std::vector<ist_op> root_ops;
  std::vector<char> root_draw(static_cast<size_t>(n_ops), 0);
  for (int k = 0; k < n_parts; ++k) if (cut[k].slot == 0) root_draw[static_cast<size_t>(cut[k].op)] = 1;
  for (int k = 0; k < n_ops; ++k)
    if (ops[k].kind != IST_OP_DRAW || root_draw[static_cast<size_t>(k)]) root_ops.push_back(ops[k]);
  // (a draw that shards into nothing draws nothing: dropping it changes no pixel)
  job->host_sink_ok = true;
  for (auto& u : job->units) {
    if (!u.in_place) job->host_sink_ok = false;
    u.band_off = job->band_need[static_cast<size_t>(u.rank)];
    job->band_need[static_cast<size_t>(u.rank)] += round256(unit_bytes(u));
    ist_op hole;
    std::memset(&hole, 0, sizeof hole);
    hole.kind = IST_OP_HOLE; hole.image = -1;
    hole.m[0] = 1.0; hole.m[3] = 1.0;
    hole.d[0] = u.X0; hole.d[1] = u.Y0; hole.d[2] = u.X1 - u.X0; hole.d[3] = u.Y1 - u.Y0;
    root_ops.push_back(hole);
  }
  job->root_job = ist_job_create(g->ctx[0], canvas_w, canvas_h, clear, root_ops.data(), static_cast<int>(root_ops.size()), images, n_images, filter, nullptr);
  if (!job->root_job) return nullptr;
  // host sink: the canvas rows that no unit delivers come from the root's launch (the complement of the units' rows)
  if (job->host_sink_ok) {
    std::vector<std::pair<int64_t, int64_t>> holes;
    for (const auto& u : job->units) holes.emplace_back(u.Y0, u.Y1);
    std::sort(holes.begin(), holes.end());
    int64_t y = 0;
    for (const auto& h : holes) {
      if (h.first > y) job->root_rows.emplace_back(y, h.first);
      y = std::max(y, h.second);
    }
    if (y < canvas_h) job->root_rows.emplace_back(y, canvas_h);
  }
  // the first fill of the list paints the background of every per-draw band (index.js:1423-1424)
  int fill_at = -1;
  for (int k = 0; k < n_ops && fill_at < 0; ++k) if (ops[k].kind == IST_OP_FILL) fill_at = k;
  std::vector<ist_op> band_ops;
  for (auto& u : job->units) {
    band_ops.clear();
    if (by_rows) {                       // the whole op list, minus the draws that do not reach these rows, in canvas order
      std::vector<char> mine(static_cast<size_t>(n_ops), 0);
      for (size_t k : u.part_idx) mine[static_cast<size_t>(job->parts[k].part.op)] = 1;
      for (int k = 0; k < n_ops; ++k) if (ops[k].kind != IST_OP_DRAW || mine[static_cast<size_t>(k)]) band_ops.push_back(ops[k]);
    } else {
      const ist_part& p = job->parts[u.part_idx[0]].part;
      if (fill_at >= 0 && fill_at < p.op) band_ops.push_back(ops[fill_at]);
      band_ops.push_back(ops[p.op]);
    }
    const ist_region clip{u.X0, u.Y0, u.X1 - u.X0, u.Y1 - u.Y0};
    u.band_job = ist_job_create(g->ctx[static_cast<size_t>(u.rank)], canvas_w, canvas_h, clear, band_ops.data(), static_cast<int>(band_ops.size()), images, n_images, filter, &clip);
    if (!u.band_job) return nullptr;
    if (u.local || u.in_place) continue;
    // staged: received into a compact band on the root, then placed by a 1:1 draw clipped to the box
    u.staging_off = job->staging_need;
    job->staging_need += round256(unit_bytes(u));
    ist_op put;
    std::memset(&put, 0, sizeof put);
    put.kind = IST_OP_DRAW; put.image = 0;
    put.m[0] = 1.0; put.m[3] = 1.0;
    put.s[2] = u.X1 - u.X0; put.s[3] = u.Y1 - u.Y0;
    put.d[0] = u.X0; put.d[1] = u.Y0; put.d[2] = u.X1 - u.X0; put.d[3] = u.Y1 - u.Y0;
    ist_image_desc band_desc;
    std::memset(&band_desc, 0, sizeof band_desc);
    band_desc.width = u.X1 - u.X0; band_desc.height = u.Y1 - u.Y0; band_desc.orientation = 1; band_desc.opaque = 1;
    u.place_job = ist_job_create(g->ctx[0], canvas_w, canvas_h, clear, &put, 1, &band_desc, 1, IST_FILTER_NEAREST, &clip);
    if (!u.place_job) return nullptr;
  }
  return job.release();
}

void ist_group_job_destroy(ist_group_job* job) { group_job_free(job); }

int ist_group_job_parts(const ist_group_job* job, ist_part* parts, int max_parts, int* n_parts) {
  if (!job || !n_parts) return fail(IST_E_INVALID, "ist_group_job_parts: NULL argument");
  *n_parts = static_cast<int>(job->parts.size());
  if (!parts) return IST_OK;
  if (max_parts < *n_parts) return fail(IST_E_INVALID, "ist_group_job_parts: part table too small");
  for (size_t k = 0; k < job->parts.size(); ++k) parts[k] = job->parts[k].part;
  return IST_OK;
}

}  // extern "C"

namespace {

// one unit of a non-root slot into `to` (a compact band, or - biased by the caller - the canvas itself), on its owner's stream
int launch_band(ist_group_job* job, size_t ui, const void* const* src, const size_t* src_pitch, void* to, size_t to_pitch, bool compact) {
  auto& u = job->units[ui];
  const size_t ni = static_cast<size_t>(job->n_images);
  std::vector<const void*> one(ni, nullptr);
  std::vector<size_t> one_pitch(ni, 0);
  for (size_t k : u.part_idx) {
    const ist_part& p = job->parts[k].part;
    if (!src[k]) return fail(IST_E_DECODE, "图片" + std::to_string(p.image) + "解码异常");
    one[static_cast<size_t>(p.image)] = src[k];
    one_pitch[static_cast<size_t>(p.image)] = src_pitch ? src_pitch[k] : 0;
  }
  void* dst = to;
  if (compact) dst = reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(to) - (static_cast<uintptr_t>(u.Y0) * to_pitch + static_cast<uintptr_t>(u.X0) * 4));
  return ist_job_launch(u.band_job, one.data(), src_pitch ? one_pitch.data() : nullptr, job->n_images, dst, to_pitch,
                        job->g->ctx[static_cast<size_t>(u.rank)]->stream);
}

int launch_root(ist_group_job* job, const void* const* src, const size_t* src_pitch, void* dst, size_t dst_pitch) {
  const size_t ni = static_cast<size_t>(job->n_images);
  std::vector<const void*> one(ni, nullptr);
  std::vector<size_t> one_pitch(ni, 0);
  for (size_t k = 0; k < job->parts.size(); ++k) {
    const ist_part& p = job->parts[k].part;
    if (p.slot != 0) continue;
    if (!src[k]) return fail(IST_E_DECODE, "图片" + std::to_string(p.image) + "解码异常");
    one[static_cast<size_t>(p.image)] = src[k];
    one_pitch[static_cast<size_t>(p.image)] = src_pitch ? src_pitch[k] : 0;
  }
  return ist_job_launch(job->root_job, one.data(), src_pitch ? one_pitch.data() : nullptr, job->n_images, dst, dst_pitch, job->g->ctx[0]->stream);
}

// the communicators of the group (one per distinct device), on first need.  Caller holds g->mu.
int ensure_rccl(ist_group* g) {
  if (!g->comm.empty()) return IST_OK;
  Rccl* R = rccl();
  if (!R->error.empty()) return fail(IST_E_NO_DEVICE, "gathering bands over xGMI needs RCCL: " + R->error);
  g->comm.assign(g->devs.size(), nullptr);
  const int rc = R->CommInitAll(g->comm.data(), static_cast<int>(g->devs.size()), g->devs.data());
  if (rc != 0) { g->comm.clear(); return nccl_fail("ncclCommInitAll", rc); }
  return IST_OK;
}

// device sink: bands + the root's launch + ONE grouped RCCL batch into the root's canvas.  Caller holds g->mu.
int group_launch_locked(ist_group_job* job, const void* const* src, const size_t* src_pitch, void* dst, size_t dst_pitch) {
  ist_group* g = job->g;
  int rc = ensure_arenas(g, job);
  if (rc) return rc;
  for (const auto& u : job->units)
    if (!u.local) { rc = ensure_rccl(g); if (rc) return rc; break; }      // before anything is queued
  bool any_remote = false;
  // 1. every unit on its owner's stream
  for (size_t k = 0; k < job->units.size(); ++k) {
    auto& u = job->units[k];
    if (u.local) rc = launch_band(job, k, src, src_pitch, dst, dst_pitch, false);
    else {
      rc = launch_band(job, k, src, src_pitch, static_cast<uint8_t*>(g->band_arena[static_cast<size_t>(u.rank)]) + u.band_off,
                       static_cast<size_t>(u.X1 - u.X0) * 4, true);
      any_remote = true;
    }
    if (rc) return rc;
  }
  // 2. the root's own launch: slot 0's pointers, by image
  rc = launch_root(job, src, src_pitch, dst, dst_pitch);
  if (rc) return rc;
  if (!any_remote) return IST_OK;
  // 3. ONE grouped batch: every sender's bands to the root, the root's receives on their own stream
  Rccl* R = rccl();
  int nrc = R->GroupStart();
  if (nrc) return nccl_fail("ncclGroupStart", nrc);
  for (auto& u : job->units) {
    if (u.local) continue;
    const size_t bytes = unit_bytes(u);
    void* into = u.in_place ? static_cast<void*>(static_cast<uint8_t*>(dst) + static_cast<size_t>(u.Y0) * dst_pitch)
                            : static_cast<void*>(static_cast<uint8_t*>(g->staging) + u.staging_off);
    const void* band = static_cast<const uint8_t*>(g->band_arena[static_cast<size_t>(u.rank)]) + u.band_off;
    nrc = R->Send(band, bytes, kNcclUint8, 0, g->comm[static_cast<size_t>(u.rank)], g->ctx[static_cast<size_t>(u.rank)]->stream);
    if (!nrc) nrc = R->Recv(into, bytes, kNcclUint8, u.rank, g->comm[0], g->recv_stream);
    if (nrc) { (void)R->GroupEnd(); return nccl_fail("ncclSend/ncclRecv", nrc); }
  }
  nrc = R->GroupEnd();
  if (nrc) return nccl_fail("ncclGroupEnd", nrc);
  // 4. staged bands: placed behind their receive (same stream)
  for (auto& u : job->units) {
    if (!u.place_job) continue;
    const void* band = static_cast<const uint8_t*>(g->staging) + u.staging_off;
    const size_t bp = static_cast<size_t>(u.X1 - u.X0) * 4;
    rc = ist_job_launch(u.place_job, &band, &bp, 1, dst, dst_pitch, g->recv_stream);
    if (rc) return rc;
  }
  return IST_OK;
}

std::string job_key(int64_t cw, int64_t ch, const ist_op* ops, int n_ops, const ist_image_desc* images, int n_images, int filter, int split) {
  std::string k;
  auto put = [&](const void* p, size_t n) { k.append(static_cast<const char*>(p), n); };
  put(&cw, sizeof cw); put(&ch, sizeof ch); put(&filter, sizeof filter); put(&split, sizeof split); put(&n_ops, sizeof n_ops); put(&n_images, sizeof n_images);
  for (int i = 0; i < n_ops; ++i) {            // field by field: struct padding is not part of the key
    const ist_op& o = ops[i];
    put(&o.kind, sizeof o.kind); put(&o.image, sizeof o.image); put(o.m, sizeof o.m); put(o.s, sizeof o.s); put(o.d, sizeof o.d); put(o.rgba, sizeof o.rgba);
  }
  for (int i = 0; i < n_images; ++i) {
    const ist_image_desc& d = images[i];
    const int32_t v[6] = {d.width, d.height, d.orientation, d.bmp_width, d.bmp_height, d.opaque};
    put(v, sizeof v);
  }
  return k;
}

}  // namespace

extern "C" {

// src[k] / src_pitch[k] belong to PART k: the device address (on the part's device) of ROW 0 of the part's image.  A slot
// that holds only rows [sy0, sy1) passes the address of row sy0 minus sy0 * pitch.  dst: the canvas on the root's device,
// rows contiguous.  Asynchronous; ist_group_sync waits for the canvas.
int ist_group_job_launch(ist_group_job* job, const void* const* src, const size_t* src_pitch, int n_parts, void* dst, size_t dst_pitch) {
  if (!job || !src || !dst) return fail(IST_E_INVALID, "ist_group_job_launch: NULL argument");
  if (n_parts != static_cast<int>(job->parts.size())) return fail(IST_E_INVALID, "ist_group_job_launch: one source pointer per part is expected");
  if (dst_pitch != static_cast<size_t>(job->cw) * 4) return fail(IST_E_INVALID, "ist_group_job_launch: the canvas rows must be contiguous (dst_pitch == canvas_w * 4)");
  std::lock_guard<std::mutex> lock(job->g->mu);
  return group_launch_locked(job, src, src_pitch, dst, dst_pitch);
}

int ist_group_sync(ist_group* g) {
  if (!g) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  return group_sync_locked(g);
}

// ---- host path: stitch(images, direction, {devices}) ------------------------------------------------------------------
// plan -> parts -> every device uploads ONLY the source rows its parts sample (its own PCIe link, its own packing threads)
// -> HOST SINK: every band is DMA'ed by its device straight into the pooled pinned result (no gather, no root readback);
//    otherwise: bands + gather -> the root's canvas comes back as one pooled pinned block.
int ist_group_stitch_rgba8(ist_group* g, const ist_image_desc* images, const uint8_t* const* src, const size_t* src_pitch, int n_images,
                           int direction, int mode, double gap, const ist_limits* limits, int filter, int split, ist_plan* out_plan,
                           uint8_t** out_pixels) {
  if (!g) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  if (!out_plan || !out_pixels) return fail(IST_E_INVALID, "ist_group_stitch_rgba8: NULL output");
  *out_pixels = nullptr;
  ist_limits lim;
  if (limits) lim = *limits; else ist_limits_unlimited(&lim);
  int rc = ist_plan_compute(images, n_images, direction, mode, gap, &lim, out_plan);
  if (rc != IST_OK) return rc;
  struct PlanGuard { ist_plan* p; bool keep = false; ~PlanGuard() { if (!keep) ist_plan_free(p); } } pg{out_plan};
  std::vector<ist_op> ops(static_cast<size_t>(out_plan->n_rects) + 1);
  int n_ops = 0;
  rc = ist_plan_ops(out_plan, images, n_images, ops.data(), &n_ops);
  if (rc != IST_OK) return rc;
  static const uint8_t transparent[4] = {0, 0, 0, 0};
  std::lock_guard<std::mutex> glock(g->mu);            // one host-path stitch in flight per group (index.js:772 isStitching)
  // the compiled group job: from the LRU, or compiled now and kept
  ist_group_job* job = nullptr;
  {
    const std::string key = job_key(out_plan->canvas_w, out_plan->canvas_h, ops.data(), n_ops, images, n_images, filter, split);
    for (size_t k = 0; k < g->cache.size(); ++k)
      if (g->cache[k].key == key) {
        ist_group::Cached c = g->cache[k];
        g->cache.erase(g->cache.begin() + static_cast<std::ptrdiff_t>(k));
        g->cache.push_back(c);
        job = c.job;
        break;
      }
    if (!job) {
      job = ist_group_job_create(g, out_plan->canvas_w, out_plan->canvas_h, transparent, ops.data(), n_ops, images, n_images, filter, split);
      if (!job) return g_last_code ? g_last_code : IST_E_INVALID;
      if (g->cache.size() >= ist_group::kCacheJobs) {    // every call ends with the group idle: the oldest job is not in flight
        group_job_free(g->cache.front().job);
        g->cache.erase(g->cache.begin());
      }
      g->cache.push_back(ist_group::Cached{key, job});
    }
  }
  rc = ensure_arenas(g, job);
  if (rc) return rc;

  // holdings: per (device, image) the union of the rows its parts sample, + 16 readable bytes behind the last row
  const size_t nd = g->devs.size();
  struct Hold { int lo = 0, hi = 0; size_t off = 0; };
  std::vector<std::map<int, Hold>> hold(nd);
  for (const auto& rt : job->parts) {
    if (!src || !src[rt.part.image]) return fail(IST_E_DECODE, "图片" + std::to_string(rt.part.image) + "解码异常");
    auto it = hold[static_cast<size_t>(rt.rank)].find(rt.part.image);
    if (it == hold[static_cast<size_t>(rt.rank)].end()) { Hold h; h.lo = rt.part.sy0; h.hi = rt.part.sy1; hold[static_cast<size_t>(rt.rank)][rt.part.image] = h; }
    else { it->second.lo = std::min(it->second.lo, rt.part.sy0); it->second.hi = std::max(it->second.hi, rt.part.sy1); }
  }
  auto width_of = [&](int i) { return static_cast<size_t>(images[i].bmp_width > 0 ? images[i].bmp_width : images[i].width); };
  const size_t canvas_pitch = static_cast<size_t>(out_plan->canvas_w) * 4;
  const size_t canvas_bytes = canvas_pitch * static_cast<size_t>(out_plan->canvas_h);
  const bool host_sink = job->host_sink_ok && !g->self_send;
  ist_ctx* root = g->ctx[0];
  {
    DeviceGuard dg(root->device);
    rc = grow_device(&root->scratch_dst, &root->scratch_dst_bytes, canvas_bytes);
    if (rc) return rc;
  }
  uint8_t* host = static_cast<uint8_t*>(pool_take(canvas_bytes));
  if (!host) return fail(IST_E_NOMEM, "out of pinned host memory for the result");
  struct HostGuard { uint8_t* p; ist_group* g; bool keep = false; ~HostGuard() { if (!keep) { (void)group_sync_locked(g); pool_give(p); } } } hg{host, g};
  // part pointers: row 0 of the image as seen from the holding (offsets are known before the uploads run)
  for (size_t r = 0; r < nd; ++r) {
    size_t total = 0;
    for (auto& kv : hold[r]) { kv.second.off = total; total += round256(width_of(kv.first) * 4 * static_cast<size_t>(kv.second.hi - kv.second.lo) + 16); }
    DeviceGuard dg(g->devs[r]);
    rc = grow_device(&g->ctx[r]->scratch_src, &g->ctx[r]->scratch_src_bytes, total ? total : 256);
    if (rc) return rc;
  }
  std::vector<const void*> psrc(job->parts.size(), nullptr);
  std::vector<size_t> ppitch(job->parts.size(), 0);
  for (size_t k = 0; k < job->parts.size(); ++k) {
    const auto& rt = job->parts[k];
    const Hold& h = hold[static_cast<size_t>(rt.rank)][rt.part.image];
    const size_t row = width_of(rt.part.image) * 4;
    psrc[k] = reinterpret_cast<const void*>(reinterpret_cast<uintptr_t>(g->ctx[static_cast<size_t>(rt.rank)]->scratch_src) + h.off - static_cast<uintptr_t>(h.lo) * row);
    ppitch[k] = row;
  }
  // one host thread per device: upload its rows, then (host sink) render its bands and send each home over its own link
  std::vector<int> up_rc(nd, IST_OK);
  std::vector<std::string> up_err(nd);
  std::vector<std::unique_lock<std::mutex>> locks;
  for (size_t r = 0; r < nd; ++r) locks.emplace_back(g->ctx[r]->mu);
  {
    std::vector<std::thread> th;
    for (size_t r = 0; r < nd; ++r) th.emplace_back([&, r]() {
      ist_ctx* c = g->ctx[r];
      DeviceGuard dg(c->device);
      int rc2 = IST_OK;
      std::vector<RowsCopy> up;
      for (auto& kv : hold[r]) {
        const size_t row = width_of(kv.first) * 4, hp = src_pitch ? src_pitch[kv.first] : row;
        if (hp < row) { rc2 = fail(IST_E_INVALID, "src_pitch too small"); break; }
        up.push_back(RowsCopy{static_cast<uint8_t*>(c->scratch_src) + kv.second.off, src[kv.first] + static_cast<size_t>(kv.second.lo) * hp, nullptr, hp, row,
                              static_cast<size_t>(kv.second.hi - kv.second.lo)});
      }
      if (rc2 == IST_OK) { if (!c->stager) c->stager.reset(new Stager(c->device)); rc2 = c->stager->upload(up, c->stream); }
      if (rc2 == IST_OK && host_sink) {
        for (size_t k = 0; k < job->units.size() && rc2 == IST_OK; ++k) {
          const auto& u = job->units[k];
          if (static_cast<size_t>(u.rank) != r) continue;
          uint8_t* band = static_cast<uint8_t*>(g->band_arena[r]) + u.band_off;
          rc2 = launch_band(job, k, psrc.data(), ppitch.data(), band, canvas_pitch, true);
          if (rc2 == IST_OK && hipMemcpyAsync(host + static_cast<size_t>(u.Y0) * canvas_pitch, band, unit_bytes(u), hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
            (void)hipGetLastError();
            rc2 = fail(IST_E_HIP, "band readback failed");
          }
        }
        if (rc2 == IST_OK && r == 0) {                 // the root: everything that is not a band
          rc2 = launch_root(job, psrc.data(), ppitch.data(), root->scratch_dst, canvas_pitch);
          for (size_t q = 0; q < job->root_rows.size() && rc2 == IST_OK; ++q) {
            const size_t y0 = static_cast<size_t>(job->root_rows[q].first), y1 = static_cast<size_t>(job->root_rows[q].second);
            if (hipMemcpyAsync(host + y0 * canvas_pitch, static_cast<uint8_t*>(root->scratch_dst) + y0 * canvas_pitch, (y1 - y0) * canvas_pitch, hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
              (void)hipGetLastError();
              rc2 = fail(IST_E_HIP, "result readback failed");
            }
          }
        }
      }
      up_rc[r] = rc2;
      if (rc2) up_err[r] = g_last_error;
    });
    for (auto& t : th) t.join();
  }
  for (size_t r = 0; r < nd; ++r) if (up_rc[r]) return fail(up_rc[r], up_err[r]);
  if (!host_sink) {
    rc = group_launch_locked(job, psrc.data(), ppitch.data(), root->scratch_dst, canvas_pitch);
    if (rc) return rc;
    rc = group_sync_locked(g);
    if (rc) return rc;
    DeviceGuard dg(root->device);
    if (hipMemcpyAsync(host, root->scratch_dst, canvas_bytes, hipMemcpyDeviceToHost, root->stream) != hipSuccess) { (void)hipGetLastError(); return fail(IST_E_HIP, "result readback failed"); }
  }
  rc = group_sync_locked(g);
  if (rc) return rc;
  if (host_sink) g_host_sink_stitches.fetch_add(1, std::memory_order_relaxed);
  hg.keep = true;
  *out_pixels = host;
  pg.keep = true;
  return IST_OK;
}

// stitch(images, direction, {devices: [...]}) in one call: groups are cached per device list (the most recent kGroups lists)
int ist_stitch_rgba8_multi(const int* devices, int ndev, const ist_image_desc* images, const uint8_t* const* src, const size_t* src_pitch,
                           int n_images, int direction, int mode, double gap, const ist_limits* limits, int filter, int split,
                           ist_plan* out_plan, uint8_t** out_pixels) {
  if (!devices || ndev < 1) return fail(IST_E_INVALID, "ist_stitch_rgba8_multi: empty device list");
  constexpr size_t kGroups = 4;
  struct Entry { std::vector<int> key; std::shared_ptr<ist_group> g; };
  static std::mutex mu;
  static std::vector<Entry>* cache = new std::vector<Entry>();   // never destroyed: no HIP calls at exit
  std::shared_ptr<ist_group> g;
  {
    std::lock_guard<std::mutex> lock(mu);
    const std::vector<int> key(devices, devices + ndev);
    for (size_t k = 0; k < cache->size(); ++k)
      if ((*cache)[k].key == key) {
        Entry e = (*cache)[k];
        cache->erase(cache->begin() + static_cast<std::ptrdiff_t>(k));
        cache->push_back(e);
        g = e.g;
        break;
      }
    if (!g) {
      ist_group* raw = ist_group_create(devices, ndev);
      if (!raw) return g_last_code ? g_last_code : IST_E_INVALID;
      g.reset(raw, ist_group_destroy);            // an evicted group is destroyed when its last call has returned
      if (cache->size() >= kGroups) cache->erase(cache->begin());
      cache->push_back(Entry{key, g});
    }
  }
  return ist_group_stitch_rgba8(g.get(), images, src, src_pitch, n_images, direction, mode, gap, limits, filter, split, out_plan, out_pixels);
}

}  // extern "C"
