// ist_host.cpp — pinned result pool + staged host<->device copies (see ist_host.h for the rules of this layer).
#include "ist_host.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

#include "ist_internal.h"

namespace ist {

namespace {

// ------------------------------------------------------------------------------------------------ result pool
constexpr size_t kPoolGranule = 2u << 20;        // block sizes are multiples of 2 MiB: similar jobs reuse each other's blocks
constexpr size_t kPoolKeepBytes = 2ull << 30;    // at most this much idle pinned memory is kept ...
constexpr size_t kPoolKeepBlocks = 6;            // ... in at most this many idle blocks

struct Block { void* p; size_t cap; bool busy; uint64_t stamp; };
std::mutex g_pool_mu;
std::vector<Block>& pool() { static std::vector<Block>* v = new std::vector<Block>(); return *v; }   // never destroyed: no HIP calls at exit
uint64_t g_pool_clock = 0;

void pool_enforce_locked() {
  for (;;) {
    size_t idle_bytes = 0, idle = 0, oldest = SIZE_MAX;
    std::vector<Block>& v = pool();
    for (size_t i = 0; i < v.size(); ++i) {
      if (v[i].busy) continue;
      idle_bytes += v[i].cap; ++idle;
      if (oldest == SIZE_MAX || v[i].stamp < v[oldest].stamp) oldest = i;
    }
    if (oldest == SIZE_MAX || (idle_bytes <= kPoolKeepBytes && idle <= kPoolKeepBlocks)) return;
    (void)hipHostFree(v[oldest].p);
    v.erase(v.begin() + static_cast<std::ptrdiff_t>(oldest));
  }
}

}  // namespace

void* pool_take(size_t bytes) {
  const size_t need = ((bytes ? bytes : 1) + kPoolGranule - 1) / kPoolGranule * kPoolGranule;
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    std::vector<Block>& v = pool();
    size_t best = SIZE_MAX;
    for (size_t i = 0; i < v.size(); ++i)
      if (!v[i].busy && v[i].cap >= need && v[i].cap <= 2 * need && (best == SIZE_MAX || v[i].cap < v[best].cap)) best = i;
    if (best != SIZE_MAX) { v[best].busy = true; v[best].stamp = ++g_pool_clock; return v[best].p; }
  }
  void* p = nullptr;
  if (hipHostMalloc(&p, need, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    pool_trim();                                  // idle blocks may be what is in the way
    if (hipHostMalloc(&p, need, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  }
  std::lock_guard<std::mutex> lock(g_pool_mu);
  pool().push_back(Block{p, need, true, ++g_pool_clock});
  return p;
}

bool pool_give(void* p) {
  std::lock_guard<std::mutex> lock(g_pool_mu);
  for (Block& b : pool())
    if (b.p == p) {
      b.busy = false; b.stamp = ++g_pool_clock;
      pool_enforce_locked();
      return true;
    }
  return false;
}

void pool_trim() {
  std::lock_guard<std::mutex> lock(g_pool_mu);
  std::vector<Block>& v = pool();
  for (size_t i = v.size(); i-- > 0;)
    if (!v[i].busy) { (void)hipHostFree(v[i].p); v.erase(v.begin() + static_cast<std::ptrdiff_t>(i)); }
}

// ------------------------------------------------------------------------------------------------ staged copies
namespace {

constexpr size_t kChunk = 4u << 20;      // one pinned chunk: 4 MiB (~75 us of PCIe, ~0.4 ms of one core's memcpy)
constexpr int kLanes = 4;                // packing threads (each with its own stream and two chunks); 8 lanes made the banded path 1.5-4x SLOWER (measured)

// a piece is what one chunk carries: n_rows whole rows, or one segment of a row that is longer than a chunk
struct Piece { uint32_t item; size_t row0, n_rows, col0, n_cols; };

void split(const std::vector<RowsCopy>& items, std::vector<Piece>* out) {
  for (size_t k = 0; k < items.size(); ++k) {
    const RowsCopy& it = items[k];
    if (it.row == 0 || it.rows == 0) continue;
    if (it.row <= kChunk) {
      const size_t per = kChunk / it.row;
      for (size_t r = 0; r < it.rows; r += per) out->push_back(Piece{static_cast<uint32_t>(k), r, std::min(per, it.rows - r), 0, it.row});
    } else {
      for (size_t r = 0; r < it.rows; ++r)
        for (size_t c = 0; c < it.row; c += kChunk) out->push_back(Piece{static_cast<uint32_t>(k), r, 1, c, std::min(kChunk, it.row - c)});
    }
  }
}

inline size_t piece_bytes(const Piece& p) { return p.n_rows * p.n_cols; }

void pack(const RowsCopy& it, const Piece& p, uint8_t* chunk) {
  const uint8_t* s = static_cast<const uint8_t*>(it.host_src) + p.row0 * it.host_pitch + p.col0;
  if (it.host_pitch == p.n_cols) { std::memcpy(chunk, s, p.n_rows * p.n_cols); return; }
  for (size_t r = 0; r < p.n_rows; ++r) std::memcpy(chunk + r * p.n_cols, s + r * it.host_pitch, p.n_cols);
}
void unpack(const RowsCopy& it, const Piece& p, const uint8_t* chunk) {
  uint8_t* d = static_cast<uint8_t*>(it.host_dst) + p.row0 * it.host_pitch + p.col0;
  if (it.host_pitch == p.n_cols) { std::memcpy(d, chunk, p.n_rows * p.n_cols); return; }
  for (size_t r = 0; r < p.n_rows; ++r) std::memcpy(d + r * it.host_pitch, chunk + r * p.n_cols, p.n_cols);
}
inline uint8_t* dev_of(const RowsCopy& it, const Piece& p) { return static_cast<uint8_t*>(it.dev) + p.row0 * it.row + p.col0; }

}  // namespace

void Stager::release_lane(Lane& l) {
  if (l.stream) (void)hipStreamSynchronize(l.stream);
  for (int s = 0; s < 2; ++s) { if (l.done[s]) (void)hipEventDestroy(l.done[s]); if (l.chunk[s]) (void)hipHostFree(l.chunk[s]); l.done[s] = nullptr; l.chunk[s] = nullptr; }
  if (l.tail) (void)hipEventDestroy(l.tail);
  if (l.stream) (void)hipStreamDestroy(l.stream);
  l.tail = nullptr; l.stream = nullptr;
}

Stager::~Stager() {
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(device_) != hipSuccess) return;
  if (big_.stream) (void)hipStreamSynchronize(big_.stream);
  for (int s = 0; s < 2; ++s) { if (big_.done[s]) (void)hipEventDestroy(big_.done[s]); if (big_.piece[s]) (void)hipHostFree(big_.piece[s]); }
  if (big_.tail) (void)hipEventDestroy(big_.tail);
  if (big_.stream) (void)hipStreamDestroy(big_.stream);
  for (Lane& l : lanes_) release_lane(l);
  if (gate_) (void)hipEventDestroy(gate_);
  if (prev >= 0) (void)hipSetDevice(prev);
}

int Stager::ensure() {
  if (!lanes_.empty()) return IST_OK;
  std::vector<Lane> lanes(kLanes);
  hipEvent_t gate = nullptr;
  bool ok = hipEventCreateWithFlags(&gate, hipEventDisableTiming) == hipSuccess;
  for (Lane& l : lanes) {
    ok = ok && hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&l.tail, hipEventDisableTiming) == hipSuccess;
    for (int s = 0; s < 2 && ok; ++s)
      ok = hipHostMalloc(&l.chunk[s], kChunk, hipHostMallocDefault) == hipSuccess &&
           hipEventCreateWithFlags(&l.done[s], hipEventDisableTiming) == hipSuccess &&
           hipEventRecord(l.done[s], l.stream) == hipSuccess;           // recorded once: synchronising it is always legal
  }
  if (!ok) {                    // nothing half-built is published: the next call starts over
    (void)hipGetLastError();
    for (Lane& l : lanes) release_lane(l);
    if (gate) (void)hipEventDestroy(gate);
    return fail(IST_E_HIP, "allocating the pinned staging ring failed");
  }
  lanes_.swap(lanes);
  gate_ = gate;
  return IST_OK;
}

namespace { constexpr size_t kBigPiece = 32u << 20; constexpr int kBigPackers = 4; }

int Stager::ensure_big() {
  if (big_.stream) return IST_OK;
  Big b;
  bool ok = hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&b.tail, hipEventDisableTiming) == hipSuccess;
  for (int s = 0; s < 2 && ok; ++s)
    ok = hipHostMalloc(&b.piece[s], kBigPiece, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&b.done[s], hipEventDisableTiming) == hipSuccess &&
         hipEventRecord(b.done[s], b.stream) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    for (int s = 0; s < 2; ++s) { if (b.done[s]) (void)hipEventDestroy(b.done[s]); if (b.piece[s]) (void)hipHostFree(b.piece[s]); }
    if (b.tail) (void)hipEventDestroy(b.tail);
    if (b.stream) (void)hipStreamDestroy(b.stream);
    return fail(IST_E_HIP, "allocating the large staging pieces failed");
  }
  big_ = b;
  return IST_OK;
}

int Stager::upload_big(const std::vector<RowsCopy>& items, hipStream_t after, WorkerPool* pool) {
  if (!pool) return upload(items, after);
  std::vector<RowsCopy> small;                       // rows that do not fit a piece: the chunked path
  bool any = false;
  for (const RowsCopy& it : items) { if (it.row == 0 || it.rows == 0) continue; if (it.row > kBigPiece) small.push_back(it); else any = true; }
  if (!small.empty()) { const int rc = upload(small, after); if (rc) return rc; }
  if (!any) return IST_OK;
  int rc = ensure_big();
  if (rc) return rc;
  for (const RowsCopy& it : items) {
    if (it.row == 0 || it.rows == 0 || it.row > kBigPiece) continue;
    // caller memory that IS page-locked (a host that keeps its images in pinned blocks; nothing is registered here) and has dense rows
    // needs no staging: one copy straight from it
    if (it.host_pitch == it.row) {
      hipPointerAttribute_t attr;
      std::memset(&attr, 0, sizeof(attr));
      if (hipPointerGetAttributes(&attr, it.host_src) == hipSuccess && attr.type == hipMemoryTypeHost) {
        if (hipMemcpyAsync(it.dev, it.host_src, it.rows * it.row, hipMemcpyHostToDevice, big_.stream) != hipSuccess) { (void)hipGetLastError(); return fail(IST_E_HIP, "host-to-device copy failed"); }
        continue;
      }
      (void)hipGetLastError();                  // (ordinary memory: the query fails, which is the answer)
    }
    const size_t per = std::max<size_t>(1, kBigPiece / it.row);
    for (size_t r0 = 0; r0 < it.rows; r0 += per) {
      const size_t nr = std::min(per, it.rows - r0);
      const int s = static_cast<int>(big_.k++ & 1u);
      if (hipEventSynchronize(big_.done[s]) != hipSuccess) { (void)hipGetLastError(); return fail(IST_E_HIP, "host-to-device staging failed"); }   // the piece's previous copy has left
      uint8_t* piece = static_cast<uint8_t*>(big_.piece[s]);
      const int workers = static_cast<int>(std::min<size_t>(kBigPackers, std::max<size_t>(1, nr * it.row >> 20)));
      const size_t share = (nr + static_cast<size_t>(workers) - 1) / static_cast<size_t>(workers);
      auto part = [&](int w) {
        const size_t a = static_cast<size_t>(w) * share, b = std::min(nr, a + share);
        if (a >= b) return;
        const uint8_t* src = static_cast<const uint8_t*>(it.host_src) + (r0 + a) * it.host_pitch;
        if (it.host_pitch == it.row) std::memcpy(piece + a * it.row, src, (b - a) * it.row);
        else for (size_t r = a; r < b; ++r) std::memcpy(piece + r * it.row, src + (r - a) * it.host_pitch, it.row);
      };
      if (workers > 1) { pool->run(workers, part); pool->wait(); } else part(0);
      if (hipMemcpyAsync(static_cast<uint8_t*>(it.dev) + r0 * it.row, piece, nr * it.row, hipMemcpyHostToDevice, big_.stream) != hipSuccess ||
          hipEventRecord(big_.done[s], big_.stream) != hipSuccess) { (void)hipGetLastError(); return fail(IST_E_HIP, "host-to-device staging failed"); }
    }
  }
  if (hipEventRecord(big_.tail, big_.stream) != hipSuccess || hipStreamWaitEvent(after, big_.tail, 0) != hipSuccess) { (void)hipGetLastError(); return fail(IST_E_HIP, "ordering the uploads before the launch failed"); }
  return IST_OK;
}

int Stager::sync() {
  bool ok = true;
  if (big_.stream) ok = hipStreamSynchronize(big_.stream) == hipSuccess;
  for (Lane& l : lanes_) if (l.stream) ok = (hipStreamSynchronize(l.stream) == hipSuccess) && ok;
  return ok ? IST_OK : fail(IST_E_HIP, "hipStreamSynchronize failed");
}

int Stager::upload(const std::vector<RowsCopy>& items, hipStream_t after) { return run(items, true, after); }
int Stager::download(const std::vector<RowsCopy>& items, hipStream_t before) { return run(items, false, before); }

int Stager::run(const std::vector<RowsCopy>& items, bool up, hipStream_t other) {
  std::vector<Piece> pieces;
  split(items, &pieces);
  if (pieces.empty()) return IST_OK;
  int rc = ensure();
  if (rc) return rc;
  size_t total = 0;
  for (const Piece& p : pieces) total += piece_bytes(p);
  const int n_lanes = total < (2u << 20) ? 1 : static_cast<int>(std::min<size_t>(kLanes, pieces.size()));
  if (!up) {                                   // the lanes read what `other` produces
    if (hipEventRecord(gate_, other) != hipSuccess) return fail(IST_E_HIP, "hipEventRecord failed");
    for (int l = 0; l < n_lanes; ++l)
      if (hipStreamWaitEvent(lanes_[static_cast<size_t>(l)].stream, gate_, 0) != hipSuccess) return fail(IST_E_HIP, "hipStreamWaitEvent failed");
  }
  std::atomic<size_t> next{0};
  std::atomic<int> err{0};
  auto work = [&](int li) {
    if (hipSetDevice(device_) != hipSuccess) { err = 1; return; }
    Lane& L = lanes_[static_cast<size_t>(li)];
    const Piece* pending[2] = {nullptr, nullptr};
    unsigned k = 0;
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= pieces.size() || err.load()) break;
      const Piece& p = pieces[i];
      const RowsCopy& it = items[p.item];
      const int s = static_cast<int>(k++ & 1u);
      if (hipEventSynchronize(L.done[s]) != hipSuccess) { err = 1; break; }        // the chunk's previous DMA has landed
      uint8_t* chunk = static_cast<uint8_t*>(L.chunk[s]);
      if (up) {
        pack(it, p, chunk);
        if (hipMemcpyAsync(dev_of(it, p), chunk, piece_bytes(p), hipMemcpyHostToDevice, L.stream) != hipSuccess) { err = 1; break; }
      } else {
        if (pending[s]) { unpack(items[pending[s]->item], *pending[s], chunk); pending[s] = nullptr; }
        if (hipMemcpyAsync(chunk, dev_of(it, p), piece_bytes(p), hipMemcpyDeviceToHost, L.stream) != hipSuccess) { err = 1; break; }
        pending[s] = &p;
      }
      if (hipEventRecord(L.done[s], L.stream) != hipSuccess) { err = 1; break; }
    }
    if (!up)                                    // drain, oldest first
      for (unsigned d = 0; d < 2; ++d) {
        const int s = static_cast<int>((k + d) & 1u);
        if (!pending[s]) continue;
        if (hipEventSynchronize(L.done[s]) != hipSuccess) { err = 1; continue; }
        unpack(items[pending[s]->item], *pending[s], static_cast<const uint8_t*>(L.chunk[s]));
      }
  };
  if (n_lanes == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int l = 1; l < n_lanes; ++l) th.emplace_back(work, l);
    work(0);
    for (std::thread& t : th) t.join();
  }
  if (err.load()) { (void)hipGetLastError(); return fail(IST_E_HIP, up ? "host-to-device staging failed" : "device-to-host staging failed"); }
  if (up)                                       // whoever uses `other` next sees the uploads
    for (int l = 0; l < n_lanes; ++l) {
      Lane& L = lanes_[static_cast<size_t>(l)];
      if (hipEventRecord(L.tail, L.stream) != hipSuccess || hipStreamWaitEvent(other, L.tail, 0) != hipSuccess)
        return fail(IST_E_HIP, "ordering the uploads before the launch failed");
    }
  return IST_OK;
}


// ------------------------------------------------------------------------------------------------ parked host threads
WorkerPool::~WorkerPool() {
  { std::lock_guard<std::mutex> lock(mu_); stop_ = true; }
  work_cv_.notify_all();
  for (std::thread& t : th_) if (t.joinable()) t.join();
}

void WorkerPool::loop() {
  std::unique_lock<std::mutex> lock(mu_);
  for (;;) {
    work_cv_.wait(lock, [&]() { return stop_ || next_ < total_; });
    if (stop_) return;
    const int i = next_++;
    lock.unlock();
    fn_(i);
    lock.lock();
    if (++done_ == total_) done_cv_.notify_all();
  }
}

void WorkerPool::run(int n, std::function<void(int)> fn) {
  if (n <= 0) return;
  {
    std::lock_guard<std::mutex> lock(mu_);
    tl_mark("pool: queue locked");
    fn_ = std::move(fn);
    next_ = 0; done_ = 0; total_ = n;
    const size_t want = static_cast<size_t>(n < kMaxThreads ? n : kMaxThreads);
    while (th_.size() < want) th_.emplace_back([this]() { loop(); });
  }
  tl_mark("pool: tasks posted");
  work_cv_.notify_all();
  tl_mark("pool: workers notified");
}

void WorkerPool::wait() {
  std::unique_lock<std::mutex> lock(mu_);
  done_cv_.wait(lock, [&]() { return done_ == total_; });
  total_ = 0; next_ = 0; done_ = 0;
}

}  // namespace ist
