// ist_host.h — host <-> device transfers of the host-buffer entry points.
//
// Reference anchor: the mini-program hands the platform a decoded bitmap per image (utils/canvas.js:27-121) and gets
// the finished strip back from the export (utils/canvas.js:205-242); on a discrete GPU both cross PCIe.
//
// Rules of this layer (the round-1 abort):
//   * the library NEVER page-locks memory it does not own (no hipHostRegister of caller buffers) and never issues a
//     2-D (pitched) runtime copy: every DMA is a linear copy between device memory and pinned memory the library
//     allocated itself (hipHostMalloc);
//   * caller buffers (pageable, any pitch) are packed into / unpacked from a ring of pinned chunks by a few host threads,
//     each with its own stream, so packing, DMA and the next pack overlap;
//   * buffers the library RETURNS (ist_stitch_rgba8, the *_png calls) come from a pool of pinned blocks, are filled by
//     one DMA with no host copy, and go back to the pool through ist_free.
#ifndef IST_HOST_H_
#define IST_HOST_H_

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <thread>
#include <mutex>
#include <functional>
#include <condition_variable>
#include <cstdint>
#include <vector>

namespace ist {

// ---- pool of pinned result buffers (process-wide) -----------------------------------------------------------------
void* pool_take(size_t bytes);          // pinned, portable (any device may DMA into it); nullptr on failure
bool pool_give(void* p);                // true when p came from pool_take (the block is kept for reuse or released)
void pool_trim();                       // release every cached block

// ---- staged copies between caller memory and device memory -------------------------------------------------------
struct RowsCopy {                       // `rows` rows of `row` bytes; the device side is always contiguous (pitch == row)
  void* dev;                            // device address of the first row
  const void* host_src;                 // upload: caller memory to read  (pitch host_pitch)
  void* host_dst;                       // download: caller memory to write (pitch host_pitch)
  size_t host_pitch, row, rows;
};

class Stager {
 public:
  explicit Stager(int device) : device_(device) {}
  ~Stager();
  Stager(const Stager&) = delete;
  Stager& operator=(const Stager&) = delete;
  // Uploads every item; on return the copies are ENQUEUED and `after` (a stream of the same device) has been made to
  // wait for them, so work submitted to `after` next sees the data.  The caller's buffers are no longer needed.
  int upload(const std::vector<RowsCopy>& items, hipStream_t after);
  // Downloads every item; `before` is a stream whose already-submitted work produces the data.  Returns when the
  // caller's buffers are complete.
  int download(const std::vector<RowsCopy>& items, hipStream_t before);
  int sync();                             // waits for every lane's stream
  // The same as upload(), for an upload that shares PCIe with downloads in flight: ONE stream and two pinned pieces of 32 MiB, each
  // packed by up to four of `pool`'s parked threads and sent as ONE copy.  (Measured, tools/exp/duplex.cpp + duplex2.cpp: 4 MiB copies on
  // four streams against concurrent 48 MB downloads fall to 12.7 GB/s each way; pieces of >= 16 MiB on one stream hold 48 GB/s each
  // way.)  `pool` must be idle; rows longer than a piece go through upload().
  int upload_big(const std::vector<RowsCopy>& items, hipStream_t after, class WorkerPool* pool);

 private:
  struct Lane { hipStream_t stream = nullptr; void* chunk[2] = {nullptr, nullptr}; hipEvent_t done[2] = {nullptr, nullptr}; hipEvent_t tail = nullptr; };
  int ensure();
  static void release_lane(Lane& l);
  int run(const std::vector<RowsCopy>& items, bool up, hipStream_t other);
  int device_;
  std::vector<Lane> lanes_;
  hipEvent_t gate_ = nullptr;
  struct Big { hipStream_t stream = nullptr; void* piece[2] = {nullptr, nullptr}; hipEvent_t done[2] = {nullptr, nullptr}; hipEvent_t tail = nullptr; unsigned k = 0; } big_;
  int ensure_big();
};


// A few parked host threads for the per-file work of a call (container parse, de-stuffing, host entropy stages): starting a
// std::thread per image cost 33 us each on the GPU boxes, in series, before the last image's parse even began (0.3 ms for nine
// files).  run() hands out the indices 0 .. n-1 and returns; wait() returns when all of them are done.  One run at a time
// (the context's mutex); the pool grows to what a run asks for, up to kMaxThreads, and tasks beyond that queue.
class WorkerPool {
 public:
  static constexpr int kMaxThreads = 32;
  WorkerPool() = default;
  ~WorkerPool();
  WorkerPool(const WorkerPool&) = delete;
  WorkerPool& operator=(const WorkerPool&) = delete;
  void run(int n, std::function<void(int)> fn);
  void wait();
 private:
  void loop();
  std::mutex mu_;
  std::condition_variable work_cv_, done_cv_;
  std::vector<std::thread> th_;
  std::function<void(int)> fn_;
  int next_ = 0, total_ = 0, done_ = 0;
  bool stop_ = false;
};

}  // namespace ist

#endif  // IST_HOST_H_
