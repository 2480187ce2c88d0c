// ist_ctx.h — the two opaque handles of the C-ABI, shared by the runtime (ist_runtime.cpp) and the device-group layer
// (ist_mgpu.cpp).  Reference anchors: a context stands for the canvas node obtained at pages/index/index.js:1196-1204; a
// job for the offscreen canvas + the draws recorded on it (utils/canvas.js:131-150, index.js:1391-1428, 1532-1551).
#ifndef IST_CTX_H_
#define IST_CTX_H_

#include <hip/hip_runtime_api.h>

#include <memory>
#include <mutex>
#include <vector>

#include "ist_jpeg.h"

#include "ist_host.h"
#include "ist_internal.h"

struct ist_ctx {
  int device = 0;
  hipStream_t stream = nullptr;          // host-buffer entry points run here; the device path takes the caller's stream
  void* scratch_src = nullptr; size_t scratch_src_bytes = 0;
  void* scratch_dst = nullptr; size_t scratch_dst_bytes = 0;
  void* scratch_dec = nullptr; size_t scratch_dec_bytes = 0;   // JPEG coefficient / sample planes of ist_decode_files_device
  void* scratch_huff = nullptr; size_t scratch_huff_bytes = 0; // GPU Huffman decoder: scans, tables, per-subsequence state
  void* scratch_png = nullptr; size_t scratch_png_bytes = 0;   // compressing PNG encoder: one slot per 16 KiB chunk + its tables
  void* scratch_file = nullptr; size_t scratch_file_bytes = 0; // device image of a PNG file on its way to the host
  void* scratch_arena = nullptr; size_t scratch_arena_bytes = 0; // file pipeline: bitmaps + JPEG planes + canvas + PNG of one call
  // file pipeline (ist_stitch_files_png / ist_decode_files_device): one stream + event + Huffman scratch per image, so that
  // the images' decode chains (upload -> Huffman passes -> reconstruction) overlap each other and the export of the bands
  // that are already final; grow-only, made on first use
  std::vector<hipStream_t> img_stream;
  std::vector<hipEvent_t> img_event;
  std::vector<void*> img_huff; std::vector<size_t> img_huff_bytes;
  std::unique_ptr<ist::WorkerPool> workers;   // parked host threads for the per-file work of a call (made on first use)
  std::vector<ist::ScanBuf> scan_bufs;        // de-stuffed scans of the last call: their memory is reused (a fresh 1.8 MB block per image and call is 450 page faults on its parse thread)
  std::vector<ist::ScanBuf> file_bufs;        // ist_stitch_paths_png: the files' bytes, read (not mapped) into blocks kept from call to call
  void* scratch_ent = nullptr; size_t scratch_ent_bytes = 0;   // sparse coefficient entries of host-decoded JPEGs (progressive, restart intervals)
  hipStream_t render = nullptr;          // file pipeline: Huffman batch + per-image reconstruction + band launches, beside the PNG encoder on `stream`
  hipEvent_t render_done = nullptr;
  hipStream_t png2 = nullptr;            // the compressing PNG encoder alternates its slabs between `stream` and this one
  hipStream_t aux = nullptr;             // second stream of the host-path entry points (PNG slabs travel on it while later ones compress)
  // device blocks of destroyed jobs' tables, re-used by the next job of this context instead of a hipMalloc + hipFree pair
  // per job (a free also synchronises the device); at most kTablePool blocks are kept (the file pipeline compiles one job per image + one)
  static constexpr int kTablePool = 32;
  struct TableBlock { uint8_t* p; size_t bytes; };
  std::vector<TableBlock> table_pool;
  std::mutex table_mu;
  std::mutex mu;                         // one host-path stitch in flight per context (index.js:772 isStitching)
  int png_level = 1;                     // 1: Paeth + run-length + Huffman; 0: stored deflate blocks (ist_ctx_set_png_level)
  std::unique_ptr<ist::Stager> stager;   // pinned staging ring, built on first use
  bool timing_on = false;                // ist_ctx_set_timing: the file pipeline records its phase times (adds a sync per phase)
  double last_ms[IST_PHASE_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace ist {
// where a compiled job's five tables sit in the job's device block
struct DevTables {
  DevOp* ops = nullptr;
  DevCell* cells = nullptr;
  DevBand* bands = nullptr;
  int32_t* stacks = nullptr;
  DevTile* tiles = nullptr;
};
}  // namespace ist

struct ist_job {
  ist_ctx* ctx = nullptr;
  ist::Compiled host;
  std::unique_ptr<ist::FlatTwin> flat;   // or none: the job is not made of whole dense rows (ist_internal.h)
  ist::DevTables flat_dt;
  uint8_t* d_tables = nullptr;           // ONE device allocation holding the five tables below
  size_t d_tables_bytes = 0;
  // the streams the job was launched on since it was created (ist_job_destroy waits for THOSE before it hands the tables to
  // the next job - not for the whole device: other streams of a shared device keep running); more than kStreams distinct
  // ones fall back to a device-wide wait
  static constexpr int kStreams = 4;
  hipStream_t launched_on[kStreams] = {nullptr, nullptr, nullptr, nullptr};
  int n_launched_on = 0;
  bool launched = false, launched_many = false;
  std::mutex launch_mu;
  ist::DevTables dt;
  int max_image = -1;
};

namespace ist {

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// grow-only device scratch
int grow_device(void** p, size_t* have, size_t need);


}  // namespace ist

#endif  // IST_CTX_H_
