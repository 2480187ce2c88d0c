// ist_jpeg.h — what the host-side JPEG entropy decoder hands to the GPU stages
#ifndef IST_JPEG_H_
#define IST_JPEG_H_

#include <cstdint>
#include <cstdlib>
#include <utility>
#include <vector>

namespace ist {

// Zero-initialised int16 plane from calloc: the kernel hands out zero pages on first touch, so there is no separate
// 36 MB memset pass per 12 MP image (std::vector::assign writes every page before the decoder does).
class CoefBuf {
 public:
  CoefBuf() = default;
  ~CoefBuf() { std::free(p_); }
  CoefBuf(const CoefBuf&) = delete;
  CoefBuf& operator=(const CoefBuf&) = delete;
  CoefBuf(CoefBuf&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
  CoefBuf& operator=(CoefBuf&& o) noexcept { if (this != &o) { std::free(p_); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; } return *this; }
  bool alloc_zero(size_t n) { std::free(p_); p_ = static_cast<int16_t*>(std::calloc(n ? n : 1, sizeof(int16_t))); n_ = p_ ? n : 0; return p_ != nullptr; }
  bool empty() const { return n_ == 0; }
  size_t size() const { return n_; }
  int16_t* data() { return p_; }
  const int16_t* data() const { return p_; }
  const int16_t* begin() const { return p_; }
  const int16_t* end() const { return p_ + n_; }
 private:
  int16_t* p_ = nullptr; size_t n_ = 0;
};

struct JpegComp {
  int id = 0, h = 1, v = 1, tq = 0;
  int blocks_x = 0, blocks_y = 0;      // padded to whole MCUs
  uint16_t q[64];                      // quantisation table, natural (row-major) order
  CoefBuf coef;                        // DENSE form (progressive files): blocks_y * blocks_x blocks of 64 coefficients,
                                       // natural order, NOT dequantised
  // SPARSE form (sequential files: every block is coded exactly once): the non-zero coefficients only, in decoding
  // order, entry = (natural-order index << 16) | uint16(value); block b (raster order) owns ent[start[b] .. +cnt[b]).
  // A 12 MP photo is ~2.5 M entries (10 MB) instead of a 36 MB plane: less host memory to fault in, 4x less H2D;
  // the GPU scatters the entries into a zeroed dense plane (jpeg_launch_scatter) in front of the IDCT.
  bool sparse = false;
  std::vector<uint32_t> ent, start;
  std::vector<uint8_t> cnt;
};

struct JpegImage {
  int width = 0, height = 0, ncomp = 0;
  int hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
  int orientation = 0;                 // EXIF 1..8, 0 = absent
  int scans = 0;
  JpegComp comp[3];
};

// ---- what the GPU entropy decoder (ist_jpeg_gpu.hip) needs from the container: the de-stuffed scan and its tables ----
// One Huffman table's codes BEHIND its look-ahead table, branch-free: lim[k] = the first code of length 10+k, left-aligned to
// 16 bits (= one past the last code of length 9+k; canonical codes make it monotonic), lim[7] = one past the last 16-bit code.
// With v = the next 16 bits: length = 10 + #{k in 1..6 : v >= lim[k]}, symbol = vals[vptr[length-10] + ((v - lim[length-10]) >> (16-length))]
struct alignas(16) JpegHuffTail {
  uint32_t lim[8];
  uint8_t vptr[8];                     // index in vals of the first symbol of length 10+k
  uint8_t pad_[8];
  uint8_t vals[256];
};
// The tables of one scan in the form the GPU decoder keeps in LDS (5 312 bytes): at most two DC and two AC tables (what
// every encoder in the field writes; a scan that names more stays on the host).  Look-ahead entries: (length << 8) | symbol,
// 0 = the code is longer than the look-ahead.  (measured, nine 12 MP photos: an 11-bit look-ahead for the AC tables - the codes
// behind a miss cost four dependent LDS reads - changed nothing, 1.19-1.25 vs 1.19-1.24 ms for the sync launches: what
// paces them is the chain of subsequences that hand a wrong state on, one decode time per link.  9 bits keep the writing
// kernel at four workgroups per CU.)
#ifndef IST_AC_LOOK_BITS
#define IST_AC_LOOK_BITS 9
#endif
constexpr int kJpegDcLookBits = 9, kJpegAcLookBits = IST_AC_LOOK_BITS;
struct alignas(16) JpegGpuTables {
  uint16_t look_dc[2][1 << kJpegDcLookBits];
  uint16_t look_ac[2][1 << kJpegAcLookBits];
  JpegHuffTail tail[4];                // 0-1: the DC tables, 2-3: the AC tables
};
// One restart interval of a scan (T.81 E.1.4: the entropy coder is reset at every RSTn, so an interval decodes on its own):
// its de-stuffed bytes start at byte_off of JpegGpuScan::stream (a multiple of 256, at least 16 zero bytes behind its last bit).
struct JpegGpuInterval { uint32_t byte_off, mcu0, n_mcus; int64_t bits; };
// The de-stuffed scan's bytes: a heap block that keeps its memory when it is cleared and can be handed from call to call
// (swap), written through data() by the de-stuffing loop.
class ScanBuf {
 public:
  ScanBuf() = default;
  ~ScanBuf() { std::free(p_); }
  ScanBuf(const ScanBuf&) = delete;
  ScanBuf& operator=(const ScanBuf&) = delete;
  ScanBuf(ScanBuf&& o) noexcept { swap(o); }
  ScanBuf& operator=(ScanBuf&& o) noexcept { swap(o); return *this; }
  void swap(ScanBuf& o) noexcept { std::swap(p_, o.p_); std::swap(n_, o.n_); std::swap(cap_, o.cap_); }
  const uint8_t* data() const { return p_; }
  uint8_t* data() { return p_; }
  size_t size() const { return n_; }
  size_t capacity() const { return cap_; }
  bool empty() const { return n_ == 0; }
  void clear() { n_ = 0; }
  void set_size(size_t n) { n_ = n; }                   // (n <= capacity: the writer filled data()[0, n))
  bool reserve(size_t want) {                           // contents are NOT kept (the writer starts over); false: out of memory
    if (want <= cap_) return true;
    std::free(p_);
    p_ = static_cast<uint8_t*>(std::malloc(want)); cap_ = p_ ? want : 0; n_ = 0;
    return p_ != nullptr;
  }
 private:
  uint8_t* p_ = nullptr; size_t n_ = 0, cap_ = 0;
};
struct JpegGpuScan {
  bool eligible = false;               // baseline, ONE interleaved scan over all components (with or without restart intervals)
  std::vector<JpegGpuInterval> iv;     // empty: the scan is one stream; otherwise one entry per restart interval, in order
  ScanBuf stream;                      // entropy-coded bytes with the FF00 stuffing removed, + 16 zero bytes
  int64_t bits = 0;                    // valid bits in stream
  int slots = 0;                       // blocks per MCU
  uint8_t slot_comp[10], slot_idx[10]; // per MCU slot: component, block index inside the component's h x v group
  uint8_t dc_tab[3], ac_tab[3];        // per component: which of the scan's two DC / two AC tables (0 or 1)
  JpegGpuTables tables;
};

// container parsing + Huffman decoding (host).  header_only stops after the frame header (size, sampling, orientation
// if the EXIF segment precedes it).  Returns IST_OK or an error code with the thread-local message set.
int jpeg_parse_and_entropy_decode(const uint8_t* file, int64_t len, JpegImage* out, bool header_only, JpegGpuScan* gpu_scan = nullptr);
// With gpu_scan: when the file qualifies (gpu_scan->eligible) the scan is NOT decoded on the host; the components then
// carry neither dense nor sparse coefficients and jpeg_gpu_entropy_decode fills the device planes.

// GPU stages (ist_jpeg_kernels.hip): coefficient planes (device) -> RGBA8 (device).  d_coef[c] / q_host[c] per component,
// planes = scratch for the reconstructed sample planes.  Asynchronous on `stream`.
struct JpegDeviceJob {
  int width, height, ncomp, hmax, vmax;
  int h[3], v[3], blocks_x[3], blocks_y[3];
  const int16_t* d_coef[3];
  const uint16_t* q_host[3];           // quantisation tables (HOST memory, natural order): passed by value to the kernel
  uint8_t* d_plane[3];                 // blocks_x*8 bytes per row (chroma only: the luma plane stays in LDS)
  uint8_t* out; size_t out_pitch;
  bool chroma_done = false;            // jpeg_launch_chroma_idct has already made d_plane[1], d_plane[2] on this stream
};
// chroma planes (one launch: components 1, 2 -> sample planes), then luma IDCT fused with upsampling + colour conversion
int jpeg_launch_reconstruct(const JpegDeviceJob& job, void* stream);
// the chroma planes of SEVERAL images in one launch (per 18 components): what the file pipeline runs once behind the Huffman
// batch, so that every image afterwards costs one (fused) launch
int jpeg_launch_chroma_idct(const JpegDeviceJob* jobs, int n_jobs, void* stream);
// d_coef (zeroed by the caller) <- the sparse entries of n_blocks blocks.  Asynchronous on `stream`.
int jpeg_launch_scatter(const uint32_t* d_ent, const uint32_t* d_start, const uint8_t* d_cnt, int16_t* d_coef, int n_blocks, void* stream);
// host-side expansion of one component to the dense form (tests, tools)
std::vector<int16_t> jpeg_dense_coefficients(const JpegComp& c);

// GPU entropy decoding of a batch of eligible images (self-synchronising parallel Huffman decoding).  d_coef[c] are the
// dense device planes (blocks_x*blocks_y*64 int16 each) the reconstruction kernels read.  ok[i] = 0 when image i failed
// the end-of-scan validation: the caller then decodes that image on the host.  Synchronises `stream`.
// d_stream (optional): the de-stuffed scan ALREADY on the device (256-byte aligned, S->stream.size() bytes, uploaded by the
// caller - e.g. by the image's parse thread while other images are still being parsed); NULL: uploaded here.
struct JpegGpuItem { const JpegImage* J; const JpegGpuScan* S; int16_t* d_coef[3]; const uint8_t* d_stream = nullptr; };
// scratch / scratch_bytes (optional): a grow-only device buffer the caller keeps across calls
int jpeg_gpu_entropy_decode(const std::vector<JpegGpuItem>& items, std::vector<uint8_t>* ok, void* stream, void** scratch = nullptr, size_t* scratch_bytes = nullptr);

}  // namespace ist

#endif  // IST_JPEG_H_
