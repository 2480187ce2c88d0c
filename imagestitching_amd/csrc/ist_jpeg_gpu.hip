// ist_jpeg_gpu.hip — JPEG entropy (Huffman) decoding on the GPU for baseline files: the serial half of the decode step
// (reference anchor: loadImageFrom, utils/canvas.js:27-121 — the platform decoder behind
// Image.src) made parallel.
//
// A Huffman-coded scan has no random access, but it is SELF-SYNCHRONISING: a decoder started at a wrong bit position
// falls into step with the true code boundaries after a few symbols.  The published scheme for GPUs (Weissenberger &
// Schmidt, "Massively Parallel Huffman Decoding on GPUs", ICPP 2018, and its JPEG follow-up) is restated here for this
// path:
//   * the de-stuffed scan is cut into subsequences of 2048 bits; one thread owns one subsequence;
//   * pass 0: every thread decodes its subsequence from its first bit, pretending a block starts there, and records
//     the state in which it crosses its end (bit position, MCU slot, zig-zag index) and how many blocks it finished;
//   * pass t >= 1: every thread decodes its subsequence again, starting from the exit state its LEFT neighbour
//     recorded in pass t-1.  Subsequence 0 starts from the true state, so after pass t subsequences 0..t are exact —
//     and because of self-synchronisation almost all others are too.  When a pass changes no exit state the states are
//     a fixed point, hence (by induction from subsequence 0) all exact; a subsequence whose start state did not change
//     since it was last decoded keeps its result, so the later passes touch only the few that are still settling;
//   * the passes of one workgroup (256 neighbouring subsequences) run INSIDE one launch, exit states exchanged through
//     LDS; only the hand-over between workgroups needs another launch (ist_jpeg_sync_kernel);
//   * an exclusive scan of the per-subsequence block counts gives every thread the index of its first block; a last
//     pass decodes once more and writes the coefficients straight into the dense planes in HBM; the DC differences
//     are integrated per component in decoding order by a block-wide scan.
// Nothing of the coefficients crosses PCIe: the upload is the compressed scan (1.8 MB for a 12 MP photo instead of a
// 36 MB plane).  The result is validated (exactly the blocks the frame header promises; no invalid code, run or DC
// category on the true path); an image that fails is reported back and decoded by the host decoder (ist_jpeg.cpp),
// which also produces the error message.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ist_internal.h"
#include "ist_jpeg.h"

namespace ist {

namespace {

constexpr int kSubBitsDefault = 2048;   // bits per subsequence (IST_JPEG_SUB_BITS overrides, tuning): measured on nine 12 MP
                                        // photos, whole call: 1024 bits 12 passes 4.5 ms, 2048 bits 7 passes 4.3 ms, 4096 bits 4 passes 5.4 ms
constexpr int kMaxPasses = 64;

__constant__ uint8_t kZig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct DevImg {
  const uint8_t* stream; int64_t bits;
  const JpegHuffTable* tables;            // 8 per image
  int32_t first_sub, n_sub;
  int32_t slots, total_blocks, mcus_x;
  uint8_t slot_comp[10], slot_idx[10], dc_tab[3], ac_tab[3];
  int16_t* coef[3];
  int32_t h[3], v[3], blocks_x[3];
  int32_t* dcdiff[3];                     // per component, decoding order
  int32_t n_dc[3];
  uint32_t* err;                          // set when the TRUE path meets what the host decoder would call corrupt data
};

struct State { uint32_t p; uint32_t c; uint32_t z; };

__device__ __forceinline__ uint32_t peek16(const uint8_t* s, uint32_t p) {
  const uint32_t b = p >> 3;
  const uint32_t w = (static_cast<uint32_t>(s[b]) << 16) | (static_cast<uint32_t>(s[b + 1]) << 8) | s[b + 2];
  return (w >> (8 - (p & 7))) & 0xFFFFu;
}

// symbol + its length; a bit pattern that is no code (only ever met on a speculative, out-of-phase path) consumes one bit
__device__ __forceinline__ int huff(const JpegHuffTable& h, uint32_t v16, uint32_t* len) {
  const uint32_t e = h.look[v16 >> 7];
  if (e) { *len = e >> 8; return static_cast<int>(e & 0xFF); }
  for (int l = 10; l <= 16; ++l) {
    const int code = static_cast<int>(v16 >> (16 - l));
    if (code <= h.maxcode[l]) { *len = l; return h.vals[(code + h.valoff[l]) & 255]; }
  }
  *len = 1;
  return -1;
}

__device__ __forceinline__ int extend_bits(uint32_t v, int s) { return (v < (1u << (s - 1))) ? static_cast<int>(v) - (1 << s) + 1 : static_cast<int>(v); }

// Decodes from state S until the bit position reaches `limit`.  WRITE: store coefficients (block index `blk` counts
// up from the caller's base).  Returns the number of blocks finished.
template <bool WRITE>
__device__ __forceinline__ uint32_t run(const DevImg& I, State& S, uint32_t limit, uint32_t blk) {
  uint32_t done = 0;
  const uint8_t* s = I.stream;
  while (S.p < limit) {
    const uint32_t comp = I.slot_comp[S.c];
    uint32_t len;
    if (S.z == 0) {
      const int t = huff(I.tables[I.dc_tab[comp]], peek16(s, S.p), &len);
      S.p += len;
      const int nb = (t > 0 && t <= 15) ? t : 0;
      if (WRITE && blk < static_cast<uint32_t>(I.total_blocks) && (t < 0 || t > 11)) *I.err = 1u;
      if (WRITE && blk < static_cast<uint32_t>(I.total_blocks)) {
        const int diff = nb ? extend_bits(peek16(s, S.p) >> (16 - nb), nb) : 0;
        const uint32_t mcu = blk / I.slots;
        const uint32_t k = mcu * (I.h[comp] * I.v[comp]) + I.slot_idx[S.c];        // decoding-order index inside the component
        I.dcdiff[comp][k] = diff;
      }
      S.p += nb;
      S.z = 1;
    } else {
      const int rs = huff(I.tables[I.ac_tab[comp]], peek16(s, S.p), &len);
      S.p += len;
      if (rs < 0) { if (WRITE && blk < static_cast<uint32_t>(I.total_blocks)) *I.err = 1u; continue; }   // out-of-phase garbage: stay in the block
      const uint32_t r = static_cast<uint32_t>(rs) >> 4, sz = static_cast<uint32_t>(rs) & 15;
      if (sz == 0) {
        if (r == 15) S.z += 16; else S.z = 64;                // ZRL / EOB
      } else {
        S.z += r;
        if (WRITE && S.z >= 64 && blk < static_cast<uint32_t>(I.total_blocks)) *I.err = 1u;          // run past the block
        if (WRITE && S.z < 64 && blk < static_cast<uint32_t>(I.total_blocks)) {
          const int val = extend_bits(peek16(s, S.p) >> (16 - sz), static_cast<int>(sz));
          const uint32_t mcu = blk / I.slots;
          const uint32_t mx = mcu % I.mcus_x, my = mcu / I.mcus_x;
          const uint32_t si = I.slot_idx[S.c];
          const uint32_t bx = mx * I.h[comp] + si % I.h[comp], by = my * I.v[comp] + si / I.h[comp];
          I.coef[comp][(static_cast<size_t>(by) * I.blocks_x[comp] + bx) * 64 + kZig[S.z]] = static_cast<int16_t>(val);
        }
        S.p += sz;
        S.z += 1;
      }
    }
    if (S.z >= 64) {                                          // block finished
      S.z = 0;
      S.c = (S.c + 1 == static_cast<uint32_t>(I.slots)) ? 0u : S.c + 1;
      ++done; ++blk;
    }
  }
  return done;
}

struct SyncArgs {
  const DevImg* imgs; const uint16_t* sub_img;
  const uint32_t* in_p; const uint32_t* in_cz;       // exit states of the previous pass
  uint32_t* out_p; uint32_t* out_cz; uint32_t* nblk;
  uint32_t* start_p; uint32_t* start_cz;               // the start state each subsequence was last decoded from
  uint32_t* changed; int32_t n_sub_total; int32_t pass; int32_t sub_bits;
};

// One LAUNCH = as many synchronisation passes as its workgroup needs: the 256 subsequences of a workgroup exchange exit
// states through LDS and iterate (re-decode whoever's start state changed, barrier) until none of them changes, so a
// chain of out-of-phase subsequences is chased to its end inside one launch instead of one subsequence per
// host-synchronised pass.  Across workgroups the first thread starts from the exit state its left neighbour reached in
// the PREVIOUS launch; a launch in which no workgroup's last exit state changed is the global fixed point.  (The
// "overflow" idea of the published scheme, restated for a barrier-synchronised workgroup.)  Measured on nine 12 MP
// photos: fixed point after 2-3 launches instead of 7.
constexpr int kInnerPasses = 48;

__global__ __launch_bounds__(256) void ist_jpeg_sync_kernel(const SyncArgs A) {
  __shared__ uint32_t ex_p[256], ex_cz[256];
  const int tid = threadIdx.x;
  const int g = blockIdx.x * blockDim.x + tid;
  const bool live = g < A.n_sub_total;
  const DevImg* Ip = live ? &A.imgs[A.sub_img[g]] : &A.imgs[0];
  const uint32_t i = live ? static_cast<uint32_t>(g - Ip->first_sub) : 0u;
  uint32_t limit = 0;
  if (live) {
    const uint64_t end = static_cast<uint64_t>(i + 1) * static_cast<uint64_t>(A.sub_bits);
    limit = static_cast<uint32_t>(end < static_cast<uint64_t>(Ip->bits) ? end : static_cast<uint64_t>(Ip->bits));
  }
  // what this subsequence last decoded from, and to (carried across launches in start_* / in_*)
  bool have = live && A.pass > 0;
  uint32_t st_p = have ? A.start_p[g] : 0u, st_cz = have ? A.start_cz[g] : 0u;
  uint32_t my_p = have ? A.in_p[g] : 0u, my_cz = have ? A.in_cz[g] : 0u, nb = have ? A.nblk[g] : 0u;
  bool settled = false;
  for (int it = 0; it < kInnerPasses; ++it) {
    uint32_t sp = 0, scz = 0;
    if (live && i != 0) {                            // (the first subsequence of an image starts from the true state 0, 0, 0)
      if (it == 0 || tid == 0) {                     // from the previous launch (or, in the very first pass, a guess: a block starts here)
        if (A.pass == 0) { sp = i * static_cast<uint32_t>(A.sub_bits); scz = 0; }
        else { sp = A.in_p[g - 1]; scz = A.in_cz[g - 1]; }
      } else { sp = ex_p[tid - 1]; scz = ex_cz[tid - 1]; }      // from the left neighbour, this launch
    }
    const bool redo = live && !(have && st_p == sp && st_cz == scz);
    if (redo) {
      State S; S.p = sp; S.c = scz >> 8; S.z = scz & 255u;
      nb = run<false>(*Ip, S, limit, 0);
      my_p = S.p; my_cz = (S.c << 8) | S.z;
      st_p = sp; st_cz = scz; have = true;
    }
    __syncthreads();                                 // every thread has read its neighbour's previous exit state
    ex_p[tid] = my_p; ex_cz[tid] = my_cz;
    if (!__syncthreads_or(redo ? 1 : 0)) { settled = true; break; }
  }
  if (!live) return;
  // the launch changed something the NEXT workgroup depends on (or ran out of inner passes): not the fixed point yet
  const bool last = tid == 255 || g == A.n_sub_total - 1;
  if (A.pass == 0 || !settled || (last && (A.in_p[g] != my_p || A.in_cz[g] != my_cz))) {
    if (A.pass == 0 ? (tid == 0) : true) *A.changed = 1u;
  }
  A.out_p[g] = my_p; A.out_cz[g] = my_cz; A.nblk[g] = nb;
  A.start_p[g] = st_p; A.start_cz[g] = st_cz;
}

struct WriteArgs { const DevImg* imgs; const uint16_t* sub_img; const uint32_t* p; const uint32_t* cz; const uint32_t* blk_excl; int32_t n_sub_total; int32_t sub_bits; };

__global__ __launch_bounds__(256) void ist_jpeg_write_kernel(const WriteArgs A) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= A.n_sub_total) return;
  const DevImg& I = A.imgs[A.sub_img[g]];
  const uint32_t i = static_cast<uint32_t>(g - I.first_sub);
  State S;
  if (i == 0) { S.p = 0; S.c = 0; S.z = 0; } else { S.p = A.p[g - 1]; S.c = A.cz[g - 1] >> 8; S.z = A.cz[g - 1] & 255u; }
  const uint64_t end = static_cast<uint64_t>(i + 1) * static_cast<uint64_t>(A.sub_bits);
  const uint32_t limit = static_cast<uint32_t>(end < static_cast<uint64_t>(I.bits) ? end : static_cast<uint64_t>(I.bits));
  run<true>(I, S, limit, A.blk_excl[g] - A.blk_excl[I.first_sub]);
}

// exclusive scan of n uint32 by ONE workgroup (n is ~130 k for nine 12 MP photos): chunk per thread, block scan, fix-up
__global__ __launch_bounds__(1024) void ist_scan_u32_kernel(const uint32_t* in, uint32_t* out, int n) {
  __shared__ uint32_t part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int a = tid * per, b = min(n, a + per);
  uint32_t s = 0;
  for (int i = a; i < b; ++i) s += in[i];
  part[tid] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t t = tid >= off ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += t;
    __syncthreads();
  }
  uint32_t run = tid ? part[tid - 1] : 0u;
  for (int i = a; i < b; ++i) { const uint32_t v = in[i]; out[i] = run; run += v; }
  if (tid == 1023) out[n] = part[1023];                       // total at [n]
}

// DC prediction: inclusive sum of the differences in decoding order, written to coefficient 0 of every block.
// One workgroup per (image, component).
__global__ __launch_bounds__(1024) void ist_jpeg_dc_kernel(const DevImg* imgs) {
  __shared__ int32_t part[1024];
  const DevImg& I = imgs[blockIdx.x / 3];
  const int comp = blockIdx.x % 3;
  const int n = I.n_dc[comp];
  if (n <= 0) return;
  const int32_t* d = I.dcdiff[comp];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int a = tid * per, b = min(n, a + per);
  int32_t s = 0;
  for (int i = a; i < b; ++i) s += d[i];
  part[tid] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int32_t t = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += t;
    __syncthreads();
  }
  int32_t run = tid ? part[tid - 1] : 0;
  const int bc = I.h[comp] * I.v[comp];
  for (int k = a; k < b; ++k) {
    run += d[k];
    const int mcu = k / bc, si = k - mcu * bc;
    const int mx = mcu % I.mcus_x, my = mcu / I.mcus_x;
    const int bx = mx * I.h[comp] + si % I.h[comp], by = my * I.v[comp] + si / I.h[comp];
    I.coef[comp][(static_cast<size_t>(by) * I.blocks_x[comp] + bx) * 64] = static_cast<int16_t>(run);
  }
}

}  // namespace

int jpeg_gpu_entropy_decode(const std::vector<JpegGpuItem>& items, std::vector<uint8_t>* ok, void* stream_, void** scratch, size_t* scratch_bytes) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const size_t n_img = items.size();
  static const bool timing = std::getenv("IST_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    (void)hipStreamSynchronize(stream);
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[ist timing]   huffman/%-20s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  int kSubBits = kSubBitsDefault;
  if (const char* e = std::getenv("IST_JPEG_SUB_BITS")) { const int v = std::atoi(e); if (v >= 256 && v <= 65536) kSubBits = v; }
  ok->assign(n_img, 0);
  if (n_img == 0) return IST_OK;
#define JG_HIP(e) do { const hipError_t e_ = (e); if (e_ != hipSuccess) return fail(IST_E_HIP, std::string(#e) + ": " + hipGetErrorString(e_)); } while (0)
  // ---- one device arena: streams, tables, image records, per-subsequence state, DC differences
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~static_cast<size_t>(255); return at; };
  std::vector<size_t> o_stream(n_img), o_tab(n_img);
  std::vector<size_t> o_dc(n_img * 3, 0);
  std::vector<DevImg> H(n_img);
  int64_t n_sub_total = 0;
  for (size_t k = 0; k < n_img; ++k) {
    const JpegImage& J = *items[k].J; const JpegGpuScan& S = *items[k].S;
    if (S.bits >= (1ll << 32) - 65536) return fail(IST_E_UNSUPPORTED, "JPEG scan too large for the GPU entropy decoder");
    // the kernels index slot_comp / slot_idx (10 entries, T.81 B.2.3) with the MCU slot: never launch outside that
    if (S.slots < 1 || S.slots > 10) return fail(IST_E_DECODE, "JPEG scan with more than 10 blocks per MCU");
    o_stream[k] = take(S.stream.size());
    o_tab[k] = take(sizeof(S.tables));
    DevImg& I = H[k];
    std::memset(&I, 0, sizeof I);
    I.bits = S.bits;
    I.first_sub = static_cast<int32_t>(n_sub_total);
    I.n_sub = static_cast<int32_t>((S.bits + kSubBits - 1) / kSubBits);
    if (I.n_sub < 1) I.n_sub = 1;
    n_sub_total += I.n_sub;
    I.slots = S.slots; I.mcus_x = J.mcus_x;
    I.total_blocks = J.mcus_x * J.mcus_y * S.slots;
    std::memcpy(I.slot_comp, S.slot_comp, 10); std::memcpy(I.slot_idx, S.slot_idx, 10);
    std::memcpy(I.dc_tab, S.dc_tab, 3); std::memcpy(I.ac_tab, S.ac_tab, 3);
    for (int c = 0; c < J.ncomp; ++c) {
      I.coef[c] = items[k].d_coef[c];
      I.h[c] = J.comp[c].h; I.v[c] = J.comp[c].v; I.blocks_x[c] = J.comp[c].blocks_x;
      I.n_dc[c] = J.mcus_x * J.mcus_y * J.comp[c].h * J.comp[c].v;
      o_dc[k * 3 + c] = take(static_cast<size_t>(I.n_dc[c]) * 4);
    }
    for (int c = J.ncomp; c < 3; ++c) { I.h[c] = I.v[c] = 1; I.n_dc[c] = 0; }
  }
  if (n_sub_total >= (1ll << 31) || n_img > 65535) return fail(IST_E_UNSUPPORTED, "too much JPEG data for one GPU entropy-decode batch");
  const int ns = static_cast<int>(n_sub_total);
  const size_t o_img = take(sizeof(DevImg) * n_img), o_sub = take(2 * static_cast<size_t>(ns));
  const size_t o_p0 = take(4 * static_cast<size_t>(ns)), o_p1 = take(4 * static_cast<size_t>(ns)), o_cz0 = take(4 * static_cast<size_t>(ns)), o_cz1 = take(4 * static_cast<size_t>(ns));
  const size_t o_sp = take(4 * static_cast<size_t>(ns)), o_scz = take(4 * static_cast<size_t>(ns));
  const size_t o_nblk = take(4 * static_cast<size_t>(ns)), o_excl = take(4 * (static_cast<size_t>(ns) + 1)), o_flag = take(4), o_err = take(4 * n_img);
  // the caller's grow-only scratch (a context keeps it across calls: no allocation, and no implicit device synchronisation
  // of a free, per call), or a one-off allocation
  uint8_t* d = nullptr;
  struct Free { void* p; ~Free() { if (p) (void)hipFree(p); } } fr{nullptr};
  if (scratch && scratch_bytes) {
    if (*scratch_bytes < off) {
      if (*scratch) { (void)hipFree(*scratch); *scratch = nullptr; *scratch_bytes = 0; }
      JG_HIP(hipMalloc(scratch, off + off / 4));
      *scratch_bytes = off + off / 4;
    }
    d = static_cast<uint8_t*>(*scratch);
  } else {
    JG_HIP(hipMalloc(reinterpret_cast<void**>(&d), off));
    fr.p = d;
  }
  std::vector<uint16_t> sub_img(static_cast<size_t>(ns));
  for (size_t k = 0; k < n_img; ++k) {
    const JpegGpuScan& S = *items[k].S;
    JG_HIP(hipMemcpyAsync(d + o_stream[k], S.stream.data(), S.stream.size(), hipMemcpyHostToDevice, stream));
    JG_HIP(hipMemcpyAsync(d + o_tab[k], S.tables, sizeof(S.tables), hipMemcpyHostToDevice, stream));
    H[k].stream = d + o_stream[k];
    H[k].tables = reinterpret_cast<const JpegHuffTable*>(d + o_tab[k]);
    for (int c = 0; c < 3; ++c) H[k].dcdiff[c] = reinterpret_cast<int32_t*>(d + o_dc[k * 3 + c]);
    H[k].err = reinterpret_cast<uint32_t*>(d + o_err) + k;
    for (int i = 0; i < H[k].n_sub; ++i) sub_img[static_cast<size_t>(H[k].first_sub + i)] = static_cast<uint16_t>(k);
    // the planes are written sparsely: zero them first
    const JpegImage& J = *items[k].J;
    for (int c = 0; c < J.ncomp; ++c)
      JG_HIP(hipMemsetAsync(items[k].d_coef[c], 0, static_cast<size_t>(J.comp[c].blocks_x) * J.comp[c].blocks_y * 128, stream));
  }
  JG_HIP(hipMemsetAsync(d + o_err, 0, 4 * n_img, stream));
  JG_HIP(hipMemcpyAsync(d + o_img, H.data(), sizeof(DevImg) * n_img, hipMemcpyHostToDevice, stream));
  JG_HIP(hipMemcpyAsync(d + o_sub, sub_img.data(), 2 * static_cast<size_t>(ns), hipMemcpyHostToDevice, stream));
  lap("alloc + uploads");
  const DevImg* d_img = reinterpret_cast<const DevImg*>(d + o_img);
  const uint16_t* d_sub = reinterpret_cast<const uint16_t*>(d + o_sub);
  uint32_t* P[2] = {reinterpret_cast<uint32_t*>(d + o_p0), reinterpret_cast<uint32_t*>(d + o_p1)};
  uint32_t* CZ[2] = {reinterpret_cast<uint32_t*>(d + o_cz0), reinterpret_cast<uint32_t*>(d + o_cz1)};
  uint32_t* d_nblk = reinterpret_cast<uint32_t*>(d + o_nblk);
  uint32_t* d_excl = reinterpret_cast<uint32_t*>(d + o_excl);
  uint32_t* d_flag = reinterpret_cast<uint32_t*>(d + o_flag);
  const unsigned grid = static_cast<unsigned>((ns + 255) / 256);
  // ---- synchronisation passes until a pass changes nothing
  int cur = 0, passes_run = 0;
  bool converged = false;
  for (int pass = 0; pass < kMaxPasses; ++pass) {
    passes_run = pass + 1;
    JG_HIP(hipMemsetAsync(d_flag, 0, 4, stream));
    SyncArgs A{d_img, d_sub, P[cur], CZ[cur], P[cur ^ 1], CZ[cur ^ 1], d_nblk, reinterpret_cast<uint32_t*>(d + o_sp), reinterpret_cast<uint32_t*>(d + o_scz), d_flag, ns, pass, kSubBits};
    hipLaunchKernelGGL(ist_jpeg_sync_kernel, dim3(grid), dim3(256), 0, stream, A);
    JG_HIP(hipGetLastError());
    cur ^= 1;
    if (pass == 0) continue;                       // (the first launch starts from guesses: a second one always runs)
    uint32_t flag = 1;
    JG_HIP(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, stream));
    JG_HIP(hipStreamSynchronize(stream));
    if (!flag) { converged = true; break; }
  }
  if (std::getenv("IST_TIMING")) std::fprintf(stderr, "[ist timing] GPU Huffman: %d subsequences, %s after %d passes\n", ns, converged ? "fixed point" : "NO fixed point", passes_run);
  if (!converged) { JG_HIP(hipStreamSynchronize(stream)); return IST_OK; }      // every ok[] stays 0: the host decodes
  lap("sync passes");
  // ---- block indices, coefficient write, DC integration
  hipLaunchKernelGGL(ist_scan_u32_kernel, dim3(1), dim3(1024), 0, stream, d_nblk, d_excl, ns);
  JG_HIP(hipGetLastError());
  WriteArgs W{d_img, d_sub, P[cur], CZ[cur], d_excl, ns, kSubBits};
  hipLaunchKernelGGL(ist_jpeg_write_kernel, dim3(grid), dim3(256), 0, stream, W);
  JG_HIP(hipGetLastError());
  hipLaunchKernelGGL(ist_jpeg_dc_kernel, dim3(static_cast<unsigned>(3 * n_img)), dim3(1024), 0, stream, d_img);
  JG_HIP(hipGetLastError());
  lap("scan + write + DC");
  // ---- validation: exactly the blocks the frame header promises, and nothing the host decoder would reject
  std::vector<uint32_t> excl(static_cast<size_t>(ns) + 1), err(n_img);
  JG_HIP(hipMemcpyAsync(excl.data(), d_excl, 4 * (static_cast<size_t>(ns) + 1), hipMemcpyDeviceToHost, stream));
  JG_HIP(hipMemcpyAsync(err.data(), d + o_err, 4 * n_img, hipMemcpyDeviceToHost, stream));
  JG_HIP(hipStreamSynchronize(stream));
  for (size_t k = 0; k < n_img; ++k) {
    const DevImg& I = H[k];
    const uint32_t blocks = excl[static_cast<size_t>(I.first_sub + I.n_sub)] - excl[static_cast<size_t>(I.first_sub)];
    (*ok)[k] = (blocks == static_cast<uint32_t>(I.total_blocks) && err[k] == 0) ? 1 : 0;
  }
  lap("validation");
#undef JG_HIP
  return IST_OK;
}

}  // namespace ist
