// ist_png.hip — lossless PNG export of an RGBA8 canvas that is resident in HBM.
//
// Reference anchor: the export step of onStitch — safeCanvasToTempFilePath(canvas, 'png', W, H, W, H) ->
// wx.canvasToTempFilePath({fileType:'png', quality:1}) (utils/canvas.js:205-242, pages/index/index.js:1577-1579).
// The reference's encoder is the WeChat client; any PNG that decodes to the same pixels is the same result, so this
// is a design for the hardware, not a restatement:
//
//   * colour type 6 (RGBA, 8 bit), filter 0 on every row, zlib stream made of STORED deflate blocks (BTYPE=00):
//     the payload is the canvas bytes themselves, so the encoder is one HBM-bound pass (read canvas, write file).
//   * every row's pixel bytes start at a 16-byte aligned file offset: the 5-byte block header + 1 filter byte are
//     preceded by k empty stored blocks (5 bytes each; 5 is coprime with 16, k <= 15) chosen so that the pixels land
//     aligned -> the copy is 16-B loads / 16-B stores like the stitch kernel's COPY path.  Rows longer than 63 KiB are
//     split into 63 KiB blocks (15 empties + header = 80 bytes between blocks keep the alignment).
//   * Adler-32 (over the filtered rows) and CRC-32 (over the IDAT bytes) are computed IN the copy pass: each lane
//     folds its 16 bytes (byte sums for Adler; slicing-by-4 table CRC from LDS, then a GF(2) multiply by
//     x^(8*bytes_after) so that lanes combine by XOR), waves reduce, one atomic per wave.  The host combines the per-row
//     partials (linear algebra over GF(2) / mod 65521, O(rows)) and patches the ~60 header/trailer bytes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "ist_crc.h"
#include "ist_internal.h"

namespace ist {

namespace {

constexpr int kChunks = 4;                       // 16-byte chunks per lane per workgroup (16 KiB of a row per workgroup)
constexpr int64_t kBlockData = 64512;            // pixel bytes per stored block: 63 KiB, a multiple of 1 KiB (wave-row)
// IDAT chunk data limit (PNG allows 2^31-1); IST_PNG_IDAT_LIMIT lowers it so that tests can exercise multi-chunk files
static int64_t idat_limit() {
  const char* e = std::getenv("IST_PNG_IDAT_LIMIT");
  const int64_t v = e ? std::atoll(e) : 0;
  return v >= 4096 ? v : (1ll << 30);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

struct PngArgs {
  const uint8_t* canvas; size_t pitch;
  uint8_t* out;
  const int64_t* row_tab;        // per row: file offset of the first pixel byte (16-B aligned) | k (low 4 bits)
  const uint32_t* xpow4;         // x^(32 i) mod P: shift of a CRC register over 4 i zero bytes
  const uint32_t* tables;        // 4 x 256 slicing tables
  unsigned long long* s1; unsigned long long* s2;   // Adler partials per row
  uint32_t* crc;                 // raw CRC per (row, block)
  int64_t row_bytes; int32_t h; int32_t nb;
  uint32_t segs;                 // 16-KiB segments per row
};

__global__ __launch_bounds__(256) void ist_png_rows_kernel(const PngArgs P) {
  __shared__ uint32_t T[4][256];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 1024; i += 256) T[i >> 8][i & 255] = P.tables[i];
  __syncthreads();
  // one workgroup per 16-KiB row segment, numbered along the row first (contiguous file bytes for neighbouring workgroups)
  const int r = static_cast<int>(blockIdx.x / P.segs);
  const int seg = static_cast<int>(blockIdx.x - static_cast<unsigned>(r) * P.segs);
  const int64_t tab = P.row_tab[r];
  const int64_t pix_off = tab & ~15ll;
  const int k = static_cast<int>(tab & 15);

  // ---- the row's framing bytes: k empty stored blocks + block header + filter byte, then 80 bytes between blocks
  if (seg == 0 && tid < 64) {
    const bool last_row = r == P.h - 1;
    const int lead = 5 * k + 6;
    for (int i = lane; i < lead; i += 64) {
      uint8_t v;
      if (i < 5 * k) { const int j = i % 5; v = (j >= 3) ? 0xFF : 0x00; }             // 00 | 00 00 | FF FF
      else {
        const int j = i - 5 * k;
        const int64_t bl = P.row_bytes < kBlockData ? P.row_bytes : kBlockData;
        const uint32_t len = static_cast<uint32_t>(bl + 1);                             // filter byte + pixel bytes
        if (j == 0) v = (last_row && P.nb == 1) ? 0x01 : 0x00;                        // BFINAL on the stream's last block
        else if (j == 1) v = len & 0xFF; else if (j == 2) v = (len >> 8) & 0xFF;
        else if (j == 3) v = (~len) & 0xFF; else if (j == 4) v = ((~len) >> 8) & 0xFF;
        else v = 0x00;                                                                  // filter type 0 (None)
      }
      P.out[pix_off - lead + i] = v;
    }
    for (int b = 1; b < P.nb; ++b) {
      const int64_t at = pix_off + b * kBlockData + 80 * (b - 1);
      const int64_t left = P.row_bytes - b * kBlockData;
      const uint32_t len = static_cast<uint32_t>(left < kBlockData ? left : kBlockData);
      for (int i = lane; i < 80; i += 64) {
        uint8_t v;
        if (i < 75) { const int j = i % 5; v = (j >= 3) ? 0xFF : 0x00; }
        else {
          const int j = i - 75;
          if (j == 0) v = (last_row && b == P.nb - 1) ? 0x01 : 0x00;
          else if (j == 1) v = len & 0xFF; else if (j == 2) v = (len >> 8) & 0xFF;
          else if (j == 3) v = (~len) & 0xFF; else v = ((~len) >> 8) & 0xFF;
        }
        P.out[at + i] = v;
      }
    }
  }

  // ---- pixels: kChunks x 16 bytes per lane (4 KiB apart: every pass of the workgroup is one contiguous 4 KiB), copy +
  // checksums.  Several passes per workgroup amortise the table staging above.
#pragma unroll 1
  for (int u = 0; u < kChunks; ++u) {
  const int64_t p = ((static_cast<int64_t>(seg) * kChunks + u) * 256 + tid) * 16;   // byte position of this lane's chunk in the row
  unsigned long long a1 = 0, a2 = 0;
  uint32_t c = 0;
  int b = 0;
  if (p < P.row_bytes) {
    const int nbytes = static_cast<int>(P.row_bytes - p < 16 ? P.row_bytes - p : 16);      // 4, 8, 12 only at a ragged row end
    b = static_cast<int>(p / kBlockData);
    const int64_t in_block = p - b * kBlockData;
    const int64_t bl = (P.row_bytes - b * kBlockData) < kBlockData ? (P.row_bytes - b * kBlockData) : kBlockData;
    const int64_t after = bl - in_block - nbytes;                                          // bytes of this block behind the chunk
    const uint8_t* src = P.canvas + static_cast<size_t>(r) * P.pitch + p;
    uint8_t* dst = P.out + pix_off + p + 80ll * b;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (nbytes == 16) {
      v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4*>(src));
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
    } else {
      v.x = *reinterpret_cast<const uint32_t*>(src); *reinterpret_cast<uint32_t*>(dst) = v.x;
      if (nbytes > 4) { v.y = *reinterpret_cast<const uint32_t*>(src + 4); *reinterpret_cast<uint32_t*>(dst + 4) = v.y; }
      if (nbytes > 8) { v.z = *reinterpret_cast<const uint32_t*>(src + 8); *reinterpret_cast<uint32_t*>(dst + 8) = v.z; }
    }
    // Adler: byte j of the chunk sits at index q = 1 + p + j of the row stream (length L = row_bytes + 1) and weighs L - q
    uint32_t t1 = 0, t2 = 0;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};              // bytes past nbytes are zero
#pragma unroll
    for (int d = 0; d < 4; ++d) {                            // v_dot4_u32_u8: byte sums and index-weighted byte sums
      t1 = __builtin_amdgcn_udot4(w[d], 0x01010101u, t1, false);
      t2 = __builtin_amdgcn_udot4(w[d], static_cast<uint32_t>(4 * d) * 0x01010101u + 0x03020100u, t2, false);
    }
    a1 = t1;
    a2 = static_cast<unsigned long long>(P.row_bytes - p) * t1 - t2;
    // raw CRC of the chunk (register starts at 0), slicing-by-4
    const int nd = nbytes >> 2;
    for (int d = 0; d < nd; ++d) {
      c ^= w[d];
      c = T[3][c & 0xFF] ^ T[2][(c >> 8) & 0xFF] ^ T[1][(c >> 16) & 0xFF] ^ T[0][c >> 24];
    }
    c = gf_mul(P.xpow4[after >> 2], c);                     // as if `after` zero bytes followed: lanes now combine by XOR
  }
  // ---- wave reduction (a wave never straddles a row or a 63-KiB block), one atomic per wave
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off); c ^= __shfl_xor(c, off);
  }
  const int bw = __builtin_amdgcn_readfirstlane(b);
  if (lane == 0 && (p < P.row_bytes)) {
    atomicAdd(&P.s1[r], a1); atomicAdd(&P.s2[r], a2);
    atomicXor(&P.crc[static_cast<size_t>(r) * P.nb + bw], c);
  }
  }
}

struct Layout {
  int64_t w = 0, h = 0, row_bytes = 0; int nb = 1;
  std::vector<int64_t> row_tab;          // pix_off | k
  std::vector<int64_t> chunk_first_row;  // IDAT chunks: first row of each
  int64_t total = 0;                     // file size
};

// one pass over the rows, mirroring exactly what the kernel and the patcher emit
void make_layout(int64_t w, int64_t h, Layout* L) {
  L->w = w; L->h = h; L->row_bytes = 4 * w;
  L->nb = static_cast<int>((L->row_bytes + kBlockData - 1) / kBlockData);
  L->row_tab.resize(static_cast<size_t>(h));
  L->chunk_first_row.clear();
  int64_t pos = 0, chunk_data = 0;
  const int64_t limit = idat_limit();
  for (int64_t r = 0; r < h; ++r) {
    const int64_t rec_upper = 12 + 81 + L->row_bytes + 80ll * (L->nb - 1);
    int pre = 0;
    if (r == 0) { pre = 8 + 25 + 8 + 2; L->chunk_first_row.push_back(0); chunk_data = 2; }
    else if (chunk_data + rec_upper > limit) { pre = 12; L->chunk_first_row.push_back(r); chunk_data = 0; }
    int k = 0;
    while ((pos + pre + 5 * k + 6) % 16 != 0) ++k;
    const int64_t pix = pos + pre + 5 * k + 6;
    L->row_tab[static_cast<size_t>(r)] = pix | k;
    const int64_t end = pix + L->row_bytes + 80ll * (L->nb - 1);
    chunk_data += end - pos - pre;
    pos = end;
  }
  L->total = pos + 4 /*adler*/ + 4 /*crc*/ + 12 /*IEND*/;
}

void put32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = (v >> 16) & 0xFF; p[2] = (v >> 8) & 0xFF; p[3] = v & 0xFF; }

}  // namespace

}  // namespace ist

using namespace ist;

extern "C" {

int64_t ist_png_bound(int64_t w, int64_t h) {
  if (w < 1 || h < 1) return 0;
  const int64_t row = 4 * w, nb = (row + kBlockData - 1) / kBlockData;
  const int64_t stored = 64 + h * (row + 81 + 80 * (nb - 1) + 16) + (h * (row + 200) / idat_limit() + 2) * 12 + 32;
  return std::max(stored, png_deflate_bound(w, h));        // either encoder fits
}

int ist_png_encode_device(ist_ctx* ctx, const void* canvas, size_t pitch, int64_t w, int64_t h, void* out, int64_t out_cap,
                          int64_t* out_len, void* stream_) {
  if (!ctx) return fail(IST_E_NO_CONTEXT, "无法获取绘图上下文");
  if (!canvas || !out || !out_len || w < 1 || h < 1 || pitch < static_cast<size_t>(w) * 4 || (pitch & 3))
    return fail(IST_E_INVALID, "ist_png_encode_device: bad argument");
  if (w > (1ll << 29) || h > 2147483647ll) return fail(IST_E_OUTPUT_SIZE, "image too large for PNG");
  if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) return fail(IST_E_INVALID, "PNG output buffer must be 16-byte aligned");
  if (ctx_png_level(ctx) > 0) return png_encode_device_deflate(ctx, canvas, pitch, w, h, out, out_cap, out_len, stream_, nullptr, nullptr);
  Layout L;
  make_layout(w, h, &L);
  if (L.total > out_cap) return fail(IST_E_INVALID, "PNG output buffer too small (see ist_png_bound)");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int dev = 0;
  (void)hipGetDevice(&dev);

  static CrcTables T;
  static std::once_flag tables_once;
  std::call_once(tables_once, []() { make_crc_tables(&T); });
  // x^(32 i): the register after 4 i zero bytes, starting from the polynomial "1"
  const int64_t n_pow = kBlockData / 4 + 1;
  std::vector<uint32_t> xpow(static_cast<size_t>(n_pow));
  {
    uint32_t reg = 0x80000000u;
    for (int64_t i = 0; i < n_pow; ++i) {
      xpow[static_cast<size_t>(i)] = reg;
      for (int z = 0; z < 4; ++z) reg = crc_byte(T, reg, 0);
    }
  }
  // device scratch: row table, xpow, tables, partial sums
  const size_t n_acc = static_cast<size_t>(h) * L.nb;
  const size_t bytes_tab = sizeof(int64_t) * static_cast<size_t>(h), bytes_pow = 4 * xpow.size(), bytes_T = sizeof(T);
  const size_t bytes_s = 8 * static_cast<size_t>(h), bytes_crc = 4 * n_acc;
  auto up16 = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  const size_t o_tab = 0, o_pow = o_tab + up16(bytes_tab), o_T = o_pow + up16(bytes_pow), o_s1 = o_T + up16(bytes_T),
               o_s2 = o_s1 + up16(bytes_s), o_crc = o_s2 + up16(bytes_s), scratch_bytes = o_crc + up16(bytes_crc);
  uint8_t* scratch = nullptr;
  if (dev_malloc(reinterpret_cast<void**>(&scratch), scratch_bytes) != hipSuccess) return fail(IST_E_NOMEM, "PNG scratch allocation failed");
  struct Free { void* p; ~Free() { dev_free(p); } } fr{scratch};    // (a free waits for the device: the early returns below are safe)
#define PNG_HIP(e) do { const hipError_t e_ = (e); if (e_ != hipSuccess) return fail(IST_E_HIP, std::string(#e) + ": " + hipGetErrorString(e_)); } while (0)
  PNG_HIP(hipMemcpyAsync(scratch + o_tab, L.row_tab.data(), bytes_tab, hipMemcpyHostToDevice, stream));
  PNG_HIP(hipMemcpyAsync(scratch + o_pow, xpow.data(), bytes_pow, hipMemcpyHostToDevice, stream));
  PNG_HIP(hipMemcpyAsync(scratch + o_T, &T, bytes_T, hipMemcpyHostToDevice, stream));
  PNG_HIP(hipMemsetAsync(scratch + o_s1, 0, scratch_bytes - o_s1, stream));
  PngArgs A;
  A.canvas = static_cast<const uint8_t*>(canvas); A.pitch = pitch; A.out = static_cast<uint8_t*>(out);
  A.row_tab = reinterpret_cast<const int64_t*>(scratch + o_tab);
  A.xpow4 = reinterpret_cast<const uint32_t*>(scratch + o_pow);
  A.tables = reinterpret_cast<const uint32_t*>(scratch + o_T);
  A.s1 = reinterpret_cast<unsigned long long*>(scratch + o_s1);
  A.s2 = reinterpret_cast<unsigned long long*>(scratch + o_s2);
  A.crc = reinterpret_cast<uint32_t*>(scratch + o_crc);
  A.row_bytes = L.row_bytes; A.h = static_cast<int32_t>(h); A.nb = L.nb;
  const unsigned gx = static_cast<unsigned>((L.row_bytes + 4096 * kChunks - 1) / (4096 * kChunks));
  if (static_cast<int64_t>(gx) * h > 2147483647ll) return fail(IST_E_OUTPUT_SIZE, "image too large for one PNG launch");
  A.segs = gx;
  hipLaunchKernelGGL(ist_png_rows_kernel, dim3(static_cast<unsigned>(gx * static_cast<unsigned>(h))), dim3(256), 0, stream, A);
  PNG_HIP(hipGetLastError());
  // ---- combine the partials on the host
  std::vector<unsigned long long> s1(static_cast<size_t>(h)), s2(static_cast<size_t>(h));
  std::vector<uint32_t> crc(n_acc);
  PNG_HIP(hipMemcpyAsync(s1.data(), scratch + o_s1, bytes_s, hipMemcpyDeviceToHost, stream));
  PNG_HIP(hipMemcpyAsync(s2.data(), scratch + o_s2, bytes_s, hipMemcpyDeviceToHost, stream));
  PNG_HIP(hipMemcpyAsync(crc.data(), scratch + o_crc, bytes_crc, hipMemcpyDeviceToHost, stream));
  PNG_HIP(hipStreamSynchronize(stream));

  const uint64_t M = 65521;
  uint64_t ad_a = 1, ad_b = 0;
  const uint64_t Lrow = static_cast<uint64_t>(L.row_bytes) + 1;
  for (int64_t r = 0; r < h; ++r) {
    ad_b = (ad_b + (Lrow % M) * ad_a + s2[static_cast<size_t>(r)] % M) % M;
    ad_a = (ad_a + s1[static_cast<size_t>(r)] % M) % M;
  }
  const uint32_t adler = static_cast<uint32_t>((ad_b << 16) | ad_a);

  // shift operators for the block lengths in use (full block, last block of a row)
  const int64_t last_bl = L.row_bytes - (L.nb - 1) * kBlockData;
  const uint32_t sh_full = xpow[static_cast<size_t>(kBlockData / 4)];
  const uint32_t sh_last = xpow[static_cast<size_t>(last_bl / 4)];
  struct Patch { int64_t at; uint8_t b[48]; int n; };
  std::vector<Patch> patches;
  uint32_t reg = 0xFFFFFFFFu;
  auto feed = [&](const uint8_t* p, int n) { for (int i = 0; i < n; ++i) reg = crc_byte(T, reg, p[i]); };
  size_t next_chunk = 0;
  int64_t chunk_len_at = 0, chunk_data_start = 0;
  int64_t pos = 0;
  for (int64_t r = 0; r < h; ++r) {
    const int64_t tab = L.row_tab[static_cast<size_t>(r)];
    const int64_t pix = tab & ~15ll; const int k = static_cast<int>(tab & 15);
    const int lead = 5 * k + 6;
    if (next_chunk < L.chunk_first_row.size() && L.chunk_first_row[next_chunk] == r) {
      Patch pt; std::memset(&pt, 0, sizeof pt);
      int n = 0;
      if (r == 0) {
        static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
        std::memcpy(pt.b, sig, 8); n = 8;
        put32(pt.b + n, 13); std::memcpy(pt.b + n + 4, "IHDR", 4);
        put32(pt.b + n + 8, static_cast<uint32_t>(w)); put32(pt.b + n + 12, static_cast<uint32_t>(h));
        pt.b[n + 16] = 8; pt.b[n + 17] = 6; pt.b[n + 18] = 0; pt.b[n + 19] = 0; pt.b[n + 20] = 0;
        uint32_t c = 0xFFFFFFFFu;
        for (int i = 4; i < 21; ++i) c = crc_byte(T, c, pt.b[n + i]);
        put32(pt.b + n + 21, c ^ 0xFFFFFFFFu);
        n += 25;
      } else {                      // close the previous IDAT: its CRC, and its length field
        put32(pt.b, reg ^ 0xFFFFFFFFu); n = 4;
        Patch lp; std::memset(&lp, 0, sizeof lp);
        lp.at = chunk_len_at; lp.n = 4; put32(lp.b, static_cast<uint32_t>(pos - chunk_data_start));
        patches.push_back(lp);
      }
      pt.at = pos;
      chunk_len_at = pos + n;                       // length is patched when the chunk closes
      std::memcpy(pt.b + n + 4, "IDAT", 4);
      reg = 0xFFFFFFFFu;
      feed(pt.b + n + 4, 4);
      n += 8;
      chunk_data_start = pos + n;
      if (r == 0) { pt.b[n] = 0x78; pt.b[n + 1] = 0x01; feed(pt.b + n, 2); n += 2; }      // zlib header: deflate, 32 K window, no dict
      pt.n = n;
      patches.push_back(pt);
      ++next_chunk;
    }
    // framing bytes of the row (the kernel wrote the same bytes into the file)
    uint8_t lead_b[96];
    for (int i = 0; i < 5 * k; ++i) lead_b[i] = (i % 5 >= 3) ? 0xFF : 0x00;
    {
      const int64_t bl = std::min(L.row_bytes, kBlockData);
      const uint32_t len = static_cast<uint32_t>(bl + 1);
      uint8_t* q = lead_b + 5 * k;
      q[0] = (r == h - 1 && L.nb == 1) ? 1 : 0; q[1] = len & 0xFF; q[2] = (len >> 8) & 0xFF; q[3] = (~len) & 0xFF; q[4] = ((~len) >> 8) & 0xFF; q[5] = 0;
    }
    feed(lead_b, lead);
    for (int b = 0; b < L.nb; ++b) {
      if (b > 0) {
        uint8_t mid[80];
        for (int i = 0; i < 75; ++i) mid[i] = (i % 5 >= 3) ? 0xFF : 0x00;
        const int64_t left = L.row_bytes - b * kBlockData;
        const uint32_t len = static_cast<uint32_t>(std::min(left, kBlockData));
        mid[75] = (r == h - 1 && b == L.nb - 1) ? 1 : 0; mid[76] = len & 0xFF; mid[77] = (len >> 8) & 0xFF; mid[78] = (~len) & 0xFF; mid[79] = ((~len) >> 8) & 0xFF;
        feed(mid, 80);
      }
      reg = gf_mul(b == L.nb - 1 ? sh_last : sh_full, reg) ^ crc[static_cast<size_t>(r) * L.nb + b];
    }
    pos = pix + L.row_bytes + 80ll * (L.nb - 1);
  }
  // trailer: adler32, close the last IDAT, IEND
  {
    Patch pt; std::memset(&pt, 0, sizeof pt);
    pt.at = pos;
    put32(pt.b, adler); feed(pt.b, 4);
    put32(pt.b + 4, reg ^ 0xFFFFFFFFu);
    put32(pt.b + 8, 0); std::memcpy(pt.b + 12, "IEND", 4);
    uint32_t c = 0xFFFFFFFFu;
    for (int i = 12; i < 16; ++i) c = crc_byte(T, c, pt.b[i]);
    put32(pt.b + 16, c ^ 0xFFFFFFFFu);
    pt.n = 20;
    patches.push_back(pt);
    Patch lp; std::memset(&lp, 0, sizeof lp);
    lp.at = chunk_len_at; lp.n = 4; put32(lp.b, static_cast<uint32_t>(pos + 4 - chunk_data_start));
    patches.push_back(lp);
  }
  for (const Patch& pt : patches)
    PNG_HIP(hipMemcpyAsync(static_cast<uint8_t*>(out) + pt.at, pt.b, static_cast<size_t>(pt.n), hipMemcpyHostToDevice, stream));
  PNG_HIP(hipStreamSynchronize(stream));
#undef PNG_HIP
  *out_len = L.total;
  return IST_OK;
}

}  // extern "C"
