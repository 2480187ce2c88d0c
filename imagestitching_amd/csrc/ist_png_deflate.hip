// ist_png_deflate.hip — the COMPRESSING PNG export: Paeth filter + run-length matches + dynamic Huffman, on the GPU.
//
// Reference anchor: the export step of onStitch — wx.canvasToTempFilePath({fileType:'png', quality:1})
// (utils/canvas.js:205-242, pages/index/index.js:1577-1579).  The reference's encoder is the WeChat client; a PNG that
// decodes to the same pixels is the same result, so the bit stream is designed for the hardware:
//
//   * the filtered stream (per row: filter byte 4 = Paeth, then the Paeth residuals of the RGBA bytes) is cut into
//     CHUNKS of at most 16 KiB — whole rows when a row fits, otherwise pieces of one row.  One 256-thread workgroup
//     compresses one chunk into its own deflate block, entirely in LDS, independently of every other chunk.
//   * tokens: thread t owns bytes [64 t, 64 t + 64) of the chunk.  A run of equal bytes becomes one literal + matches
//     of distance 1 (what zlib calls Z_RLE, its recommended strategy for PNG); flat areas — gaps, margins, screenshots
//     — filter to zeros and collapse to a few bits per 64 bytes.  Runs do not cross a thread's span, so no thread
//     needs its neighbour's state.
//   * code: one dynamic Huffman code per chunk over the 286 literal/length symbols, built from the chunk's histogram
//     (bitonic sort + two-queue merge + the zlib length-limit fix-up); the code-length alphabet uses a fixed 4-bit code
//     so no third code has to be built.  Every thread re-walks its span twice more: once to count its bits (block-wide
//     exclusive scan gives its bit offset), once to emit them (LDS atomicOr).
//   * a chunk whose Huffman form is not smaller than its bytes is written as a stored block (random data stays 1:1).
//   * every chunk ends byte-aligned (empty stored block = zlib's sync flush) and is padded with further empty stored
//     blocks (5 bytes each, coprime with 16) to a multiple of 16 bytes, so chunks concatenate with 16-byte copies.
//   * each workgroup also returns the Adler-32 partial sums of its filtered bytes and the raw CRC-32 of its output
//     bytes; the host combines them (O(chunks)), lays the chunks out behind one another (several IDAT chunks if the
//     stream exceeds the IDAT limit), a second kernel copies them into place, and ~100 header/trailer bytes are patched.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#include "ist_crc.h"
#include "ist_host.h"
#include "ist_internal.h"

namespace ist {

namespace {

constexpr int kCopyGrid = 0;             // workgroups of the device-to-host copy kernel (0: the runtime's copy); IST_PNG_COPY_GRID overrides in tuning mode
constexpr int CH = 16384;              // most filtered-stream bytes in one chunk
constexpr int SPAN = 64;               // bytes per thread
constexpr int SLOT = CH + 128;         // bytes per chunk in the scratch area (stored form + framing + alignment pads)
constexpr int NSYM = 286;              // literal/length symbols
constexpr int HDR_BITS = 3 + 5 + 5 + 4 + 19 * 3 + (NSYM + 1) * 4;      // block header with every code length sent in 4 bits
constexpr int PADDED = CH + CH / 16;   // LDS bytes of the filtered chunk: 4 pad bytes after every 64 (bank spread)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

struct DeflArgs {
  const uint8_t* canvas; size_t pitch; int64_t w, h;
  uint8_t* slots;                // n_chunks * SLOT
  uint32_t* len16;               // per chunk: bytes written / 16
  uint32_t* crc;                 // per chunk: raw CRC (register from 0) of the bytes written
  uint32_t* ad_a; uint32_t* ad_b; uint32_t* ad_n;    // per chunk: sum of bytes, index-weighted sum mod 65521, byte count
  const uint32_t* tables;        // 4 x 256 CRC slicing tables
  const uint32_t* xpow16;        // x^(128 i) mod P: shift of a CRC register over 16 i zero bytes
  int32_t rows_per_chunk;        // > 0: a chunk is this many whole rows; 0: a chunk is a piece of one row
  int32_t pieces_per_row, piece_px;
  int64_t chunk0;                // first chunk of this launch (the canvas is encoded in slabs of chunks)
  unsigned long long* dbg;       // IST_PNG_PHASES (tuning): 8 clock samples per chunk
};

__device__ __forceinline__ int padpos(int p) { return p + ((p >> 6) << 2); }

// Paeth residuals of one RGBA pixel (PNG spec 9.4): per channel the neighbour - left, up or upper left - closest to left +
// up - upper left, ties in that order.  Two channels at a time as packed 16-bit lanes (v_pk_sub_i16 / v_pk_max_i16 /
// v_pk_ashrrev_i16): pa = |up - ul|, pb = |left - ul|, pc = |(up - ul) + (left - ul)|; a comparison is the sign of a packed
// difference spread over its lane, a selection is a bit-field insert.  52 operations per pixel instead of ~85 byte-wise ones; the
// kernel is instruction-issue-bound (87 % of its issue slots busy, profiles/r03_counters_png.txt).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t paeth_pred2(uint32_t l, uint32_t u, uint32_t ul) {      // two channels, 16-bit lanes holding 0..255
  const s16x2 zero = {0, 0};
  const s16x2 d1 = as_s16x2(u) - as_s16x2(ul), d2 = as_s16x2(l) - as_s16x2(ul), d3 = d1 + d2;
  const s16x2 pa = __builtin_elementwise_max(d1, zero - d1), pb = __builtin_elementwise_max(d2, zero - d2), pc = __builtin_elementwise_max(d3, zero - d3);
  const uint32_t a_gt_b = as_u32((pb - pa) >> 15), a_gt_c = as_u32((pc - pa) >> 15), b_gt_c = as_u32((pc - pb) >> 15);      // all ones where greater
  const uint32_t not_a = a_gt_b | a_gt_c;
  const uint32_t u_or_ul = (ul & b_gt_c) | (u & ~b_gt_c);
  return (u_or_ul & not_a) | (l & ~not_a);
}
__device__ __forceinline__ uint32_t paeth4(uint32_t cur, uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t m = 0x00FF00FFu;
  const uint32_t pe = paeth_pred2(a & m, b & m, c & m), po = paeth_pred2((a >> 8) & m, (b >> 8) & m, (c >> 8) & m);
  const uint32_t pred = pe | (po << 8);
  // cur - pred in every byte: borrow-free subtraction of the low seven bits, the top bits by xor
  return ((cur | 0x80808080u) - (pred & 0x7F7F7F7Fu)) ^ ((cur ^ ~pred) & 0x80808080u);
}

// deflate length symbol for a match of l bytes (3 <= l <= 66): symbol, extra-bit count, extra-bit value
__device__ __forceinline__ void len_code(int l, int* sym, int* eb, int* ev) {
  const int m = l - 3;
  if (m < 8) { *sym = 257 + m; *eb = 0; *ev = 0; return; }
  const int e = (31 - __builtin_clz(m)) - 2;
  *sym = 257 + 4 * (e + 1) + ((m >> e) & 3); *eb = e; *ev = m & ((1 << e) - 1);
}

struct BitWriter {
  uint32_t* w; unsigned long long acc; int n; int wp;
  __device__ void init(uint32_t* words, int bitpos) { w = words; wp = bitpos >> 5; n = bitpos & 31; acc = 0; }
  __device__ void put(uint32_t v, int bits) {
    acc |= static_cast<unsigned long long>(v) << n; n += bits;
    if (n >= 32) { atomicOr(&w[wp++], static_cast<uint32_t>(acc)); acc >>= 32; n -= 32; }
  }
  __device__ void flush() { if (n > 0 && acc) atomicOr(&w[wp], static_cast<uint32_t>(acc)); }
};

__device__ __forceinline__ void or_bits(uint32_t* w, int bitpos, uint32_t v, int bits) {
  const int wp = bitpos >> 5, sh = bitpos & 31;
  if (v) {
    atomicOr(&w[wp], v << sh);
    if (sh + bits > 32) atomicOr(&w[wp + 1], v >> (32 - sh));
  }
}

__device__ __forceinline__ uint32_t rev_bits(uint32_t code, int len) { return __brev(code) >> (32 - len); }

// The span lives in REGISTERS (16 dwords): the LDS image of the filtered chunk is dead once every thread holds its span, so
// the output bit buffer takes its place (one 17 KiB buffer instead of two: 6 workgroups per CU instead of 4 - the kernel's
// clock is LDS latency, so residency is throughput).  The walk's dword loop has a wave-uniform trip count, which makes the
// register pick a scalar jump.
__device__ __forceinline__ uint32_t span_word(const uint32_t (&sp)[16], int d) {
  switch (d) {
    case 0: return sp[0]; case 1: return sp[1]; case 2: return sp[2]; case 3: return sp[3];
    case 4: return sp[4]; case 5: return sp[5]; case 6: return sp[6]; case 7: return sp[7];
    case 8: return sp[8]; case 9: return sp[9]; case 10: return sp[10]; case 11: return sp[11];
    case 12: return sp[12]; case 13: return sp[13]; case 14: return sp[14]; default: return sp[15];
  }
}

// Which bytes of a span are literals and where its matches start, as 64-bit masks - computed once per walk, so that the walk
// itself carries no run state from byte to byte (the stateful walk cost ~23 instructions per byte in each of the three
// passes; this one ~6).  eq bit i: byte i equals byte i-1.  A run of R >= 4 equal bytes is its first byte as a literal + ONE
// match of R - 1 bytes at distance 1, shorter runs are literals: `covered` = the bytes inside matches = every eq position that
// belongs to three consecutive eq bits; a match starts where a covered stretch starts and is as long as the stretch.
struct SpanMasks { unsigned long long lit, mstart, covered; };
__device__ __forceinline__ SpanMasks span_masks(const uint32_t (&sp)[16], int n) {
  unsigned long long eq = 0;
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    const uint32_t w = sp[d];
    const uint32_t prev = d ? (sp[d ? d - 1 : 0] >> 24) : (~w & 0xFFu);            // (byte 0 has no predecessor: never equal)
    const uint32_t x = w ^ ((w << 8) | prev);                                      // a zero byte where a byte equals the one before it
    const uint32_t nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;     // 0x80 in every NON-zero byte (exact)
    const uint32_t z = (nz ^ 0x80808080u) >> 7;                                    // bit 0 / 8 / 16 / 24: the byte is zero
    const uint32_t nib = ((z * 0x00204081u) >> 21) & 15u;                          // those four bits side by side
    eq |= static_cast<unsigned long long>(nib) << (4 * d);
  }
  const unsigned long long valid = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
  eq &= valid & ~1ull;
  const unsigned long long t = eq & (eq >> 1) & (eq >> 2);                         // bit i: bytes i-1 .. i+2 are equal
  SpanMasks m;
  m.covered = (t | (t << 1) | (t << 2)) & valid;
  m.lit = valid & ~m.covered;
  m.mstart = m.covered & ~(m.covered << 1);
  return m;
}

// One pass of a thread over its span.  PASS 0: histogram; 1: count bits; 2: emit.  The span is read as 16 dwords (the
// padded layout keeps every span 4-byte aligned); the dword loop has a wave-uniform trip count, which makes the register
// pick a scalar jump.
template <int PASS>
__device__ __forceinline__ int walk_span(const uint32_t (&sp)[16], int n, const SpanMasks& M, uint32_t* hist, const uint32_t* clen, const uint32_t* code, BitWriter* bw) {
  int bits = 0;
  if (PASS != 2) {                                   // matches in any order: a loop over the (few) places where one starts
    unsigned long long ms = M.mstart;
    while (ms) {
      const int i = __ffsll(static_cast<long long>(ms)) - 1;
      ms &= ms - 1ull;
      const unsigned long long rest = ~(M.covered >> i);
      const int mlen = rest ? __ffsll(static_cast<long long>(rest)) - 1 : 64 - i;       // 3 .. 63 covered bytes in a row
      int msym, meb, mev;
      len_code(mlen, &msym, &meb, &mev);
      if (PASS == 0) atomicAdd(&hist[msym], 1u);
      else bits += static_cast<int>(clen[msym]) + meb + 1;
    }
  }
#pragma unroll 1
  for (int d = 0; d < 16; ++d) {
    if (4 * d >= n) break;                           // (wave-uniform only in full chunks; a scalar branch there)
    const uint32_t w = span_word(sp, d);
    const uint32_t lit4 = static_cast<uint32_t>(M.lit >> (4 * d)) & 15u;
    if (PASS != 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t b = (w >> (8 * k)) & 255u;
        if (lit4 & (1u << k)) {
          if (PASS == 0) atomicAdd(&hist[b], 1u);
          else bits += static_cast<int>(clen[b]);
        }
      }
    } else {
      // emission, two bytes per put: a byte that is no literal contributes zero bits (a select, not a branch), so the pair's
      // codes go out as ONE word of at most 30 bits; the match that starts at one of the two positions - its bytes are no
      // literals, so the order inside the pair does not matter - follows in a rarely taken branch
      const uint32_t ms4 = static_cast<uint32_t>(M.mstart >> (4 * d)) & 15u;
#pragma unroll
      for (int k = 0; k < 4; k += 2) {
        const uint32_t p0 = code[(w >> (8 * k)) & 255u], p1 = code[(w >> (8 * k + 8)) & 255u];      // (code[] carries the length in its upper half)
        const uint32_t c0 = (lit4 & (1u << k)) ? p0 : 0u, c1 = (lit4 & (2u << k)) ? p1 : 0u;
        const uint32_t l0 = c0 >> 16;
        bw->put((c0 & 0xFFFFu) | ((c1 & 0xFFFFu) << l0), static_cast<int>(l0 + (c1 >> 16)));
        if (ms4 & (3u << k)) {                       // the match behind a run's first byte (distance 1: one zero bit)
          const int i = 4 * d + k + ((ms4 >> k) & 1u ? 0 : 1);
          const unsigned long long rest = ~(M.covered >> i);
          const int mlen = rest ? __ffsll(static_cast<long long>(rest)) - 1 : 64 - i;
          int msym, meb, mev;
          len_code(mlen, &msym, &meb, &mev);
          const uint32_t pm = code[msym];
          bw->put(pm & 0xFFFFu, static_cast<int>(pm >> 16));
          if (meb) bw->put(static_cast<uint32_t>(mev), meb);
          bw->put(0u, 1);
        }
      }
    }
  }
  return bits;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7))) void ist_png_deflate_kernel(const DeflArgs P) {
  // ONE buffer, two lives: the filtered chunk (padded layout) until every thread has taken its span into registers, then the
  // Huffman scratch and the output bit stream
  static_assert(PADDED >= SLOT, "the output buffer aliases the filtered chunk");
  __shared__ __attribute__((aligned(16))) uint32_t buf[PADDED / 4];
  uint8_t* const filt = reinterpret_cast<uint8_t*>(buf);
  uint32_t* const outw = buf;
  // one LDS area, two lives: histogram / code lengths / codes while the block is built, then the CRC tables
  __shared__ __attribute__((aligned(16))) uint32_t area[1024];      // (>= 3 * 288)
  uint32_t* const hist = area; uint32_t* const clen = area + 288; uint32_t* const code = area + 576;
  uint32_t (*const T)[256] = reinterpret_cast<uint32_t (*)[256]>(area);
  __shared__ uint32_t wsum[8];
  __shared__ int s_out_len, s_skip;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t chunk = P.chunk0 + blockIdx.x;

  // ---- which pixels: rows [y0, y0 + nrows) x columns [x0, x0 + npr); the filter byte belongs to the piece with x0 == 0
  int64_t y0; int nrows, x0, npr;
  if (P.rows_per_chunk > 0) {
    y0 = chunk * P.rows_per_chunk; nrows = static_cast<int>(min(static_cast<int64_t>(P.rows_per_chunk), P.h - y0)); x0 = 0; npr = static_cast<int>(P.w);
  } else {
    y0 = chunk / P.pieces_per_row; nrows = 1;
    const int piece = static_cast<int>(chunk - y0 * P.pieces_per_row);
    x0 = piece * P.piece_px; npr = static_cast<int>(min(static_cast<int64_t>(P.piece_px), P.w - x0));
  }
  const int hasf = x0 == 0 ? 1 : 0;
  const int rowlen = hasf + 4 * npr;
  const int len = rowlen * nrows;                // <= CH by construction

  for (int i = tid; i < 288; i += 256) { hist[i] = 0; clen[i] = 0; code[i] = 0; }
  if (P.dbg && tid == 0) P.dbg[chunk * 8 + 0] = wall_clock64();
  // ---- A. load + Paeth filter into LDS
  const int npix = npr * nrows;
  // four pixels per thread and round, all sixteen loads issued before the first filter: the one-pixel-per-iteration form
  // had four loads in flight per thread and spent 15 us of a chunk's 96 waiting for them (IST_PNG_PHASES, measured)
  for (int q0 = tid; q0 < npix; q0 += 1024) {
    uint32_t cur[4], a[4], b[4], c[4]; int pos[4]; bool first[4], on[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + 256 * u;
      on[u] = q < npix;
      cur[u] = a[u] = b[u] = c[u] = 0u; pos[u] = 0; first[u] = false;
      if (on[u]) {
        const int r = nrows == 1 ? 0 : q / npr, xx = q - r * npr;      // (one row per chunk - every image wider than 2047 px - needs no division)
        const int64_t y = y0 + r; const int x = x0 + xx;
        const uint8_t* row = P.canvas + static_cast<size_t>(y) * P.pitch;
        cur[u] = *reinterpret_cast<const uint32_t*>(row + 4 * static_cast<size_t>(x));
        if (x > 0) a[u] = *reinterpret_cast<const uint32_t*>(row + 4 * static_cast<size_t>(x - 1));
        if (y > 0) {
          b[u] = *reinterpret_cast<const uint32_t*>(row - P.pitch + 4 * static_cast<size_t>(x));
          if (x > 0) c[u] = *reinterpret_cast<const uint32_t*>(row - P.pitch + 4 * static_cast<size_t>(x - 1));
        }
        pos[u] = r * rowlen + hasf + 4 * xx;
        first[u] = hasf && xx == 0;
        if (first[u]) filt[padpos(r * rowlen)] = 4;              // filter type of the row
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!on[u]) continue;
      const uint32_t fv = paeth4(cur[u], a[u], b[u], c[u]);
#pragma unroll
      for (int k = 0; k < 4; ++k) filt[padpos(pos[u] + k)] = static_cast<uint8_t>(fv >> (8 * k));
    }
  }
  __syncthreads();

  if (P.dbg && tid == 0) P.dbg[chunk * 8 + 1] = wall_clock64();
  // ---- B. histogram (+ the end-of-block symbol), Adler partials
  const int base = tid * SPAN;
  const int n = max(0, min(SPAN, len - base));
  uint32_t sp[16];
  {
    const u32x4* mine = reinterpret_cast<const u32x4*>(filt + tid * (SPAN + 4));      // (68-byte pitch: 4-byte aligned)
    const uint32_t* mw = reinterpret_cast<const uint32_t*>(mine);
#pragma unroll
    for (int d = 0; d < 16; ++d) sp[d] = mw[d];
  }
  __syncthreads();                                   // every span is in registers: the buffer is free
  walk_span<0>(sp, n, span_masks(sp, n), hist, nullptr, nullptr, nullptr);
  if (tid == 0) atomicAdd(&hist[256], 1u);
  uint32_t a1 = 0, a2 = 0;
#pragma unroll
  for (int d = 0; d < 16; ++d)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = 4 * d + k;
      const uint32_t b = i < n ? (sp[d] >> (8 * k)) & 255u : 0u;
      a1 += b; a2 += static_cast<uint32_t>(len - (base + i)) * b;
    }
  a2 %= 65521u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off); }
  if (lane == 0) { wsum[wave] = a1; wsum[4 + wave] = a2; }
  __syncthreads();
  if (tid == 0) {
    P.ad_a[chunk] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    P.ad_b[chunk] = (wsum[4] + wsum[5] + wsum[6] + wsum[7]) % 65521u;
    P.ad_n[chunk] = static_cast<uint32_t>(len);
  }

  if (P.dbg && tid == 0) P.dbg[chunk * 8 + 2] = wall_clock64();
  // ---- C. code lengths and codes: ONE WAVE, in registers.  Not Huffman's algorithm: Shannon lengths (the smallest l with
  // count * 2^l >= total, <= 15 for a 16 KiB chunk; their Kraft sum is <= 1) and then the slack handed back - symbols
  // are shortened by one bit, shortest codes first, as long as the Kraft sum allows, round after round until the code is
  // COMPLETE (inflate rejects an incomplete literal/length code).  Every shortening costs a multiple of the smallest unit
  // left, so the rounds end with the slack at exactly zero (<= 12 rounds over 3000 random histograms; photo chunks need
  // 2-4).  Which symbols of a length class go first is decided by symbol index, so the file is the same on every run.
  // Against Huffman's lengths the token bits grow by 0.4 % (photo chunks) to 1.2 % (noise) - tools/sim_code_lengths.py -
  // and the sort (36-45 barrier steps), the serial two-queue merge (n - 1 dependent LDS round trips on one thread), the
  // depth walk and the 286-long rank loops are gone: 24 us of a chunk's ~90 (IST_PNG_PHASES) become a few.
  // Lane j holds symbols j, 64 + j, ... 256 + j.  A chunk whose code does not complete in 16 rounds is stored.
  const int lead = chunk == 0 ? 2 : 0;               // the zlib header travels with the first chunk
  __syncthreads();                                   // thread 0 has read the Adler partials out of wsum[]; hist[] is final
  if (tid == 0) s_skip = 0;
  if (wave == 0) {
    uint32_t cnt[5]; int ln[5];
    uint32_t tot = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) { const int sy = 64 * q + lane; cnt[q] = sy < NSYM ? hist[sy] : 0u; tot += cnt[q]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
    int kraft = 0, cls[16];                          // cls[l]: symbols of length l (wave-uniform)
#pragma unroll
    for (int l = 0; l < 16; ++l) cls[l] = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      ln[q] = 0;
      if (cnt[q]) {
        const uint32_t r = (tot + cnt[q] - 1) / cnt[q];                       // ceil(total / count), 1 .. 16385
        int l = r <= 1u ? 0 : 32 - __clz(static_cast<int>(r - 1u));           // ceil(log2(r))
        l = l < 1 ? 1 : (l > 15 ? 15 : l);
        ln[q] = l;
        kraft += 1 << (15 - l);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kraft += __shfl_xor(kraft, off);
#pragma unroll
    for (int l = 1; l < 16; ++l) {
      int c = 0;
#pragma unroll
      for (int q = 0; q < 5; ++q) c += __popcll(__ballot(ln[q] == l));
      cls[l] = c;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    // rank of this lane's five symbols among the symbols of length l, in symbol order (q-major, then lane)
    auto ranks_in = [&](int l, int (&rk)[5]) {
      int acc = 0;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const unsigned long long m = __ballot(ln[q] == l);
        rk[q] = acc + __popcll(m & below);
        acc += __popcll(m);
      }
    };
    int slack = 32768 - kraft;                       // >= 0 (Shannon lengths); 0 = complete
    for (int round = 0; round < 16 && slack > 0; ++round) {
#pragma unroll
      for (int l = 2; l < 16; ++l) {                 // (ascending: a symbol moved to l - 1 is not looked at again this round)
        const int cost = 1 << (15 - l);
        const int take = min(cls[l], slack / cost);
        if (take > 0) {                              // (wave-uniform)
          int rk[5];
          ranks_in(l, rk);
#pragma unroll
          for (int q = 0; q < 5; ++q) if (ln[q] == l && rk[q] < take) ln[q] = l - 1;
          cls[l] -= take; cls[l - 1] += take; slack -= take * cost;
        }
      }
    }
    if (slack != 0 || tot == 0) { if (lane == 0) s_skip = 1; }
    else {
      // canonical codes (RFC 1951 3.2.2): first code of every length, then symbol order inside a length; stored bit-reversed
      int first[16]; int cd = 0;
      first[0] = 0;
#pragma unroll
      for (int l = 1; l < 16; ++l) { cd = (cd + cls[l - 1]) << 1; first[l] = cd; }
      int cdq[5] = {0, 0, 0, 0, 0};
#pragma unroll
      for (int l = 1; l < 16; ++l) {
        if (cls[l] == 0) continue;                   // (wave-uniform)
        int rk[5];
        ranks_in(l, rk);
#pragma unroll
        for (int q = 0; q < 5; ++q) if (ln[q] == l) cdq[q] = first[l] + rk[q];
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int sy = 64 * q + lane;
        if (sy < NSYM) {
          clen[sy] = static_cast<uint32_t>(ln[q]);
          code[sy] = ln[q] ? (rev_bits(static_cast<uint32_t>(cdq[q]), ln[q]) | (static_cast<uint32_t>(ln[q]) << 16)) : 0u;
        }
      }
    }
  }
  __syncthreads();
  const bool skip = s_skip != 0;

  if (P.dbg && tid == 0) P.dbg[chunk * 8 + 3] = wall_clock64();
  // ---- D. bits per thread, exclusive scan, choice between the Huffman and the stored form
  const SpanMasks masks = span_masks(sp, n);         // (for the bit count and the emission: kept across the scan between them)
  const int mybits = skip ? 0 : walk_span<1>(sp, n, masks, nullptr, clen, nullptr, nullptr);
  int incl = mybits;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
  __syncthreads();                                   // outw scratch (weights) no longer needed by thread 0
  if (lane == 63) wsum[wave] = static_cast<uint32_t>(incl);
  __syncthreads();
  int wave_base = 0;
  for (int k = 0; k < wave; ++k) wave_base += static_cast<int>(wsum[k]);
  const int total_tok_bits = static_cast<int>(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  const int my_start = lead * 8 + HDR_BITS + wave_base + incl - mybits;
  const int end_bit = lead * 8 + HDR_BITS + total_tok_bits + static_cast<int>(clen[256]);
  const int huff_bytes = (end_bit + 3 + 7) / 8 + 4;  // + the empty stored block that re-aligns to a byte
  const bool stored = skip || huff_bytes >= lead + 5 + len;
  for (int i = tid; i < SLOT / 4; i += 256) outw[i] = 0;
  __syncthreads();
  if (P.dbg && tid == 0) P.dbg[chunk * 8 + 4] = wall_clock64();
  uint8_t* outb = reinterpret_cast<uint8_t*>(outw);
  int body_end;                                       // bytes before the alignment pads
  if (!stored) {
    // header: BFINAL 0, BTYPE 2, HLIT 29, HDIST 0, HCLEN 15, the 19 code-length-code lengths, 286 + 1 code lengths.
    // Code-length alphabet: symbols 0..15 all 4 bits long (a complete code: canonical code of symbol s = s), 16-18 unused.
    const int hb = lead * 8;
    if (tid == 0) {
      if (lead) { outb[0] = 0x78; outb[1] = 0x01; }
      or_bits(outw, hb, 4u | (29u << 3) | (0u << 8) | (15u << 13), 17);
    }
    if (tid < 19) or_bits(outw, hb + 17 + 3 * tid, tid < 3 ? 0u : 4u, 3);       // order 16,17,18,0,8,7,...: first three are unused
    for (int s = tid; s < NSYM + 1; s += 256) {
      const uint32_t l = s < NSYM ? clen[s] : 1u;                                // the one distance code: 1 bit
      or_bits(outw, hb + 17 + 57 + 4 * s, rev_bits(l, 4), 4);
    }
    BitWriter bw; bw.init(outw, my_start);
    walk_span<2>(sp, n, masks, nullptr, clen, code, &bw);
    bw.flush();
    if (tid == 0) or_bits(outw, end_bit - static_cast<int>(clen[256]), code[256] & 0xFFFFu, static_cast<int>(clen[256]));
    body_end = (end_bit + 3 + 7) / 8;                 // 3 zero bits: BFINAL 0, BTYPE 00; then to the byte boundary
    __syncthreads();
    if (tid == 0) { outb[body_end + 2] = 0xFF; outb[body_end + 3] = 0xFF; }      // LEN 0000, NLEN FFFF
    body_end += 4;
  } else {
    if (tid == 0) {
      if (lead) { outb[0] = 0x78; outb[1] = 0x01; }
      uint8_t* q = outb + lead;
      q[0] = 0; q[1] = len & 0xFF; q[2] = (len >> 8) & 0xFF; q[3] = (~len) & 0xFF; q[4] = ((~len) >> 8) & 0xFF;
    }
#pragma unroll
    for (int d = 0; d < 16; ++d)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * d + k < n) outb[lead + 5 + base + 4 * d + k] = static_cast<uint8_t>(sp[d] >> (8 * k));
    body_end = lead + 5 + len;
  }
  __syncthreads();
  if (tid == 0) {
    int total = body_end;
    while (total & 15) { outb[total + 3] = 0xFF; outb[total + 4] = 0xFF; total += 5; }   // empty stored blocks: 00 00 00 FF FF
    s_out_len = total;
    P.len16[chunk] = static_cast<uint32_t>(total >> 4);
  }
  __syncthreads();

  if (P.dbg && tid == 0) P.dbg[chunk * 8 + 5] = wall_clock64();
  // ---- E. write the chunk to its slot; raw CRC of its bytes (each 16-byte unit shifted to the end of the chunk)
  for (int i = tid; i < 1024; i += 256) area[i] = P.tables[i];          // the code tables are dead: the area now holds the CRC tables
  __syncthreads();
  const int out_len = s_out_len;
  uint8_t* slot = P.slots + static_cast<size_t>(chunk) * SLOT;
  uint32_t crc = 0;
  for (int u = tid; u < (out_len >> 4); u += 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(outw + 4 * u);
    *reinterpret_cast<u32x4*>(slot + 16 * static_cast<size_t>(u)) = v;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t c = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      c ^= w[d];
      c = T[3][c & 0xFF] ^ T[2][(c >> 8) & 0xFF] ^ T[1][(c >> 16) & 0xFF] ^ T[0][c >> 24];
    }
    crc ^= gf_mul(P.xpow16[(out_len >> 4) - 1 - u], c);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) crc ^= __shfl_xor(crc, off);
  __syncthreads();
  if (lane == 0) wsum[wave] = crc;
  __syncthreads();
  if (tid == 0) P.crc[chunk] = wsum[0] ^ wsum[1] ^ wsum[2] ^ wsum[3];
  if (P.dbg && tid == 0) P.dbg[chunk * 8 + 6] = wall_clock64();
}

struct GatherArgs { const uint8_t* slots; uint8_t* out; const int64_t* dst; const uint32_t* len16; int64_t chunk0; };

// chunk j: len16[j] 16-byte units from its slot to file offset dst[j] (4-byte aligned)
__global__ __launch_bounds__(256) void ist_png_gather_kernel(const GatherArgs G) {
  const int64_t j = G.chunk0 + blockIdx.x;
  const uint8_t* s = G.slots + static_cast<size_t>(j) * SLOT;
  uint8_t* d = G.out + G.dst[j];
  const int units = static_cast<int>(G.len16[j]);
  for (int u = threadIdx.x; u < units; u += 256)
    *reinterpret_cast<u32x4_a4*>(d + 16 * static_cast<size_t>(u)) = *reinterpret_cast<const u32x4*>(s + 16 * static_cast<size_t>(u));
}

// device -> pinned host, 16-byte units, a FIXED small grid striding over the range: the slab's trip over PCIe as a kernel that
// holds a few dozen workgroup slots, instead of the runtime's device-to-host copy, which runs as a blit kernel of thousands of
// workgroups parked on PCIe writes beside the compressing kernel (PNG stage 3.5 ms with those copies, 2.4 ms without them -
// measured with IST_PNG_SKIP_D2H - against 2.6-2.75 ms of PCIe time)
__global__ __launch_bounds__(256) void ist_png_to_host_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, const int64_t units) {
  for (int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; u < units; u += static_cast<int64_t>(gridDim.x) * 256)
    dst[u] = __builtin_nontemporal_load(src + u);
}

void put32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = (v >> 16) & 0xFF; p[2] = (v >> 8) & 0xFF; p[3] = v & 0xFF; }

struct ChunkGrid { int64_t n_chunks; int rows_per_chunk, pieces_per_row, piece_px; };

ChunkGrid make_grid(int64_t w, int64_t h) {
  ChunkGrid g{0, 0, 0, 0};
  const int64_t R = 4 * w + 1;
  if (R <= CH) { g.rows_per_chunk = static_cast<int>(CH / R); g.n_chunks = (h + g.rows_per_chunk - 1) / g.rows_per_chunk; }
  else { g.piece_px = (CH - 1) / 4; g.pieces_per_row = static_cast<int>((w + g.piece_px - 1) / g.piece_px); g.n_chunks = h * g.pieces_per_row; }
  return g;
}

int64_t idat_limit() {
  const char* e = std::getenv("IST_PNG_IDAT_LIMIT");
  const int64_t v = e ? std::atoll(e) : 0;
  return v >= 4096 ? (v & ~15ll) : (1ll << 30);
}

constexpr int kDataStart = 64;     // signature 8 + IHDR 25 + tEXt 23 + IDAT length/type 8: the stream starts 16-byte aligned

}  // namespace

// upper bound of the compressed form: every chunk stored
int64_t png_deflate_bound(int64_t w, int64_t h) {
  const ChunkGrid g = make_grid(w, h);
  const int64_t data = g.n_chunks * SLOT + 16;
  return kDataStart + data + (data / idat_limit() + 2) * 12 + 64;
}

// host_out (optional, pinned, out_cap bytes) + aux: the file is ALSO delivered to host memory, slab by slab, on the aux
// stream while later slabs are still being compressed on `stream` — the PNG's trip over PCIe (2.6 ms for the 146 MB of a
// 439 MB photo canvas) then hides behind the encoder (3.3 ms) instead of following it.  Each slab is its own IDAT chunk.
int png_encode_device_deflate(ist_ctx* ctx, const void* canvas, size_t pitch, int64_t w, int64_t h, void* out, int64_t out_cap,
                              int64_t* out_len, void* stream_, uint8_t* host_out, void* aux_,
                              const std::function<int(int64_t, void*)>& need_rows, int64_t slab_rows_hint, void* stream2_) {
  const ChunkGrid g = make_grid(w, h);
  if (g.n_chunks > 2147483647ll) return fail(IST_E_OUTPUT_SIZE, "image too large for one PNG launch");
  if (png_deflate_bound(w, h) > out_cap) return fail(IST_E_INVALID, "PNG output buffer too small (see ist_png_bound)");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipStream_t aux = host_out ? static_cast<hipStream_t>(aux_) : stream;
  // slabs alternate between two streams (when the caller has a second one): consecutive kernels of ONE stream do not overlap,
  // so every slab paid its own tail - the last, partly filled round of workgroups - with the rest of the chip idle; on two
  // streams slab s+1 fills in as slab s drains (a slab is ~3 rounds of workgroups: measured 0.59 ms per 3024-chunk slab alone)
  hipStream_t stream2 = (host_out && stream2_) ? static_cast<hipStream_t>(stream2_) : stream;
  static CrcTables T;
  static std::vector<uint32_t> xpow;
  static std::once_flag once;
  std::call_once(once, []() {
    make_crc_tables(&T);
    xpow.resize(SLOT / 16 + 2);
    uint32_t reg = 0x80000000u;                      // the polynomial "1"
    for (size_t i = 0; i < xpow.size(); ++i) { xpow[i] = reg; for (int z = 0; z < 16; ++z) reg = crc_byte(T, reg, 0); }
  });
  const size_t n = static_cast<size_t>(g.n_chunks);
  auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  // per-chunk results, five arrays interleaved per SLAB so that a slab's results are one contiguous copy:
  // [len16 | crc | ad_a | ad_b | ad_n] x slab
  // 64 MiB of filtered stream per slab - or, when the producer renders the canvas band by band (need_rows), about one band
  // (never less than a quarter of the default: a slab is a launch, a layout pass and a copy), so that slab k is complete
  // when band k is and nothing of band k+1 is waited for
  size_t slab_chunks = 4096;
  if (slab_rows_hint > 0) {
    const int64_t rows_chunks = g.pieces_per_row > 0 ? slab_rows_hint * g.pieces_per_row : (slab_rows_hint + g.rows_per_chunk - 1) / std::max(1, g.rows_per_chunk);
    slab_chunks = static_cast<size_t>(std::min<int64_t>(4096, std::max<int64_t>(1024, rows_chunks)));
  }
  // slab s = chunks [slab_at[s], slab_at[s + 1]).  The FIRST slab is short: nothing crosses PCIe before it has been compressed,
  // laid out and gathered, so its size is the fill time of the pipeline (0.66 ms with a full first slab sharing the chip with
  // the second: measured); behind it the copies are back to back
  std::vector<size_t> slab_at{0};
  if (host_out) {
    const size_t first = std::min<size_t>(n, std::min<size_t>(slab_chunks, 768));
    for (size_t c = first; c < n; c += slab_chunks) slab_at.push_back(c);
  }
  slab_at.push_back(n);
  const size_t n_slabs = slab_at.size() - 1;
  const size_t per_slab = host_out ? slab_chunks : n;          // the largest slab (sizes the per-slab result arrays)
  // the last canvas row (exclusive) chunks [0, c_end) read
  auto rows_of = [&](size_t c_end) -> int64_t {
    const int64_t r = g.pieces_per_row > 0 ? (static_cast<int64_t>(c_end) + g.pieces_per_row - 1) / g.pieces_per_row : static_cast<int64_t>(c_end) * g.rows_per_chunk;
    return std::min<int64_t>(h, r);
  };
  const size_t o_T = 0, o_pow = o_T + up(sizeof T), o_slots = o_pow + up(4 * xpow.size()), total_scratch = o_slots + n * SLOT;
  // the context's grow-only scratch: a 439 MB canvas needs 443 MB of slots, and allocating + freeing that per call cost
  // more than the gather kernel (and a free synchronises the device)
  uint8_t* scratch = nullptr;
  {
    void* p = nullptr;
    const int rc = ctx_png_scratch(ctx, total_scratch, &p);
    if (rc) return rc;
    scratch = static_cast<uint8_t*>(p);
  }
#define PNG_HIP(e) do { const hipError_t e_ = (e); if (e_ != hipSuccess) return fail(IST_E_HIP, std::string(#e) + ": " + hipGetErrorString(e_)); } while (0)
  PNG_HIP(hipMemcpyAsync(scratch + o_T, &T, sizeof T, hipMemcpyHostToDevice, stream));
  PNG_HIP(hipMemcpyAsync(scratch + o_pow, xpow.data(), 4 * xpow.size(), hipMemcpyHostToDevice, stream));
  if (stream2 != stream) PNG_HIP(hipStreamSynchronize(stream));      // (both are blocking staged copies of static data: the tables are in place for either stream)
  // ---- every slab's compression goes out now, each followed by the copy of its per-chunk results and an event
  // (pinned: a device-to-host copy into pageable memory would block this thread until the slab is compressed, and the
  // slabs' launches would no longer run ahead of the host)
  struct Events { std::vector<hipEvent_t> ev; ~Events() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); } } evs;   // (a slab that was never launched has no event: destroying NULL would leave a sticky error for the next launch check)
  evs.ev.assign(n_slabs, nullptr);
  Events gathered;
  gathered.ev.assign(n_slabs, nullptr);

  struct Pinned { uint8_t* p; ~Pinned() { if (p) pool_give(p); } } res{static_cast<uint8_t*>(pool_take(20 * per_slab * n_slabs))};
  if (!res.p) return fail(IST_E_NOMEM, "out of pinned host memory for the PNG encoder");
  // file offset of every chunk: pinned host memory the gather kernel reads in place (a pageable source would cost one small
  // staged copy per slab on the aux stream; those, and the header patches, were ~0.5 ms of stream time per slab)
  struct PinnedDst { int64_t* p; ~PinnedDst() { if (p) pool_give(p); } } dstp{static_cast<int64_t*>(pool_take(8 * n))};
  if (!dstp.p) return fail(IST_E_NOMEM, "out of pinned host memory for the PNG encoder");
  unsigned long long* d_dbg = nullptr;
  // EVERY way out of this function below - the errors too - first waits for both streams: the kernels in flight write their
  // per-chunk results into `res` and read `dstp` in place, and a block given back to the pool while slab s+1 is still
  // compressing could be handed to another thread's call.  (Declared after the two blocks: destroyed before them.)
  struct Drain {
    hipStream_t a, b, c; unsigned long long** dbg; bool armed = true;
    ~Drain() {
      if (armed) { (void)hipStreamSynchronize(a); if (b != a) (void)hipStreamSynchronize(b); if (c != a && c != b) (void)hipStreamSynchronize(c); }
      if (*dbg) { dev_free(*dbg); *dbg = nullptr; }
    }
  } drain{stream, aux, stream2, &d_dbg};
  static const bool phases = tuning_mode() && std::getenv("IST_PNG_PHASES") != nullptr;      // (IST_TUNING=1 processes only)
  if (phases && dev_malloc(reinterpret_cast<void**>(&d_dbg), 64 * n) != hipSuccess) d_dbg = nullptr;
  // test knob (IST_TUNING=1 IST_PNG_FAIL_AT=<slab>): fail the layout of that slab the way a damaged result would, so that the
  // error path above is exercised with kernels in flight
  static const long fail_at = (tuning_mode() && std::getenv("IST_PNG_FAIL_AT")) ? std::atol(std::getenv("IST_PNG_FAIL_AT")) : -1;
  auto compress = [&](size_t s) -> int {
    const size_t c0 = slab_at[s], cn = slab_at[s + 1] - c0;
    // (the Paeth filter of a chunk's first row reads the row above it: that row belongs to an earlier slab, already asked for)
    hipStream_t st = (s & 1) ? stream2 : stream;
    if (need_rows) { const int rc = need_rows(rows_of(c0 + cn), st); if (rc) return rc; }
    // the kernel writes its per-chunk results straight into the pinned host block (visible to the host behind the event).
    // As small device-to-host COPIES on this stream they shared the copy engine's queue with the slabs' big copies on the
    // aux stream: each big copy then took 0.9 ms instead of 0.4 and the PNG phase 6.8 ms instead of 3.4 (measured)
    uint8_t* r = res.p + 20 * per_slab * s;
    DeflArgs A;
    A.canvas = static_cast<const uint8_t*>(canvas); A.pitch = pitch; A.w = w; A.h = h;
    A.slots = scratch + o_slots;
    // (the kernel indexes the arrays with the global chunk number)
    A.len16 = reinterpret_cast<uint32_t*>(r) - c0; A.crc = reinterpret_cast<uint32_t*>(r + 4 * per_slab) - c0;
    A.ad_a = reinterpret_cast<uint32_t*>(r + 8 * per_slab) - c0; A.ad_b = reinterpret_cast<uint32_t*>(r + 12 * per_slab) - c0; A.ad_n = reinterpret_cast<uint32_t*>(r + 16 * per_slab) - c0;
    A.tables = reinterpret_cast<const uint32_t*>(scratch + o_T); A.xpow16 = reinterpret_cast<const uint32_t*>(scratch + o_pow);
    A.rows_per_chunk = g.rows_per_chunk; A.pieces_per_row = g.pieces_per_row; A.piece_px = g.piece_px;
    A.chunk0 = static_cast<int64_t>(c0);
    A.dbg = d_dbg;
    hipLaunchKernelGGL(ist_png_deflate_kernel, dim3(static_cast<unsigned>(cn)), dim3(256), 0, st, A);
    PNG_HIP(hipGetLastError());
    tl_mark("png:   compression launched, slab", static_cast<long>(s));
    PNG_HIP(hipEventCreateWithFlags(&evs.ev[s], hipEventDisableTiming));
    PNG_HIP(hipEventRecord(evs.ev[s], st));
    return IST_OK;
  };
  // one slab ahead: slab s+1 is compressing while the host lays out slab s and the aux stream carries it away
  { const int rc = compress(0); if (rc) return rc; }

  // ---- host, slab by slab: Adler-32 of the filtered stream, the layout of the chunks in the file, the CRC of every
  // IDAT; then the slab's gather (and its trip to the host) on the aux stream
  const uint64_t M = 65521;
  uint64_t a = 1, b = 0;
  struct Patch { int64_t at; uint8_t b[64]; int n; };
  int64_t* const dst = dstp.p;
  const int64_t limit = idat_limit();
  static const uint8_t trailer_block[5] = {0x01, 0x00, 0x00, 0xFF, 0xFF};          // final, empty stored block
  uint32_t reg = 0xFFFFFFFFu;
  auto feed = [&](const uint8_t* p, int k) { for (int i = 0; i < k; ++i) reg = crc_byte(T, reg, p[i]); };
  std::vector<std::vector<Patch>> slab_patches(n_slabs);     // (all of them live until the final synchronisation)
  std::vector<Patch>* cur = &slab_patches[0];
#define patches (*cur)
  {
    Patch pt; std::memset(&pt, 0, sizeof pt);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::memcpy(pt.b, sig, 8);
    put32(pt.b + 8, 13); std::memcpy(pt.b + 12, "IHDR", 4);
    put32(pt.b + 16, static_cast<uint32_t>(w)); put32(pt.b + 20, static_cast<uint32_t>(h));
    pt.b[24] = 8; pt.b[25] = 6; pt.b[26] = 0; pt.b[27] = 0; pt.b[28] = 0;
    uint32_t c = 0xFFFFFFFFu;
    for (int i = 12; i < 29; ++i) c = crc_byte(T, c, pt.b[i]);
    put32(pt.b + 29, c ^ 0xFFFFFFFFu);
    // an 11-byte tEXt chunk: its only job is to start the IDAT data at file offset 64
    put32(pt.b + 33, 11); std::memcpy(pt.b + 37, "tEXtSoftware\0is", 15);
    c = 0xFFFFFFFFu;
    for (int i = 37; i < 52; ++i) c = crc_byte(T, c, pt.b[i]);
    put32(pt.b + 52, c ^ 0xFFFFFFFFu);
    pt.at = 0; pt.n = 56;
    patches.push_back(pt);
  }
  int64_t pos = 56, idat_len_at = 0, idat_data = 0;
  auto open_idat = [&]() {
    Patch pt; std::memset(&pt, 0, sizeof pt);
    pt.at = pos; idat_len_at = pos;
    std::memcpy(pt.b + 4, "IDAT", 4);
    pt.n = 8; patches.push_back(pt);
    reg = 0xFFFFFFFFu; feed(pt.b + 4, 4);
    pos += 8; idat_data = 0;
  };
  auto close_idat = [&]() {
    Patch pt; std::memset(&pt, 0, sizeof pt);
    pt.at = pos; put32(pt.b, reg ^ 0xFFFFFFFFu); pt.n = 4; patches.push_back(pt);
    Patch lp; std::memset(&lp, 0, sizeof lp);
    lp.at = idat_len_at; put32(lp.b, static_cast<uint32_t>(idat_data)); lp.n = 4; patches.push_back(lp);
    pos += 4;
  };
  for (size_t s = 0; s < n_slabs; ++s) {
    const size_t c0 = slab_at[s], cn = slab_at[s + 1] - c0;
    if (s + 1 < n_slabs) { const int rc = compress(s + 1); if (rc) return rc; tl_mark("png: submitted the compression of slab", static_cast<long>(s + 1)); }
    PNG_HIP(hipEventSynchronize(evs.ev[s]));
    tl_mark("png: compressed (event passed), slab", static_cast<long>(s));
    if (fail_at >= 0 && static_cast<size_t>(fail_at) == s) return fail(IST_E_HIP, "PNG deflate kernel returned an impossible chunk length (forced: IST_PNG_FAIL_AT)");
    const uint8_t* r = res.p + 20 * per_slab * s;
    const uint32_t* len16 = reinterpret_cast<const uint32_t*>(r);
    const uint32_t* crc = reinterpret_cast<const uint32_t*>(r + 4 * per_slab);
    const uint32_t* ada = reinterpret_cast<const uint32_t*>(r + 8 * per_slab);
    const uint32_t* adb = reinterpret_cast<const uint32_t*>(r + 12 * per_slab);
    const uint32_t* adn = reinterpret_cast<const uint32_t*>(r + 16 * per_slab);
    const int64_t slab_begin = s == 0 ? 0 : pos;       // file bytes [slab_begin, pos) are final once this slab is laid out
    cur = &slab_patches[s];
    open_idat();                                       // a slab starts its own IDAT
    for (size_t k = 0; k < cn; ++k) {
      const size_t j = c0 + k;
      b = (b + (adn[k] % M) * a + adb[k]) % M;
      a = (a + ada[k]) % M;
      const int64_t bytes = static_cast<int64_t>(len16[k]) * 16;
      if (bytes <= 0 || bytes > SLOT) return fail(IST_E_HIP, "PNG deflate kernel returned an impossible chunk length");
      if (idat_data > 0 && idat_data + bytes + 9 > limit) { close_idat(); open_idat(); }
      dst[j] = pos;
      reg = gf_mul(xpow[static_cast<size_t>(len16[k])], reg) ^ crc[k];
      pos += bytes; idat_data += bytes;
    }
    if (s + 1 == n_slabs) {
      const uint32_t adler = static_cast<uint32_t>((b << 16) | a);
      Patch pt; std::memset(&pt, 0, sizeof pt);
      pt.at = pos;
      std::memcpy(pt.b, trailer_block, 5); put32(pt.b + 5, adler);
      feed(pt.b, 9);
      pt.n = 9; patches.push_back(pt);
      pos += 9; idat_data += 9;
      close_idat();
      Patch ie; std::memset(&ie, 0, sizeof ie);
      ie.at = pos; put32(ie.b, 0); std::memcpy(ie.b + 4, "IEND", 4);
      uint32_t c = 0xFFFFFFFFu;
      for (int i = 4; i < 8; ++i) c = crc_byte(T, c, ie.b[i]);
      put32(ie.b + 8, c ^ 0xFFFFFFFFu);
      ie.n = 12; patches.push_back(ie);
      pos += 12;
    } else close_idat();
    if (pos > out_cap) return fail(IST_E_INVALID, "PNG output buffer too small (see ist_png_bound)");
    // The gather rides on the stream that compressed the slab (its slots and results are complete: the event above has passed;
    // the next slab is already compressing on the other stream), and the aux stream carries NOTHING but the slabs' trips over
    // PCIe, each behind its gather's event.  With gather and copy both on aux every slab cost gather + copy in series (0.1 +
    // 0.3 ms, nine times: the whole phase was that stream, 4.0 ms for 2.75 ms of PCIe - measured, rocprofv3 timeline).
    // (Tried and dropped: the gather writing straight into the pinned file image with a small grid, no device image and no
    // copy - 4-byte-aligned 16-byte stores over PCIe from 128 workgroups were slower than gather + the runtime's copy: the phase
    // went from 3.66 to 4.23 ms.)
    tl_mark("png:   host layout done, slab", static_cast<long>(s));
    hipStream_t gs = host_out ? ((s & 1) ? stream2 : stream) : aux;
    GatherArgs G{scratch + o_slots, static_cast<uint8_t*>(out), dst, reinterpret_cast<const uint32_t*>(res.p + 20 * per_slab * s) - c0, static_cast<int64_t>(c0)};
    hipLaunchKernelGGL(ist_png_gather_kernel, dim3(static_cast<unsigned>(cn)), dim3(256), 0, gs, G);
    PNG_HIP(hipGetLastError());
    tl_mark("png:   gather launched, slab", static_cast<long>(s));
    if (host_out) {
      PNG_HIP(hipEventCreateWithFlags(&gathered.ev[s], hipEventDisableTiming));
      PNG_HIP(hipEventRecord(gathered.ev[s], gs));
      PNG_HIP(hipStreamWaitEvent(aux, gathered.ev[s], 0));
      tl_mark("png:   event created + recorded + aux ordered behind it, slab", static_cast<long>(s));
    }
    if (!host_out)                                     // (with a host sink the headers are written there, below)
      for (const Patch& pt : patches)
        PNG_HIP(hipMemcpyAsync(static_cast<uint8_t*>(out) + pt.at, pt.b, static_cast<size_t>(pt.n), hipMemcpyHostToDevice, aux));
    if (host_out) {
      // 256-byte aligned ends.  The bytes past `pos` in the last 256 are the next slab's: its own copy, ordered behind
      // this one, delivers them
      const int64_t c_lo = slab_begin & ~255ll, c_hi = std::min<int64_t>((pos + 255) & ~255ll, out_cap);
      static const bool skip_d2h = tuning_mode() && std::getenv("IST_PNG_SKIP_D2H") != nullptr;      // experiment: what the copies cost the kernel beside them (the file is then NOT delivered)
      static const int copy_grid = (tuning_mode() && std::getenv("IST_PNG_COPY_GRID")) ? std::atoi(std::getenv("IST_PNG_COPY_GRID")) : kCopyGrid;
      if (!skip_d2h && copy_grid > 0) {
        hipLaunchKernelGGL(ist_png_to_host_kernel, dim3(static_cast<unsigned>(copy_grid)), dim3(256), 0, aux, reinterpret_cast<const u32x4*>(static_cast<const uint8_t*>(out) + c_lo),
                           reinterpret_cast<u32x4*>(host_out + c_lo), (c_hi - c_lo) / 16);
        PNG_HIP(hipGetLastError());
      } else if (!skip_d2h) {
        // (round 4, tools/exp_slow_call.py + the marks around this call: one call in 600 - 3000 loses 6.5 - 7.3 ms INSIDE this
        // hipMemcpyAsync - the runtime's submission of one slab's device-to-host copy, slab 4, 8 or 9, on four different boxes; the
        // only stall of the call that is neither a wait of ours nor host scheduling.  Bounding the copies queued on the aux stream to
        // four - waiting for copy s-4's event before submitting copy s - did not remove it (1 and 3 such calls in 2 x 3000, against 2
        // and 1 unbounded) and cost 0.15 - 0.2 ms of the median call: dropped.)
        PNG_HIP(hipMemcpyAsync(host_out + c_lo, static_cast<const uint8_t*>(out) + c_lo, static_cast<size_t>(c_hi - c_lo), hipMemcpyDeviceToHost, aux));
      }
      tl_mark("png: laid out, gather + copy submitted, slab", static_cast<long>(s));
    }
  }
  tl_mark("png: every slab submitted; waiting for the copies (aux stream)");
  PNG_HIP(hipStreamSynchronize(aux));
  tl_mark("png: aux stream idle (the file is in host memory)");
  if (aux != stream) PNG_HIP(hipStreamSynchronize(stream));
  if (stream2 != stream) PNG_HIP(hipStreamSynchronize(stream2));
  tl_mark("png: encoder streams idle");
  drain.armed = false;                                 // the streams are idle
  if (d_dbg) {
    std::vector<unsigned long long> hdbg(8 * n);
    (void)hipMemcpy(hdbg.data(), d_dbg, 64 * n, hipMemcpyDeviceToHost);
    double sum[6] = {0, 0, 0, 0, 0, 0};
    for (size_t j = 0; j < n; ++j) for (int k = 0; k < 6; ++k) sum[k] += static_cast<double>(hdbg[8 * j + k + 1] - hdbg[8 * j + k]);
    static const char* names[6] = {"A load+filter", "B histogram+adler", "C code build", "D bit counts+scan", "body emit", "E slot write+crc"};
    for (int k = 0; k < 6; ++k) std::fprintf(stderr, "[png phases] %-18s %8.2f us per chunk (100 MHz clock)\n", names[k], sum[k] / static_cast<double>(n) / 100.0);
  }
#undef PNG_HIP
#undef patches
  if (host_out)                                        // signature, chunk headers, lengths, CRCs, trailer: ~30 bytes per slab
    for (const std::vector<Patch>& v : slab_patches)
      for (const Patch& pt : v) std::memcpy(host_out + pt.at, pt.b, static_cast<size_t>(pt.n));
  *out_len = pos;
  return IST_OK;
}

}  // namespace ist
