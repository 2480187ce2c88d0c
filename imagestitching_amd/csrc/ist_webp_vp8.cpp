// ist_webp_vp8.cpp — the lossy WebP bitstream: one VP8 key frame -> RGBA8 (alpha 255; an ALPH chunk is applied by
// ist_webp.cpp).  SURVEY.md section 8f rank 3: 'webp' is in SUPPORTED_IMAGE_TYPES (pages/index/index.js:4) and what phone
// galleries usually hand over under that extension is the lossy form.
//
// Source of the algorithm: the published format, RFC 6386 "VP8 Data Format and Decoding Guide" (frame header section 9 /
// 19.2, boolean entropy decoder 7, intra prediction 12, token decoding 13, dequantisation 14.1, inverse transforms 14.3-14.4,
// loop filter 15) and, for the step the RFC leaves to the application, libwebp's DOCUMENTED output conversion (the
// "fancy" 9-3-3-1 chroma upsampler and the 14-bit fixed-point YUV -> RGB of its dsp/yuv.h), so that the bytes equal what
// PIL / libwebp return for the same file (tests/test_webp_decode.py).  The format's constant tables are generated
// (ist_webp_vp8_tables.h, tools/extract_vp8_tables.py).
//
// Host code: the token partitions are serial boolean-coded streams and intra prediction chains every block to its
// reconstructed neighbours, so the frame is rebuilt on the host like the PNG / VP8L paths.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ist_internal.h"
#include "ist_webp.h"
#include "ist_webp_vp8_tables.h"

namespace ist {

namespace {

using namespace vp8;

// ---- boolean entropy decoder (RFC 6386 section 7.3) -----------------------------------------------------------------
struct BoolDec {
  const uint8_t* p = nullptr; const uint8_t* end = nullptr;
  uint32_t value = 0, range = 255;
  int bit_count = 0;
  void init(const uint8_t* b, size_t n) {
    p = b; end = b + n; range = 255; bit_count = 0;
    value = static_cast<uint32_t>(next()) << 8;
    value |= next();
  }
  inline uint32_t next() { return p < end ? *p++ : 0u; }
  inline int bit(int prob) {
    const uint32_t split = 1 + (((range - 1) * static_cast<uint32_t>(prob)) >> 8);
    const uint32_t big = split << 8;
    int r;
    if (value >= big) { r = 1; range -= split; value -= big; } else { r = 0; range = split; }
    while (range < 128) {
      value <<= 1; range <<= 1;
      if (++bit_count == 8) { bit_count = 0; value |= next(); }
    }
    return r;
  }
  inline int literal(int n) { int v = 0; while (n-- > 0) v |= bit(128) << n; return v; }
  inline int sliteral(int n) { const int v = literal(n); return bit(128) ? -v : v; }
};

enum { B_DC = 0, B_TM, B_VE, B_HE, B_RD, B_VR, B_LD, B_VL, B_HD, B_HU };      // sub-block modes; DC/TM/V/H of a whole block share 0..3

const uint8_t kZigzag[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
const uint8_t kBands[17] = {0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7, 0};
const uint8_t kCat3[] = {173, 148, 140, 0}, kCat4[] = {176, 155, 140, 135, 0}, kCat5[] = {180, 157, 141, 134, 130, 0};
const uint8_t kCat6[] = {254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129, 0};
const uint8_t* const kCat3456[4] = {kCat3, kCat4, kCat5, kCat6};

typedef uint8_t Probs[8][3][11];          // [band][context][node] of one block type

inline int large_value(BoolDec& br, const uint8_t* p) {
  int v;
  if (!br.bit(p[3])) {
    if (!br.bit(p[4])) v = 2; else v = 3 + br.bit(p[5]);
  } else if (!br.bit(p[6])) {
    if (!br.bit(p[7])) v = 5 + br.bit(159);
    else { v = 7 + 2 * br.bit(165); v += br.bit(145); }
  } else {
    const int bit1 = br.bit(p[8]), bit0 = br.bit(p[9 + bit1]);
    const int cat = 2 * bit1 + bit0;
    v = 0;
    for (const uint8_t* t = kCat3456[cat]; *t; ++t) v += v + br.bit(*t);
    v += 3 + (8 << cat);
  }
  return v;
}

// tokens of one 4x4 block (section 13): returns the index after the last non-zero coefficient
int get_coeffs(BoolDec& br, const Probs& probs, int ctx, int dq_dc, int dq_ac, int n, int16_t* out) {
  const uint8_t* p = probs[kBands[n]][ctx];
  for (; n < 16; ++n) {
    if (!br.bit(p[0])) return n;                       // end of block
    while (!br.bit(p[1])) {                            // zeros
      p = probs[kBands[++n]][0];
      if (n == 16) return 16;
    }
    const uint8_t (*next)[11] = probs[kBands[n + 1]];
    int v;
    if (!br.bit(p[2])) { v = 1; p = next[1]; } else { v = large_value(br, p); p = next[2]; }
    out[kZigzag[n]] = static_cast<int16_t>((br.bit(128) ? -v : v) * (n > 0 ? dq_ac : dq_dc));
  }
  return 16;
}

// ---- inverse transforms (section 14.3, 14.4) ------------------------------------------------------------------------
inline int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
// (64-bit products: a hostile file's coefficients reach 2^16 and more after the first pass, where the int product the format's
// reference code uses overflows; valid streams never get there, so the results are unchanged)
inline int mul1(int a) { return static_cast<int>((static_cast<int64_t>(a) * 20091) >> 16) + a; }
inline int mul2(int a) { return static_cast<int>((static_cast<int64_t>(a) * 35468) >> 16); }

void idct_add(const int16_t* in, uint8_t* dst, int stride) {
  int tmp[16];
  int* t = tmp;
  for (int i = 0; i < 4; ++i) {
    const int a = in[0] + in[8], b = in[0] - in[8];
    const int c = mul2(in[4]) - mul1(in[12]), d = mul1(in[4]) + mul2(in[12]);
    t[0] = a + d; t[1] = b + c; t[2] = b - c; t[3] = a - d;
    t += 4; ++in;
  }
  t = tmp;
  for (int i = 0; i < 4; ++i) {
    const int dc = t[0] + 4;
    const int a = dc + t[8], b = dc - t[8];
    const int c = mul2(t[4]) - mul1(t[12]), d = mul1(t[4]) + mul2(t[12]);
    dst[0] = static_cast<uint8_t>(clip8(dst[0] + ((a + d) >> 3)));
    dst[1] = static_cast<uint8_t>(clip8(dst[1] + ((b + c) >> 3)));
    dst[2] = static_cast<uint8_t>(clip8(dst[2] + ((b - c) >> 3)));
    dst[3] = static_cast<uint8_t>(clip8(dst[3] + ((a - d) >> 3)));
    ++t; dst += stride;
  }
}

void iwht(const int16_t* in, int16_t* dst) {            // dst: the DC slot of each of the 16 luma blocks (stride 16)
  int tmp[16];
  for (int i = 0; i < 4; ++i) {
    const int a0 = in[0 + i] + in[12 + i], a1 = in[4 + i] + in[8 + i], a2 = in[4 + i] - in[8 + i], a3 = in[0 + i] - in[12 + i];
    tmp[0 + i] = a0 + a1; tmp[8 + i] = a0 - a1; tmp[4 + i] = a3 + a2; tmp[12 + i] = a3 - a2;
  }
  for (int i = 0; i < 4; ++i) {
    const int dc = tmp[0 + i * 4] + 3;
    const int a0 = dc + tmp[3 + i * 4], a1 = tmp[1 + i * 4] + tmp[2 + i * 4], a2 = tmp[1 + i * 4] - tmp[2 + i * 4], a3 = dc - tmp[3 + i * 4];
    dst[0] = static_cast<int16_t>((a0 + a1) >> 3); dst[16] = static_cast<int16_t>((a3 + a2) >> 3);
    dst[32] = static_cast<int16_t>((a0 - a1) >> 3); dst[48] = static_cast<int16_t>((a3 - a2) >> 3);
    dst += 64;
  }
}

// ---- intra prediction (section 12) ------------------------------------------------------------------------------------
// planes carry a one-pixel border: the row above the frame reads 127 (corner included), the column left of it 129
inline int avg2(int a, int b) { return (a + b + 1) >> 1; }
inline int avg3(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }

void predict_block(uint8_t* d, int stride, int size, int mode, bool has_top, bool has_left) {       // 16x16 luma / 8x8 chroma
  const uint8_t* top = d - stride;
  if (mode == B_DC) {
    int dc;
    const int shift = size == 16 ? 4 : 3;
    if (has_top && has_left) { int s = 0; for (int i = 0; i < size; ++i) s += top[i] + d[i * stride - 1]; dc = (s + size) >> (shift + 1); }
    else if (has_top) { int s = 0; for (int i = 0; i < size; ++i) s += top[i]; dc = (s + (size >> 1)) >> shift; }
    else if (has_left) { int s = 0; for (int i = 0; i < size; ++i) s += d[i * stride - 1]; dc = (s + (size >> 1)) >> shift; }
    else dc = 128;
    for (int y = 0; y < size; ++y) std::memset(d + y * stride, dc, static_cast<size_t>(size));
  } else if (mode == B_TM) {
    const int tl = top[-1];
    for (int y = 0; y < size; ++y) { const int l = d[y * stride - 1] - tl; for (int x = 0; x < size; ++x) d[y * stride + x] = static_cast<uint8_t>(clip8(top[x] + l)); }
  } else if (mode == B_VE) {
    for (int y = 0; y < size; ++y) std::memcpy(d + y * stride, top, static_cast<size_t>(size));
  } else {
    for (int y = 0; y < size; ++y) std::memset(d + y * stride, d[y * stride - 1], static_cast<size_t>(size));
  }
}

// one 4x4 sub-block; tr = the four pixels above and to the right of it
void predict_4x4(uint8_t* d, int stride, int mode, const uint8_t* tr) {
  const uint8_t* top = d - stride;
  const int X = top[-1], A = top[0], B = top[1], C = top[2], D = top[3], E = tr[0], F = tr[1], G = tr[2], H = tr[3];
  const int I = d[-1], J = d[stride - 1], K = d[2 * stride - 1], L = d[3 * stride - 1];
#define DST(x, y) d[(x) + (y) * stride]
  switch (mode) {
    case B_DC: {
      const int dc = (A + B + C + D + I + J + K + L + 4) >> 3;
      for (int y = 0; y < 4; ++y) std::memset(d + y * stride, dc, 4);
      break;
    }
    case B_TM:
      for (int y = 0; y < 4; ++y) { const int l = d[y * stride - 1] - X; for (int x = 0; x < 4; ++x) DST(x, y) = static_cast<uint8_t>(clip8(top[x] + l)); }
      break;
    case B_VE: {
      const uint8_t v[4] = {static_cast<uint8_t>(avg3(X, A, B)), static_cast<uint8_t>(avg3(A, B, C)), static_cast<uint8_t>(avg3(B, C, D)), static_cast<uint8_t>(avg3(C, D, E))};
      for (int y = 0; y < 4; ++y) std::memcpy(d + y * stride, v, 4);
      break;
    }
    case B_HE:
      std::memset(d, avg3(X, I, J), 4); std::memset(d + stride, avg3(I, J, K), 4);
      std::memset(d + 2 * stride, avg3(J, K, L), 4); std::memset(d + 3 * stride, avg3(K, L, L), 4);
      break;
    case B_RD:
      DST(0, 3) = static_cast<uint8_t>(avg3(J, K, L));
      DST(0, 2) = DST(1, 3) = static_cast<uint8_t>(avg3(I, J, K));
      DST(0, 1) = DST(1, 2) = DST(2, 3) = static_cast<uint8_t>(avg3(X, I, J));
      DST(0, 0) = DST(1, 1) = DST(2, 2) = DST(3, 3) = static_cast<uint8_t>(avg3(A, X, I));
      DST(1, 0) = DST(2, 1) = DST(3, 2) = static_cast<uint8_t>(avg3(B, A, X));
      DST(2, 0) = DST(3, 1) = static_cast<uint8_t>(avg3(C, B, A));
      DST(3, 0) = static_cast<uint8_t>(avg3(D, C, B));
      break;
    case B_VR:
      DST(0, 0) = DST(1, 2) = static_cast<uint8_t>(avg2(X, A));
      DST(1, 0) = DST(2, 2) = static_cast<uint8_t>(avg2(A, B));
      DST(2, 0) = DST(3, 2) = static_cast<uint8_t>(avg2(B, C));
      DST(3, 0) = static_cast<uint8_t>(avg2(C, D));
      DST(0, 3) = static_cast<uint8_t>(avg3(K, J, I));
      DST(0, 2) = static_cast<uint8_t>(avg3(J, I, X));
      DST(0, 1) = DST(1, 3) = static_cast<uint8_t>(avg3(I, X, A));
      DST(1, 1) = DST(2, 3) = static_cast<uint8_t>(avg3(X, A, B));
      DST(2, 1) = DST(3, 3) = static_cast<uint8_t>(avg3(A, B, C));
      DST(3, 1) = static_cast<uint8_t>(avg3(B, C, D));
      break;
    case B_LD:
      DST(0, 0) = static_cast<uint8_t>(avg3(A, B, C));
      DST(1, 0) = DST(0, 1) = static_cast<uint8_t>(avg3(B, C, D));
      DST(2, 0) = DST(1, 1) = DST(0, 2) = static_cast<uint8_t>(avg3(C, D, E));
      DST(3, 0) = DST(2, 1) = DST(1, 2) = DST(0, 3) = static_cast<uint8_t>(avg3(D, E, F));
      DST(3, 1) = DST(2, 2) = DST(1, 3) = static_cast<uint8_t>(avg3(E, F, G));
      DST(3, 2) = DST(2, 3) = static_cast<uint8_t>(avg3(F, G, H));
      DST(3, 3) = static_cast<uint8_t>(avg3(G, H, H));
      break;
    case B_VL:
      DST(0, 0) = static_cast<uint8_t>(avg2(A, B));
      DST(1, 0) = DST(0, 2) = static_cast<uint8_t>(avg2(B, C));
      DST(2, 0) = DST(1, 2) = static_cast<uint8_t>(avg2(C, D));
      DST(3, 0) = DST(2, 2) = static_cast<uint8_t>(avg2(D, E));
      DST(0, 1) = static_cast<uint8_t>(avg3(A, B, C));
      DST(1, 1) = DST(0, 3) = static_cast<uint8_t>(avg3(B, C, D));
      DST(2, 1) = DST(1, 3) = static_cast<uint8_t>(avg3(C, D, E));
      DST(3, 1) = DST(2, 3) = static_cast<uint8_t>(avg3(D, E, F));
      DST(3, 2) = static_cast<uint8_t>(avg3(E, F, G));
      DST(3, 3) = static_cast<uint8_t>(avg3(F, G, H));
      break;
    case B_HD:
      DST(0, 3) = static_cast<uint8_t>(avg2(L, K));
      DST(0, 2) = DST(2, 3) = static_cast<uint8_t>(avg2(K, J));
      DST(0, 1) = DST(2, 2) = static_cast<uint8_t>(avg2(J, I));
      DST(0, 0) = DST(2, 1) = static_cast<uint8_t>(avg2(I, X));
      DST(3, 0) = static_cast<uint8_t>(avg3(A, B, C));
      DST(2, 0) = static_cast<uint8_t>(avg3(X, A, B));
      DST(1, 0) = DST(3, 1) = static_cast<uint8_t>(avg3(I, X, A));
      DST(1, 1) = DST(3, 2) = static_cast<uint8_t>(avg3(J, I, X));
      DST(1, 2) = DST(3, 3) = static_cast<uint8_t>(avg3(K, J, I));
      DST(1, 3) = static_cast<uint8_t>(avg3(L, K, J));
      break;
    default:   // B_HU
      DST(0, 0) = static_cast<uint8_t>(avg2(I, J));
      DST(2, 0) = DST(0, 1) = static_cast<uint8_t>(avg2(J, K));
      DST(2, 1) = DST(0, 2) = static_cast<uint8_t>(avg2(K, L));
      DST(1, 0) = static_cast<uint8_t>(avg3(I, J, K));
      DST(3, 0) = DST(1, 1) = static_cast<uint8_t>(avg3(J, K, L));
      DST(3, 1) = DST(1, 2) = static_cast<uint8_t>(avg3(K, L, L));
      DST(3, 2) = DST(2, 2) = DST(0, 3) = DST(1, 3) = DST(2, 3) = DST(3, 3) = static_cast<uint8_t>(L);
      break;
  }
#undef DST
}

// ---- loop filter (section 15) -------------------------------------------------------------------------------------------
inline int sclip1(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }     // [-1020, 1020] -> [-128, 127]
inline int sclip2(int v) { return v < -16 ? -16 : (v > 15 ? 15 : v); }         // [-112, 112] -> [-16, 15]

inline void filter2(uint8_t* p, int step) {
  const int p1 = p[-2 * step], p0 = p[-step], q0 = p[0], q1 = p[step];
  const int a = 3 * (q0 - p0) + sclip1(p1 - q1);
  const int a1 = sclip2((a + 4) >> 3), a2 = sclip2((a + 3) >> 3);
  p[-step] = static_cast<uint8_t>(clip8(p0 + a2));
  p[0] = static_cast<uint8_t>(clip8(q0 - a1));
}
inline void filter4(uint8_t* p, int step) {
  const int p1 = p[-2 * step], p0 = p[-step], q0 = p[0], q1 = p[step];
  const int a = 3 * (q0 - p0);
  const int a1 = sclip2((a + 4) >> 3), a2 = sclip2((a + 3) >> 3), a3 = (a1 + 1) >> 1;
  p[-2 * step] = static_cast<uint8_t>(clip8(p1 + a3));
  p[-step] = static_cast<uint8_t>(clip8(p0 + a2));
  p[0] = static_cast<uint8_t>(clip8(q0 - a1));
  p[step] = static_cast<uint8_t>(clip8(q1 - a3));
}
inline void filter6(uint8_t* p, int step) {
  const int p2 = p[-3 * step], p1 = p[-2 * step], p0 = p[-step], q0 = p[0], q1 = p[step], q2 = p[2 * step];
  const int a = sclip1(3 * (q0 - p0) + sclip1(p1 - q1));
  const int a1 = (27 * a + 63) >> 7, a2 = (18 * a + 63) >> 7, a3 = (9 * a + 63) >> 7;
  p[-3 * step] = static_cast<uint8_t>(clip8(p2 + a3));
  p[-2 * step] = static_cast<uint8_t>(clip8(p1 + a2));
  p[-step] = static_cast<uint8_t>(clip8(p0 + a1));
  p[0] = static_cast<uint8_t>(clip8(q0 - a1));
  p[step] = static_cast<uint8_t>(clip8(q1 - a2));
  p[2 * step] = static_cast<uint8_t>(clip8(q2 - a3));
}
inline bool needs_filter(const uint8_t* p, int step, int t) {
  const int p1 = p[-2 * step], p0 = p[-step], q0 = p[0], q1 = p[step];
  return (4 * std::abs(p0 - q0) + std::abs(p1 - q1)) <= t;
}
inline bool needs_filter2(const uint8_t* p, int step, int t, int it) {
  const int p3 = p[-4 * step], p2 = p[-3 * step], p1 = p[-2 * step], p0 = p[-step], q0 = p[0], q1 = p[step], q2 = p[2 * step], q3 = p[3 * step];
  if ((4 * std::abs(p0 - q0) + std::abs(p1 - q1)) > t) return false;
  return std::abs(p3 - p2) <= it && std::abs(p2 - p1) <= it && std::abs(p1 - p0) <= it && std::abs(q3 - q2) <= it && std::abs(q2 - q1) <= it && std::abs(q1 - q0) <= it;
}
inline bool hev(const uint8_t* p, int step, int thresh) { return std::abs(p[-2 * step] - p[-step]) > thresh || std::abs(p[step] - p[0]) > thresh; }

// `along` = distance between the pixels of the edge, `across` = distance between the taps of one filter
void simple_edge(uint8_t* p, int along, int across, int size, int thresh) {
  const int t2 = 2 * thresh + 1;
  for (int i = 0; i < size; ++i, p += along) if (needs_filter(p, across, t2)) filter2(p, across);
}
void mb_edge(uint8_t* p, int along, int across, int size, int thresh, int ithresh, int hev_t) {     // macroblock edges: 6-tap
  const int t2 = 2 * thresh + 1;
  for (int i = 0; i < size; ++i, p += along)
    if (needs_filter2(p, across, t2, ithresh)) { if (hev(p, across, hev_t)) filter2(p, across); else filter6(p, across); }
}
void inner_edge(uint8_t* p, int along, int across, int size, int thresh, int ithresh, int hev_t) {  // sub-block edges: 4-tap
  const int t2 = 2 * thresh + 1;
  for (int i = 0; i < size; ++i, p += along)
    if (needs_filter2(p, across, t2, ithresh)) { if (hev(p, across, hev_t)) filter2(p, across); else filter4(p, across); }
}

// ---- frame ---------------------------------------------------------------------------------------------------------------
struct Header {
  int w = 0, h = 0;
  size_t first_part = 0;
};

int parse_tag(const uint8_t* d, size_t n, Header* H) {
  if (n < 10) return fail(IST_E_DECODE, "WebP: truncated VP8 frame");
  const uint32_t tag = d[0] | (d[1] << 8) | (static_cast<uint32_t>(d[2]) << 16);
  if (tag & 1) return fail(IST_E_DECODE, "WebP: the VP8 frame is not a key frame");
  if (((tag >> 1) & 7) > 3) return fail(IST_E_DECODE, "WebP: unknown VP8 profile");
  if (d[3] != 0x9D || d[4] != 0x01 || d[5] != 0x2A) return fail(IST_E_DECODE, "WebP: bad VP8 start code");
  H->first_part = tag >> 5;
  H->w = (d[6] | (d[7] << 8)) & 0x3FFF;
  H->h = (d[8] | (d[9] << 8)) & 0x3FFF;
  if (H->w < 1 || H->h < 1) return fail(IST_E_DECODE, "WebP: bad VP8 frame size");
  return IST_OK;
}

struct MbInfo { uint8_t ymode, uvmode, segment, skip, is_i4x4, inner; uint8_t modes[16]; };
struct FilterInfo { uint8_t limit, ilevel, hev; };

inline int clipq(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

int decode_frame(const uint8_t* d, size_t n, uint8_t* out, size_t pitch) {
  Header H;
  int rc = parse_tag(d, n, &H);
  if (rc) return rc;
  if (10 + H.first_part > n) return fail(IST_E_DECODE, "WebP: truncated VP8 first partition");
  BoolDec br;
  br.init(d + 10, H.first_part);
  br.bit(128);                                                  // colour space
  br.bit(128);                                                  // clamping type (the reconstruction below always clamps)
  // segmentation
  const int seg_enabled = br.bit(128);
  int update_map = 0, abs_delta = 1;
  int seg_quant[4] = {0, 0, 0, 0}, seg_lf[4] = {0, 0, 0, 0};
  uint8_t seg_prob[3] = {255, 255, 255};
  if (seg_enabled) {
    update_map = br.bit(128);
    if (br.bit(128)) {
      abs_delta = br.bit(128);
      for (int i = 0; i < 4; ++i) seg_quant[i] = br.bit(128) ? br.sliteral(7) : 0;
      for (int i = 0; i < 4; ++i) seg_lf[i] = br.bit(128) ? br.sliteral(6) : 0;
    }
    if (update_map) for (int i = 0; i < 3; ++i) seg_prob[i] = br.bit(128) ? static_cast<uint8_t>(br.literal(8)) : 255;
  }
  // loop filter
  const int simple = br.bit(128);
  const int level = br.literal(6), sharpness = br.literal(3);
  int ref_delta[4] = {0, 0, 0, 0}, mode_delta[4] = {0, 0, 0, 0};
  const int delta_enabled = br.bit(128);
  if (delta_enabled && br.bit(128)) {
    for (int i = 0; i < 4; ++i) if (br.bit(128)) ref_delta[i] = br.sliteral(6);
    for (int i = 0; i < 4; ++i) if (br.bit(128)) mode_delta[i] = br.sliteral(6);
  }
  // token partitions
  const int n_parts = 1 << br.literal(2);
  const uint8_t* part = d + 10 + H.first_part;
  size_t left = n - 10 - H.first_part;
  if (left < 3u * static_cast<size_t>(n_parts - 1)) return fail(IST_E_DECODE, "WebP: truncated VP8 partition table");
  const uint8_t* sz = part;
  part += 3 * (n_parts - 1); left -= 3u * static_cast<size_t>(n_parts - 1);
  std::vector<BoolDec> tok(static_cast<size_t>(n_parts));
  for (int p = 0; p < n_parts; ++p) {
    size_t ps = left;
    if (p < n_parts - 1) { ps = sz[0] | (sz[1] << 8) | (static_cast<size_t>(sz[2]) << 16); if (ps > left) ps = left; sz += 3; }
    tok[static_cast<size_t>(p)].init(part, ps);
    part += ps; left -= ps;
  }
  // quantisers
  const int base_q = br.literal(7);
  const int dqy1_dc = br.bit(128) ? br.sliteral(4) : 0, dqy2_dc = br.bit(128) ? br.sliteral(4) : 0, dqy2_ac = br.bit(128) ? br.sliteral(4) : 0;
  const int dquv_dc = br.bit(128) ? br.sliteral(4) : 0, dquv_ac = br.bit(128) ? br.sliteral(4) : 0;
  struct Quant { int y1[2], y2[2], uv[2]; } quant[4];
  for (int s = 0; s < 4; ++s) {
    int q = base_q;
    if (seg_enabled) q = abs_delta ? seg_quant[s] : base_q + seg_quant[s];
    Quant& Q = quant[s];
    Q.y1[0] = kDcTable[clipq(q + dqy1_dc, 127)]; Q.y1[1] = kAcTable[clipq(q, 127)];
    Q.y2[0] = kDcTable[clipq(q + dqy2_dc, 127)] * 2;
    Q.y2[1] = (kAcTable[clipq(q + dqy2_ac, 127)] * 101581) >> 16;
    if (Q.y2[1] < 8) Q.y2[1] = 8;
    Q.uv[0] = kDcTable[clipq(q + dquv_dc, 117)]; Q.uv[1] = kAcTable[clipq(q + dquv_ac, 127)];
  }
  br.bit(128);                                                  // refresh_entropy_probs: there is no next frame
  // token probabilities
  static_assert(sizeof(Probs) * 4 == sizeof(kCoeffProbs0), "table shape");
  Probs probs[4];
  std::memcpy(probs, kCoeffProbs0, sizeof(kCoeffProbs0));
  {
    const uint8_t* up = kCoeffUpdateProbs;
    for (int t = 0; t < 4; ++t) for (int b = 0; b < 8; ++b) for (int c = 0; c < 3; ++c) for (int k = 0; k < 11; ++k, ++up)
      if (br.bit(*up)) probs[t][b][c][k] = static_cast<uint8_t>(br.literal(8));
  }
  const int use_skip = br.bit(128);
  const int skip_p = use_skip ? br.literal(8) : 0;

  const int mbw = (H.w + 15) >> 4, mbh = (H.h + 15) >> 4;
  // planes with a border: one row above (127) and one column left (129); luma gets 4 spare columns right for "top-right"
  const int ys = mbw * 16 + 1 + 4, cs = mbw * 8 + 1;
  std::vector<uint8_t> Yb(static_cast<size_t>(ys) * (mbh * 16 + 1)), Ub(static_cast<size_t>(cs) * (mbh * 8 + 1)), Vb(Ub.size());
  auto init_plane = [](std::vector<uint8_t>& P, int stride, int rows) {
    std::memset(P.data(), 127, static_cast<size_t>(stride));
    for (int y = 1; y <= rows; ++y) P[static_cast<size_t>(y) * stride] = 129;
  };
  init_plane(Yb, ys, mbh * 16); init_plane(Ub, cs, mbh * 8); init_plane(Vb, cs, mbh * 8);
  uint8_t* Y = Yb.data() + ys + 1; uint8_t* U = Ub.data() + cs + 1; uint8_t* V = Vb.data() + cs + 1;

  std::vector<MbInfo> mbs(static_cast<size_t>(mbw) * mbh);
  std::vector<uint8_t> intra_t(static_cast<size_t>(mbw) * 4, B_DC);
  std::vector<uint8_t> top_nz(static_cast<size_t>(mbw) * 9, 0);          // per macroblock column: 4 Y, 2 U, 2 V, 1 Y2
  int16_t coef[25 * 16];
  for (int my = 0; my < mbh; ++my) {
    uint8_t intra_l[4] = {B_DC, B_DC, B_DC, B_DC};
    uint8_t left_nz[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    BoolDec& tb = tok[static_cast<size_t>(my & (n_parts - 1))];
    for (int mx = 0; mx < mbw; ++mx) {
      MbInfo& M = mbs[static_cast<size_t>(my) * mbw + mx];
      // -- macroblock header (first partition, section 19.3)
      M.segment = 0;
      if (update_map) M.segment = static_cast<uint8_t>(!br.bit(seg_prob[0]) ? br.bit(seg_prob[1]) : 2 + br.bit(seg_prob[2]));
      M.skip = static_cast<uint8_t>(use_skip ? br.bit(skip_p) : 0);
      M.is_i4x4 = static_cast<uint8_t>(!br.bit(145));
      uint8_t* it = intra_t.data() + 4 * mx;
      if (!M.is_i4x4) {
        M.ymode = static_cast<uint8_t>(br.bit(156) ? (br.bit(128) ? B_TM : B_HE) : (br.bit(163) ? B_VE : B_DC));
        std::memset(it, M.ymode, 4); std::memset(intra_l, M.ymode, 4);
      } else {
        for (int y = 0; y < 4; ++y) {
          int ymode = intra_l[y];
          for (int x = 0; x < 4; ++x) {
            const uint8_t* p = kBModeProbs + (static_cast<size_t>(it[x]) * 10 + static_cast<size_t>(ymode)) * 9;
            ymode = !br.bit(p[0]) ? B_DC : !br.bit(p[1]) ? B_TM : !br.bit(p[2]) ? B_VE :
                    !br.bit(p[3]) ? (!br.bit(p[4]) ? B_HE : (!br.bit(p[5]) ? B_RD : B_VR))
                                  : (!br.bit(p[6]) ? B_LD : (!br.bit(p[7]) ? B_VL : (!br.bit(p[8]) ? B_HD : B_HU)));
            it[x] = static_cast<uint8_t>(ymode);
            M.modes[y * 4 + x] = static_cast<uint8_t>(ymode);
          }
          intra_l[y] = static_cast<uint8_t>(ymode);
        }
      }
      M.uvmode = static_cast<uint8_t>(!br.bit(142) ? B_DC : !br.bit(114) ? B_VE : br.bit(183) ? B_TM : B_HE);
      // -- residual tokens (token partition of this row)
      std::memset(coef, 0, sizeof coef);
      uint8_t* tnz = top_nz.data() + 9 * mx;
      bool any = false;
      if (!M.skip) {
        const Quant& Q = quant[M.segment];
        int first = 0;
        const Probs* ac = &probs[3];
        if (!M.is_i4x4) {
          int16_t y2[16];
          std::memset(y2, 0, sizeof y2);
          const int nz = get_coeffs(tb, probs[1], tnz[8] + left_nz[8], Q.y2[0], Q.y2[1], 0, y2);
          tnz[8] = left_nz[8] = static_cast<uint8_t>(nz > 0);
          if (nz > 0) { iwht(y2, coef); for (int k = 0; k < 16; ++k) any |= coef[k * 16] != 0; }
          first = 1; ac = &probs[0];
        }
        for (int y = 0; y < 4; ++y) for (int x = 0; x < 4; ++x) {
          int16_t* c = coef + (y * 4 + x) * 16;
          const int nz = get_coeffs(tb, *ac, tnz[x] + left_nz[y], Q.y1[0], Q.y1[1], first, c);
          tnz[x] = left_nz[y] = static_cast<uint8_t>(nz > first);
          any |= nz > first;
        }
        for (int ch = 0; ch < 2; ++ch) for (int y = 0; y < 2; ++y) for (int x = 0; x < 2; ++x) {
          int16_t* c = coef + (16 + ch * 4 + y * 2 + x) * 16;
          const int nz = get_coeffs(tb, probs[2], tnz[4 + ch * 2 + x] + left_nz[4 + ch * 2 + y], Q.uv[0], Q.uv[1], 0, c);
          tnz[4 + ch * 2 + x] = left_nz[4 + ch * 2 + y] = static_cast<uint8_t>(nz > 0);
          any |= nz > 0;
        }
      } else {
        std::memset(tnz, 0, 8); std::memset(left_nz, 0, 8);
        if (!M.is_i4x4) tnz[8] = left_nz[8] = 0;
      }
      M.inner = static_cast<uint8_t>(M.is_i4x4 || any);
      // -- reconstruction: prediction from the UNFILTERED neighbours + residue (sections 12, 14)
      uint8_t* yd = Y + static_cast<size_t>(my) * 16 * ys + mx * 16;
      uint8_t* ud = U + static_cast<size_t>(my) * 8 * cs + mx * 8;
      uint8_t* vd = V + static_cast<size_t>(my) * 8 * cs + mx * 8;
      if (M.is_i4x4) {
        // the four pixels above-right of the macroblock: the next macroblock's bottom row above, else a replica of the last
        // pixel above this one; the top row of the frame reads 127 throughout
        uint8_t tr[4];
        if (my == 0) std::memset(tr, 127, 4);
        else if (mx < mbw - 1) std::memcpy(tr, yd - ys + 16, 4);
        else std::memset(tr, yd[-ys + 15], 4);
        for (int by = 0; by < 4; ++by) for (int bx = 0; bx < 4; ++bx) {
          uint8_t* b = yd + by * 4 * ys + bx * 4;
          const uint8_t* trp = (bx < 3) ? b - ys + 4 : tr;       // column 3 uses the macroblock's above-right pixels on every row
          uint8_t local[4];
          if (bx < 3 && by == 0 && my == 0) { std::memset(local, 127, 4); trp = local; }
          predict_4x4(b, ys, M.modes[by * 4 + bx], trp);
          idct_add(coef + (by * 4 + bx) * 16, b, ys);
        }
      } else {
        predict_block(yd, ys, 16, M.ymode, my > 0, mx > 0);
        for (int by = 0; by < 4; ++by) for (int bx = 0; bx < 4; ++bx) idct_add(coef + (by * 4 + bx) * 16, yd + by * 4 * ys + bx * 4, ys);
      }
      predict_block(ud, cs, 8, M.uvmode, my > 0, mx > 0);
      predict_block(vd, cs, 8, M.uvmode, my > 0, mx > 0);
      for (int by = 0; by < 2; ++by) for (int bx = 0; bx < 2; ++bx) {
        idct_add(coef + (16 + by * 2 + bx) * 16, ud + by * 4 * cs + bx * 4, cs);
        idct_add(coef + (20 + by * 2 + bx) * 16, vd + by * 4 * cs + bx * 4, cs);
      }
    }
  }
  // -- loop filter over the reconstructed frame, macroblock by macroblock in raster order (section 15)
  if (level > 0) {                      // (a frame-level 0 switches the filter off whatever the segments say: libwebp's reading, and the witness here)
    FilterInfo fi[4][2];
    for (int s = 0; s < 4; ++s) {
      int base = level;
      if (seg_enabled) base = abs_delta ? seg_lf[s] : base + seg_lf[s];
      for (int i4 = 0; i4 < 2; ++i4) {
        int lv = base;
        if (delta_enabled) { lv += ref_delta[0]; if (i4) lv += mode_delta[0]; }
        lv = lv < 0 ? 0 : (lv > 63 ? 63 : lv);
        FilterInfo& F = fi[s][i4];
        F.limit = 0; F.ilevel = 0; F.hev = 0;
        if (lv > 0) {
          int il = lv;
          if (sharpness > 0) { il >>= (sharpness > 4) ? 2 : 1; if (il > 9 - sharpness) il = 9 - sharpness; }
          if (il < 1) il = 1;
          F.ilevel = static_cast<uint8_t>(il);
          F.limit = static_cast<uint8_t>(2 * lv + il);
          F.hev = static_cast<uint8_t>(lv >= 40 ? 2 : (lv >= 15 ? 1 : 0));
        }
      }
    }
    for (int my = 0; my < mbh; ++my) for (int mx = 0; mx < mbw; ++mx) {
      const MbInfo& M = mbs[static_cast<size_t>(my) * mbw + mx];
      const FilterInfo& F = fi[M.segment][M.is_i4x4];
      const int limit = F.limit;
      if (!limit) continue;
      uint8_t* yd = Y + static_cast<size_t>(my) * 16 * ys + mx * 16;
      if (simple) {
        if (mx > 0) simple_edge(yd, ys, 1, 16, limit + 4);
        if (M.inner) for (int k = 1; k < 4; ++k) simple_edge(yd + 4 * k, ys, 1, 16, limit);
        if (my > 0) simple_edge(yd, 1, ys, 16, limit + 4);
        if (M.inner) for (int k = 1; k < 4; ++k) simple_edge(yd + 4 * k * ys, 1, ys, 16, limit);
      } else {
        uint8_t* ud = U + static_cast<size_t>(my) * 8 * cs + mx * 8;
        uint8_t* vd = V + static_cast<size_t>(my) * 8 * cs + mx * 8;
        const int il = F.ilevel, hv = F.hev;
        if (mx > 0) { mb_edge(yd, ys, 1, 16, limit + 4, il, hv); mb_edge(ud, cs, 1, 8, limit + 4, il, hv); mb_edge(vd, cs, 1, 8, limit + 4, il, hv); }
        if (M.inner) {
          for (int k = 1; k < 4; ++k) inner_edge(yd + 4 * k, ys, 1, 16, limit, il, hv);
          inner_edge(ud + 4, cs, 1, 8, limit, il, hv); inner_edge(vd + 4, cs, 1, 8, limit, il, hv);
        }
        if (my > 0) { mb_edge(yd, 1, ys, 16, limit + 4, il, hv); mb_edge(ud, 1, cs, 8, limit + 4, il, hv); mb_edge(vd, 1, cs, 8, limit + 4, il, hv); }
        if (M.inner) {
          for (int k = 1; k < 4; ++k) inner_edge(yd + 4 * k * ys, 1, ys, 16, limit, il, hv);
          inner_edge(ud + 4 * cs, 1, cs, 8, limit, il, hv); inner_edge(vd + 4 * cs, 1, cs, 8, limit, il, hv);
        }
      }
    }
  }
  // -- output: 9-3-3-1 chroma upsampling + fixed-point YUV -> RGB, cropped to the frame size
  const int w = H.w, h = H.h, cw = (w + 1) >> 1;
  auto yuv2rgba = [](int y, int u, int v, uint8_t* o) {
    auto clip = [](int x) { return (x & ~16383) == 0 ? (x >> 6) : (x < 0 ? 0 : 255); };
    const int yy = (y * 19077) >> 8;
    o[0] = static_cast<uint8_t>(clip(yy + ((v * 26149) >> 8) - 14234));
    o[1] = static_cast<uint8_t>(clip(yy - ((u * 6419) >> 8) - ((v * 13320) >> 8) + 8708));
    o[2] = static_cast<uint8_t>(clip(yy + ((u * 33050) >> 8) - 17685));
    o[3] = 255;
  };
  // one output row from its two chroma rows: `near` weighs 3, `far` weighs 1 vertically
  auto emit_row = [&](int row, const uint8_t* nu, const uint8_t* nv, const uint8_t* fu, const uint8_t* fv) {
    const uint8_t* yr = Y + static_cast<size_t>(row) * ys;
    uint8_t* o = out + static_cast<size_t>(row) * pitch;
    {
      const int u0 = (3 * nu[0] + fu[0] + 2) >> 2, v0 = (3 * nv[0] + fv[0] + 2) >> 2;
      yuv2rgba(yr[0], u0, v0, o);
    }
    const int last_pair = (w - 1) >> 1;
    for (int x = 1; x <= last_pair; ++x) {
      // near row: a = nu[x-1], b = nu[x]; far row: c = fu[x-1], d = fu[x]
      const int ua = nu[x - 1], ub = nu[x], uc = fu[x - 1], ud2 = fu[x];
      const int va = nv[x - 1], vb = nv[x], vc = fv[x - 1], vd2 = fv[x];
      const int uavg = ua + ub + uc + ud2 + 8, vavg = va + vb + vc + vd2 + 8;
      // pixel 2x-1 leans to (near, x-1): (9a + 3b + 3c + d + 8) >> 4 in two steps, as the format's reference decoder rounds it
      const int u_l = (((uavg + 2 * (ub + uc)) >> 3) + ua) >> 1, v_l = (((vavg + 2 * (vb + vc)) >> 3) + va) >> 1;
      const int u_r = (((uavg + 2 * (ua + ud2)) >> 3) + ub) >> 1, v_r = (((vavg + 2 * (va + vd2)) >> 3) + vb) >> 1;
      yuv2rgba(yr[2 * x - 1], u_l, v_l, o + 4 * (2 * x - 1));
      yuv2rgba(yr[2 * x], u_r, v_r, o + 4 * (2 * x));
    }
    if (!(w & 1)) {
      const int u0 = (3 * nu[cw - 1] + fu[cw - 1] + 2) >> 2, v0 = (3 * nv[cw - 1] + fv[cw - 1] + 2) >> 2;
      yuv2rgba(yr[w - 1], u0, v0, o + 4 * (w - 1));
    }
  };
  for (int row = 0; row < h; ++row) {
    // rows 2k-1 and 2k share chroma rows k-1 and k; row 0 and (for an even height) the last row see one chroma row only
    const int k_near = row >> 1;
    int k_far = (row & 1) ? k_near + 1 : k_near - 1;
    const int ch = (h + 1) >> 1;
    if (k_far < 0 || k_far >= ch) k_far = k_near;
    emit_row(row, U + static_cast<size_t>(k_near) * cs, V + static_cast<size_t>(k_near) * cs, U + static_cast<size_t>(k_far) * cs, V + static_cast<size_t>(k_far) * cs);
  }
  return IST_OK;
}

}  // namespace

int vp8_info(const uint8_t* d, size_t n, int* w, int* h) {
  Header H;
  const int rc = parse_tag(d, n, &H);
  if (rc) return rc;
  *w = H.w; *h = H.h;
  return IST_OK;
}

int vp8_decode_rgba8(const uint8_t* d, size_t n, uint8_t* out, size_t pitch) {
  try { return decode_frame(d, n, out, pitch); }
  catch (const std::bad_alloc&) { return fail(IST_E_NOMEM, "out of memory while decoding the WebP"); }
}

}  // namespace ist
