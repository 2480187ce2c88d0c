// ist_kernels.hip — the fused resample + blit kernel for gfx950 (MI355X, CDNA4).
//
// Stands in for the raster work the WeChat Canvas performs for the calls onStitch issues
// (reference: pages/index/index.js:1423-1424 fillRect, utils/canvas.js:153-202 drawImage under a CTM,
// index.js:1577-1579 same-size readback).  ONE launch writes every canvas pixel exactly once:
//
//   grid      one 256-thread workgroup (4 wave64) per output tile; tiles enumerate the cells of ist_compile.cpp
//   tile      256 px x 8 rows for fill / copy cells (2 rows per wave: ~64 KB in flight per CU, the measured optimum);
//             a wave row is 64 lanes x 16 B = 1 KiB contiguous
//             (one global_store_dwordx4 per lane); 64 x 64 for general cells
//   paths     FILL    constant colour                                   (gaps, centring margins, rounding slack)
//             COPY    1:1 rect: 16-B nt loads -> 16-B nt stores (the BASELINE configs)
//             SAMPLE_LDS bilinear with the source footprint staged in LDS by LDS-DMA (mixed-size strips)
//             SAMPLE  nearest / bilinear resample, source x driven by canvas x: per-lane column taps computed
//                     once per tile in fp64 (bit-identical to the oracle), rows streamed, fp32 lerp
//             GENERAL paint stack evaluated per pixel in canvas order (EXIF quarter turns, overlapping draws,
//                     translucent canvas)
//
// HBM-bound byte movement: no MFMA.
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (the fp64 coordinate math must not be fused).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "ist_internal.h"
#include "ist_launch.h"

namespace ist {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));   // RGBA rows are only pixel (4-byte) aligned in general
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));

#define IST_DEV static __device__ __forceinline__


IST_DEV u32x4 ld16(const uint8_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4*>(p)); }
IST_DEV void st16(uint8_t* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4_a4*>(p)); }
IST_DEV u32x4 ld16_plain(const uint8_t* p) { return *reinterpret_cast<const u32x4_a4*>(p); }
IST_DEV void st16_plain(uint8_t* p, u32x4 v) { *reinterpret_cast<u32x4_a4*>(p) = v; }
IST_DEV uint32_t ld4(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
IST_DEV u32x2 ld8(const uint8_t* p) { return *reinterpret_cast<const u32x2_a4*>(p); }
IST_DEV void st4(uint8_t* p, uint32_t v) { __builtin_nontemporal_store(v, reinterpret_cast<uint32_t*>(p)); }

IST_DEV uint32_t ch(uint32_t px, int c) { return (px >> (8 * c)) & 0xFFu; }

// integer source-over of one straight-alpha pixel onto a premultiplied destination (exact, no ties: 255 is odd)
IST_DEV uint32_t over_int(uint32_t s, uint32_t d) {
  const uint32_t a = s >> 24;
  if (a == 255u) return s;
  const uint32_t ia = 255u - a;
  const uint32_t r = (ch(s, 0) * a + ch(d, 0) * ia + 127u) / 255u;
  const uint32_t g = (ch(s, 1) * a + ch(d, 1) * ia + 127u) / 255u;
  const uint32_t b = (ch(s, 2) * a + ch(d, 2) * ia + 127u) / 255u;
  const uint32_t o = (255u * a + (d >> 24) * ia + 127u) / 255u;
  return r | (g << 8) | (b << 16) | (o << 24);
}

IST_DEV float lerpf(float a, float b, float t) { return __fmaf_rn(t, b - a, a); }

// round half up to a byte.  Every caller's v lies in [0, 255] (blends of bytes with weights in [0, 1]), where
// floor(v + 0.5) clamped to [0, 255] — what the oracle writes — equals the truncating, saturating conversion of v + 0.5
IST_DEV uint32_t to_u8(float v) { return static_cast<uint32_t>(v + 0.5f); }

// Four opaque pixels of one lane at once, two channels per instruction: the packed fp32 ALU (v_pk_add_f32 /
// v_pk_fma_f32) does the same IEEE operations in the same order as lerpf on each half, so the bytes do not change.
typedef float f32x2 __attribute__((ext_vector_type(2)));
IST_DEV f32x2 lerp2(f32x2 a, f32x2 b, f32x2 t) { return __builtin_elementwise_fma(t, b - a, a); }
// (the empty asm keeps the compiler from rewriting float(b) - float(a) as float(b - a): exact either way, but the
// integer form needs a scalar subtract + a second conversion per channel where the float form is half a v_pk_add_f32)
IST_DEV f32x2 chan2(uint32_t p, uint32_t q, int c) { f32x2 r; r.x = static_cast<float>(ch(p, c)); r.y = static_cast<float>(ch(q, c)); asm("" : "+v"(r)); return r; }
IST_DEV f32x2 rg(uint32_t p) { f32x2 r; r.x = static_cast<float>(ch(p, 0)); r.y = static_cast<float>(ch(p, 1)); asm("" : "+v"(r)); return r; }
IST_DEV void bilerp4_opaque(const uint32_t p00[4], const uint32_t p01[4], const uint32_t p10[4], const uint32_t p11[4],
                            const float tx[4], float ty, uint32_t o[4]) {
  const f32x2 ty2 = {ty, ty}, half = {0.5f, 0.5f};
#pragma unroll
  for (int p = 0; p < 4; ++p) {                   // red + green of pixel p
    const f32x2 t = {tx[p], tx[p]};
    const f32x2 v = lerp2(lerp2(rg(p00[p]), rg(p01[p]), t), lerp2(rg(p10[p]), rg(p11[p]), t), ty2) + half;
    o[p] = 0xFF000000u | static_cast<uint32_t>(v.x) | (static_cast<uint32_t>(v.y) << 8);
  }
#pragma unroll
  for (int p = 0; p < 4; p += 2) {                // blue of pixels p and p + 1
    const f32x2 t = {tx[p], tx[p + 1]};
    const f32x2 v = lerp2(lerp2(chan2(p00[p], p00[p + 1], 2), chan2(p01[p], p01[p + 1], 2), t),
                          lerp2(chan2(p10[p], p10[p + 1], 2), chan2(p11[p], p11[p + 1], 2), t), ty2) + half;
    o[p] |= static_cast<uint32_t>(v.x) << 16;
    o[p + 1] |= static_cast<uint32_t>(v.y) << 16;
  }
}

// the same for the narrower SAMPLE_LDS tiles (NP = 1 or 2 pixels per lane and row): identical IEEE operations per channel
template <int NP>
IST_DEV void bilerpN_opaque(const uint32_t p00[NP], const uint32_t p01[NP], const uint32_t p10[NP], const uint32_t p11[NP],
                            const float tx[NP], float ty, uint32_t o[NP]) {
  if constexpr (NP == 4) { bilerp4_opaque(p00, p01, p10, p11, tx, ty, o); return; }
  const f32x2 ty2 = {ty, ty}, half = {0.5f, 0.5f};
#pragma unroll
  for (int p = 0; p < NP; ++p) {                  // red + green of pixel p
    const f32x2 t = {tx[p], tx[p]};
    const f32x2 v = lerp2(lerp2(rg(p00[p]), rg(p01[p]), t), lerp2(rg(p10[p]), rg(p11[p]), t), ty2) + half;
    o[p] = 0xFF000000u | static_cast<uint32_t>(v.x) | (static_cast<uint32_t>(v.y) << 8);
  }
  if constexpr (NP == 2) {                        // blue of both pixels in one packed chain
    const f32x2 t = {tx[0], tx[1]};
    const f32x2 v = lerp2(lerp2(chan2(p00[0], p00[1], 2), chan2(p01[0], p01[1], 2), t),
                          lerp2(chan2(p10[0], p10[1], 2), chan2(p11[0], p11[1], 2), t), ty2) + half;
    o[0] |= static_cast<uint32_t>(v.x) << 16;
    o[1] |= static_cast<uint32_t>(v.y) << 16;
  } else {
    const float top = lerpf(static_cast<float>(ch(p00[0], 2)), static_cast<float>(ch(p01[0], 2)), tx[0]);
    const float bot = lerpf(static_cast<float>(ch(p10[0], 2)), static_cast<float>(ch(p11[0], 2)), tx[0]);
    o[0] |= to_u8(lerpf(top, bot, ty)) << 16;
  }
}

// bilinear blend of four straight-alpha taps, composited over a premultiplied destination pixel
// `opaque` (wave-uniform, the caller's hint for JPEG-decoded bitmaps) skips the per-pixel alpha test
IST_DEV uint32_t bilerp_over(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11, float tx, float ty, uint32_t d, bool opaque = false) {
  if (opaque || ((p00 & p01 & p10 & p11) >> 24) == 255u) {   // all taps opaque: plain bilinear, result replaces the destination
    const f32x2 t2 = {tx, tx}, ty2 = {ty, ty}, half = {0.5f, 0.5f};
    const f32x2 v = lerp2(lerp2(rg(p00), rg(p01), t2), lerp2(rg(p10), rg(p11), t2), ty2) + half;       // red, green
    const float top = lerpf(static_cast<float>(ch(p00, 2)), static_cast<float>(ch(p01, 2)), tx);
    const float bot = lerpf(static_cast<float>(ch(p10, 2)), static_cast<float>(ch(p11, 2)), tx);
    return 0xFF000000u | static_cast<uint32_t>(v.x) | (static_cast<uint32_t>(v.y) << 8) | (to_u8(lerpf(top, bot, ty)) << 16);
  }
  const float a00 = static_cast<float>(p00 >> 24), a01 = static_cast<float>(p01 >> 24);
  const float a10 = static_cast<float>(p10 >> 24), a11 = static_cast<float>(p11 >> 24);
  const float A = lerpf(lerpf(a00, a01, tx), lerpf(a10, a11, tx), ty);
  const float keep = 1.0f - A * (1.0f / 255.0f);
  uint32_t o = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float top = lerpf(static_cast<float>(ch(p00, c)) * a00, static_cast<float>(ch(p01, c)) * a01, tx);
    const float bot = lerpf(static_cast<float>(ch(p10, c)) * a10, static_cast<float>(ch(p11, c)) * a11, tx);
    const float P = lerpf(top, bot, ty) * (1.0f / 255.0f);
    o |= to_u8(P + static_cast<float>(ch(d, c)) * keep) << (8 * c);
  }
  o |= to_u8(A + static_cast<float>(d >> 24) * keep) << 24;
  return o;
}

// one axis of the sampling map for the bilinear filter: first tap index (clamped so that base+1 is readable when
// the axis has >= 2 samples) and the weight of the second tap
struct Tap { int32_t base; float t; };
IST_DEV Tap bilinear_tap(double k, double o, int w, int lo, int hi) {
  const double f = (k * (static_cast<double>(w) + 0.5) + o) - 0.5;
  double fl = floor(f);
  float t = static_cast<float>(f - fl);
  fl = fmin(fmax(fl, -2.0e9), 2.0e9);
  const int i0 = static_cast<int>(fl);
  Tap r;
  const int top = hi > lo ? hi - 1 : lo;
  r.base = min(max(i0, lo), top);
  if (i0 < lo) t = 0.0f;         // both taps clamp to lo
  if (i0 >= hi) t = 1.0f;        // both taps clamp to hi (= base+1 when hi > lo)
  r.t = t;
  return r;
}
IST_DEV int nearest_tap(double k, double o, int w, int lo, int hi) {
  double fl = floor(k * (static_cast<double>(w) + 0.5) + o);
  fl = fmin(fmax(fl, -2.0e9), 2.0e9);
  return min(max(static_cast<int>(fl), lo), hi);
}

// ragged right edge of a cell: the last lane of a row segment holds 1..3 pixels
IST_DEV u32x4 ld_px(const uint8_t* p, int nv) {
  if (nv >= 4) return ld16(p);
  u32x4 v = {0u, 0u, 0u, 0u};
  v.x = ld4(p);
  if (nv > 1) v.y = ld4(p + 4);
  if (nv > 2) v.z = ld4(p + 8);
  return v;
}
IST_DEV void st_px(uint8_t* p, u32x4 v, int nv) {
  if (nv >= 4) { st16(p, v); return; }
  st4(p, v.x);
  if (nv > 1) st4(p + 4, v.y);
  if (nv > 2) st4(p + 8, v.z);
}

// A tile is walked as ITEMS: one item = one wave-row = 64 lanes x 4 px = 1 KiB of one canvas row.  Items are numbered
// along the row first (tile width = 256 << lg px), so a wave's U consecutive items are contiguous bytes when lg > 0.
// Everything but the lane offset is wave-uniform (scalar registers).
#define IST_ITEM(k)                                             \
  const int row_ = (k) >> lg, chunk_ = (k) & ((1 << lg) - 1);   \
  const int X = X0 + (chunk_ << 8) + lane4, Y = Y0 + row_;      \
  const int nv = X1 - X

// ------------------------------------------------------------------------------------------------ FILL
IST_DEV void tile_fill(const LaunchArgs& A, uint32_t colour, int lg, int X0, int Y0, int X1, int Y1) {
  const int lane4 = (threadIdx.x & 63) * 4;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int items = (Y1 - Y0) << lg;
  const u32x4 v = {colour, colour, colour, colour};
  for (int k = wave; k < items; k += 4) {
    IST_ITEM(k);
    if (nv > 0) st_px(A.dst + static_cast<size_t>(Y) * A.dst_pitch + static_cast<size_t>(X) * 4, v, nv);
  }
}

// ------------------------------------------------------------------------------------------------ COPY
// U = wave-rows in flight per wave; IL = consecutive items go to different waves (true) or to one wave (false);
// NTL / NTS = non-temporal hint on loads / stores.
template <int U, bool IL, bool NTL, bool NTS>
IST_DEV void tile_copy(const LaunchArgs& A, const DevOp op, uint32_t bg, int lg, int X0, int Y0, int X1, int Y1) {
  const int lane4 = (threadIdx.x & 63) * 4;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int items = (Y1 - Y0) << lg;
  const int64_t sp = static_cast<int64_t>(A.pitch[op.image]);
  // unit-scale map: ix = X + ox, or ox - 1 - X when mirrored (EXIF 2/3/4 at 1:1); rows likewise
  const bool fx = (op.flags & OPF_FLIPX) != 0, fy = (op.flags & OPF_FLIPY) != 0;
  const int64_t bx = static_cast<int64_t>(op.ox) - (fx ? 1 : 0), by = static_cast<int64_t>(op.oy) - (fy ? 1 : 0);
  const uint8_t* s = A.src[op.image];
  const bool opaque = (op.flags & OPF_OPAQUE) != 0;
  for (int j = 0; j * 4 * U < items; ++j) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = IL ? ((j * U + u) * 4 + wave) : ((j * 4 + wave) * U + u);
      IST_ITEM(k);
      if (k < items && nv > 0) {
        const int64_t iy = fy ? by - Y : by + Y;
        if (!fx) {
          const uint8_t* p = s + iy * sp + (bx + X) * 4;
          v[u] = nv >= 4 ? (NTL ? ld16(p) : ld16_plain(p)) : ld_px(p, nv);
        } else if (nv >= 4) {           // mirrored: canvas X..X+3 <- source bx-X-3 .. bx-X, reversed in registers
          const u32x4 t = ld16(s + iy * sp + (bx - X - 3) * 4);
          v[u].x = t.w; v[u].y = t.z; v[u].z = t.y; v[u].w = t.x;
        } else {
          v[u].x = ld4(s + iy * sp + (bx - X) * 4);
          v[u].y = nv > 1 ? ld4(s + iy * sp + (bx - X - 1) * 4) : 0u;
          v[u].z = nv > 2 ? ld4(s + iy * sp + (bx - X - 2) * 4) : 0u;
          v[u].w = 0u;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = IL ? ((j * U + u) * 4 + wave) : ((j * 4 + wave) * U + u);
      IST_ITEM(k);
      if (k < items && nv > 0) {
        u32x4 o = v[u];
        if (!opaque && ((o.x & o.y & o.z & o.w) >> 24) != 255u) {
          o.x = over_int(o.x, bg); o.y = over_int(o.y, bg); o.z = over_int(o.z, bg); o.w = over_int(o.w, bg);
        }
        uint8_t* q = A.dst + static_cast<size_t>(Y) * A.dst_pitch + static_cast<size_t>(X) * 4;
        if (nv >= 4) { if (NTS) st16(q, o); else st16_plain(q, o); }
        else st_px(q, o, nv);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ SAMPLE
template <int FILTER>
IST_DEV void tile_sample(const LaunchArgs& A, const DevOp op, uint32_t bg, int X0, int Y0, int X1, int Y1) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int X = X0 + lane * 4;
  const int nv = X1 - X;
  if (nv <= 0) return;
  const size_t sp = A.pitch[op.image];
  const uint8_t* src = A.src[op.image];
  uint8_t* d = A.dst + static_cast<size_t>(X) * 4;

  if (FILTER == IST_FILTER_NEAREST) {
    int ix[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) ix[p] = nearest_tap(op.kx, op.ox, X + p, op.cx0, op.cx1);
    for (int Y = Y0 + wave; Y < Y1; Y += 4) {
      const int iy = nearest_tap(op.ky, op.oy, Y, op.cy0, op.cy1);
      const uint8_t* row = src + static_cast<size_t>(iy) * sp;
      u32x4 o;
      o.x = over_int(ld4(row + 4 * static_cast<size_t>(ix[0])), bg);
      o.y = over_int(ld4(row + 4 * static_cast<size_t>(ix[1])), bg);
      o.z = over_int(ld4(row + 4 * static_cast<size_t>(ix[2])), bg);
      o.w = over_int(ld4(row + 4 * static_cast<size_t>(ix[3])), bg);
      uint8_t* dp = d + static_cast<size_t>(Y) * A.dst_pitch;
      if (nv >= 4) st16(dp, o);
      else { st4(dp, o.x); if (nv > 1) st4(dp + 4, o.y); if (nv > 2) st4(dp + 8, o.z); }
    }
    return;
  }

  Tap tx[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) tx[p] = bilinear_tap(op.kx, op.ox, X + p, op.cx0, op.cx1);
  const bool pair_x = op.cx1 > op.cx0;                 // a 1-pixel-wide source has no second column
  const size_t row_step = op.cy1 > op.cy0 ? sp : 0;    // a 1-pixel-high source has no second row
  for (int Y = Y0 + wave; Y < Y1; Y += 4) {
    const Tap ty = bilinear_tap(op.ky, op.oy, Y, op.cy0, op.cy1);
    const uint8_t* r0 = src + static_cast<size_t>(ty.base) * sp;
    const uint8_t* r1 = r0 + row_step;
    uint32_t o[4], p00[4], p01[4], p10[4], p11[4], all = 0xFFFFFFFFu;
    float wx[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const size_t off = 4 * static_cast<size_t>(tx[p].base);
      if (pair_x) {
        const u32x2 a = ld8(r0 + off), b = ld8(r1 + off);
        p00[p] = a.x; p01[p] = a.y; p10[p] = b.x; p11[p] = b.y;
      } else {
        p00[p] = p01[p] = ld4(r0 + off); p10[p] = p11[p] = ld4(r1 + off);
      }
      wx[p] = tx[p].t;
      all &= p00[p] & p01[p] & p10[p] & p11[p];
    }
    if ((op.flags & OPF_OPAQUE) || (all >> 24) == 255u) bilerp4_opaque(p00, p01, p10, p11, wx, ty.t, o);
    else {
#pragma unroll
      for (int p = 0; p < 4; ++p) o[p] = bilerp_over(p00[p], p01[p], p10[p], p11[p], wx[p], ty.t, bg, false);
    }
    uint8_t* dp = d + static_cast<size_t>(Y) * A.dst_pitch;
    if (nv >= 4) { const u32x4 v = {o[0], o[1], o[2], o[3]}; st16(dp, v); }
    else { st4(dp, o[0]); if (nv > 1) st4(dp + 4, o[1]); if (nv > 2) st4(dp + 8, o[2]); }
  }
}

// The taps of the tile's rows are wave-uniform: lane i computes the tap of row R0+i ONCE, the row loop then picks its
// row's tap with v_readlane (tiles are at most 64 rows tall).
struct RowTaps { int base; float t; };
IST_DEV RowTaps row_taps(double k, double o, int R0, int R1, int lo, int hi) {
  const int lane = threadIdx.x & 63;
  const Tap t = bilinear_tap(k, o, min(R0 + lane, R1 - 1), lo, hi);
  RowTaps r; r.base = t.base; r.t = t.t;
  // pin the values here, while every lane of the wave is active: without this the compiler may sink the computation
  // below a later divergent branch, and v_readlane would then read lanes that never computed it
  asm volatile("" : "+v"(r.base), "+v"(r.t));
  return r;
}
IST_DEV Tap row_tap(const RowTaps& r, int j) {
  Tap t;
  t.base = __builtin_amdgcn_readlane(r.base, j);
  t.t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.t), j));
  return t;
}

// ------------------------------------------------------------------------------------------------ SAMPLE via LDS
// Bilinear resample with the tile's source footprint staged in LDS: every needed source byte crosses the vector
// memory path ONCE, as coalesced 16-B loads (the direct path above issues 8-byte gathers whose lanes straddle ~3x
// as many cache lines), and the 16 taps per lane then come from LDS (ds_read2_b32).  The host sizes tile_h so that
// the footprint fits kLdsWords; the kernel re-checks and falls back to the direct path if it ever does not.
// The footprint buffer is dynamic LDS sized by the host (LaunchArgs.lds_half words = the largest footprint any stage
// of any cell needs).

// NP = pixels per lane and row: the tile is 64 * NP pixels wide.  The host picks the width per cell (ist_compile.cpp): a
// strong shrink has a tall footprint per output row, and a narrower, taller tile then carries more output pixels per
// 24 KiB of footprint (kx = ky = 1.87: 128 x 12 instead of 256 x 4) and re-reads fewer halo rows.
// fresh: the workgroup has not touched LDS yet (one tile per workgroup), so the first stage needs no leading barrier.
template <int NP>
IST_DEV void tile_sample_lds(const LaunchArgs& A, const DevOp op, uint32_t bg, int X0, int Y0, int X1, int Y1, int sub_h, uint32_t* lds, bool fresh) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  // The tile is a column of stages of sub_h rows, each loaded (LDS-DMA, no VGPRs) and then resampled: the per-tile
  // set-up (tile look-up, op fetch, the fp64 column taps) is paid once per column instead of once per stage.
  // Measured on MI355X (tools/exp_mixed.py): 2 stages per workgroup beat 1 by ~4 %; more stages leave too few
  // workgroups for the tail; a double-buffered variant (stage s+1 loading under stage s's arithmetic) lost 10-20 %
  // because the second buffer halves the workgroups per CU.
  // x footprint (wave-uniform, shared by all stages): the taps are monotonic in X, so the tile's corners bound it
  const Tap xa = bilinear_tap(op.kx, op.ox, X0, op.cx0, op.cx1), xb = bilinear_tap(op.kx, op.ox, X1 - 1, op.cx0, op.cx1);
  const int fx0 = __builtin_amdgcn_readfirstlane(min(xa.base, xb.base)), fx1 = __builtin_amdgcn_readfirstlane(max(xa.base, xb.base) + 1);
  const int wl = (fx1 - fx0 + 4) & ~3;            // LDS row stride in pixels (multiple of 4: 16-B aligned rows)
  const int nsub = (Y1 - Y0 + sub_h - 1) / sub_h;
  // y footprint of stage s: first source row + row count (wave-uniform)
  auto foot = [&](int s, int* fy0, int* fh) {
    const int Ya = Y0 + s * sub_h, Yb = min(Ya + sub_h, Y1);
    const Tap ya = bilinear_tap(op.ky, op.oy, Ya, op.cy0, op.cy1), yb = bilinear_tap(op.ky, op.oy, Yb - 1, op.cy0, op.cy1);
    const int lo = __builtin_amdgcn_readfirstlane(min(ya.base, yb.base)), hi = __builtin_amdgcn_readfirstlane(max(ya.base, yb.base) + 1);
    *fy0 = lo; *fh = hi - lo + 1;
  };
  bool fits = sub_h > 0 && op.cx1 > op.cx0 && op.cy1 > op.cy0;
  for (int s = 0; s < nsub && fits; ++s) { int y, h; foot(s, &y, &h); fits = wl * h <= A.lds_half; }
  if (!fits) {                                    // uniform; not expected (the host sizes the stages)
    tile_sample<IST_FILTER_BILINEAR>(A, op, bg, X0, Y0, X1, Y1);
    return;
  }
  const size_t sp = A.pitch[op.image];
  const uint8_t* src = A.src[op.image];
  // ---- stage: wave w takes footprint rows w, w+4, ...; a pass moves 64 lanes x 16 B = 256 px of one row.
  // LDS-DMA (global_load_lds_dwordx4): per-lane global address, LDS destination = uniform base + lane*16, no VGPR
  // staging, so every pass of the wave is in flight at once.
  const int chunks = wl >> 2;
  auto stage = [&](int fy0, int fh, uint32_t* buf) {
    for (int r = wave; r < fh; r += 4) {
      const uint8_t* grow = src + static_cast<size_t>(fy0 + r) * sp;
      uint32_t* lrow = buf + r * wl;
      // reading up to 12 B past the last sampled column is harmless (next row of the same bitmap) except on the last
      // sampled row, where it could leave the allocation: that row's edge pass goes through registers instead
      const bool last_row = (fy0 + r) >= op.cy1;
      for (int c0 = 0; c0 < chunks; c0 += 64) {
        const int c = c0 + lane;
        const int col = fx0 + 4 * c;
        const bool edge = last_row && (fx0 + 4 * min(c0 + 63, chunks - 1) + 3 > op.cx1);    // wave-uniform
        if (!edge) {
          if (c < chunks)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) uint32_t*)(grow + static_cast<size_t>(col) * 4),
                                             (__attribute__((address_space(3))) uint32_t*)(lrow + 4 * c0), 16, 0, 0);
        } else if (c < chunks) {
          u32x4 v;
          if (col + 3 <= op.cx1) v = ld16(grow + static_cast<size_t>(col) * 4);
          else {
            v.x = ld4(grow + static_cast<size_t>(min(col, op.cx1)) * 4);
            v.y = ld4(grow + static_cast<size_t>(min(col + 1, op.cx1)) * 4);
            v.z = ld4(grow + static_cast<size_t>(min(col + 2, op.cx1)) * 4);
            v.w = ld4(grow + static_cast<size_t>(min(col + 3, op.cx1)) * 4);
          }
          *reinterpret_cast<u32x4*>(lrow + 4 * c) = v;
        }
      }
    }
  };
  // The first stage's loads go out BEFORE the per-lane set-up below (four fp64 column taps per lane, ~0.4 us): a
  // workgroup that has just started then already has its footprint in flight while it computes them.
  int fy0, fh;
  foot(0, &fy0, &fh);
  if (!fresh) __syncthreads();                            // every wave is done reading the previous tile's footprint
  stage(fy0, fh, lds);
  // Lane l owns pixels X0 + l + 64 p (p = 0..3), NOT 4 neighbours: consecutive lanes then read LDS words |kx| apart
  // instead of 4|kx| apart (measured: 77 % of the LDS cycles were bank conflicts with the neighbour mapping), and each
  // of the 4 stores of a wave is still 256 contiguous bytes.  Lanes past a ragged right edge keep computing (on the
  // last column) and only skip their stores: the row taps below are exchanged with v_readlane and the barriers of
  // later stages need every wave.
  int Xl = X0 + lane;
  asm volatile("" : "+v"(Xl));                            // (keeps the tap arithmetic below the loads issued above)
  int lx[NP]; float wx[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const Tap t = bilinear_tap(op.kx, op.ox, min(Xl + 64 * p, X1 - 1), op.cx0, op.cx1);
    lx[p] = t.base - fx0; wx[p] = t.t;
  }
  uint8_t* d = A.dst + static_cast<size_t>(Xl) * 4;
  const bool opaque = (op.flags & OPF_OPAQUE) != 0;
  // one stage's arithmetic: LDS taps -> blend -> 256-B-per-wave stores
  auto compute = [&](int s, int sfy0, const RowTaps& rows, const uint32_t* buf) {
    const int Ya = Y0 + s * sub_h, Yb = min(Ya + sub_h, Y1);
    for (int Y = Ya + wave; Y < Yb; Y += 4) {
      const Tap ty = row_tap(rows, Y - Ya);
      const uint32_t* r0 = buf + (ty.base - sfy0) * wl;
      const uint32_t* r1 = r0 + wl;
      uint32_t o[NP];
      uint32_t p00[NP], p01[NP], p10[NP], p11[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) { p00[p] = r0[lx[p]]; p01[p] = r0[lx[p] + 1]; p10[p] = r1[lx[p]]; p11[p] = r1[lx[p] + 1]; }
      uint32_t all = 0xFFFFFFFFu;
#pragma unroll
      for (int p = 0; p < NP; ++p) all &= p00[p] & p01[p] & p10[p] & p11[p];
      if (opaque || (all >> 24) == 255u) bilerpN_opaque<NP>(p00, p01, p10, p11, wx, ty.t, o);
      else {
#pragma unroll
        for (int p = 0; p < NP; ++p) o[p] = bilerp_over(p00[p], p01[p], p10[p], p11[p], wx[p], ty.t, bg, false);
      }
      uint8_t* dp = d + static_cast<size_t>(Y) * A.dst_pitch;
#pragma unroll
      for (int p = 0; p < NP; ++p)
        if (Xl + 64 * p < X1) st4(dp + 256 * p, o[p]);
    }
  };
  for (int s = 0; s < nsub; ++s) {
    if (s > 0) {
      __syncthreads();                                    // every wave is done reading the previous stage
      stage(fy0, fh, lds);
    }
    int ny0 = 0, nh = 0;
    if (s + 1 < nsub) foot(s + 1, &ny0, &nh);             // (tap arithmetic under the loads)
    const int Ya = Y0 + s * sub_h, Yb = min(Ya + sub_h, Y1);
    const RowTaps rows = row_taps(op.ky, op.oy, Ya, Yb, op.cy0, op.cy1);   // all 64 lanes active (readlane source)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LDS-DMA is tracked by vmcnt only
    __syncthreads();
    compute(s, fy0, rows, lds);
    fy0 = ny0; fh = nh;
  }
}

// ------------------------------------------------------------------------------------------------ SAMPLE, streamed
// The same arithmetic as tile_sample_lds, with NO workgroup barrier: every wave owns output rows Y0 + wave + 4 j and a
// private LDS ring of DEPTH row pairs.  It keeps the two source rows of its next DEPTH-1 output rows in flight (LDS-DMA)
// while it blends the current one, and waits with a COUNTED s_waitcnt (vector memory operations retire in issue order,
// loads and stores together), so loads, LDS reads, arithmetic and stores of one wave overlap instead of alternating
// between a load phase and a compute phase of the whole workgroup.  Adjacent output rows belong to different waves of
// the workgroup and fetch their shared source row at about the same time: the second request is an L2 hit.
// ORDERING RULE RELIED ON (MI355X_MICROARCH.md, "Per-instruction cycle constants"): "s_waitcnt vmcnt(N) waits until all but the
// wave's N youngest vector-memory operations are done.  Loads, stores, atomics and LDS-DMA count together, in issue order
// (flat_* excepted: out of order)".  Every access below is a global_* instruction (address-space-1 pointers / kernel
// arguments), never flat_*.
IST_DEV void wait_vm_upto(int n) {                 // wave-uniform n; waits until at most min(n, 24) operations remain
#define IST_WAIT_CASE(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
  switch (n) {
    IST_WAIT_CASE(0) IST_WAIT_CASE(1) IST_WAIT_CASE(2) IST_WAIT_CASE(3) IST_WAIT_CASE(4) IST_WAIT_CASE(5)
    IST_WAIT_CASE(6) IST_WAIT_CASE(7) IST_WAIT_CASE(8) IST_WAIT_CASE(9) IST_WAIT_CASE(10) IST_WAIT_CASE(11)
    IST_WAIT_CASE(12) IST_WAIT_CASE(13) IST_WAIT_CASE(14) IST_WAIT_CASE(15) IST_WAIT_CASE(16) IST_WAIT_CASE(17)
    IST_WAIT_CASE(18) IST_WAIT_CASE(19) IST_WAIT_CASE(20) IST_WAIT_CASE(21) IST_WAIT_CASE(22) IST_WAIT_CASE(23)
    default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
  }
#undef IST_WAIT_CASE
}

template <int NP>
IST_DEV void tile_sample_stream(const LaunchArgs& A, const DevOp op, uint32_t bg, int X0, int Y0, int X1, int Y1, int depth, uint32_t* lds, bool fresh) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const Tap xa = bilinear_tap(op.kx, op.ox, X0, op.cx0, op.cx1), xb = bilinear_tap(op.kx, op.ox, X1 - 1, op.cx0, op.cx1);
  const int fx0 = __builtin_amdgcn_readfirstlane(min(xa.base, xb.base)), fx1 = __builtin_amdgcn_readfirstlane(max(xa.base, xb.base) + 1);
  const int wl = (fx1 - fx0 + 4) & ~3;            // LDS row stride in pixels (multiple of 4: 16-B aligned rows)
  if (depth < 2 || Y1 - Y0 > 64 || op.cx1 <= op.cx0 || op.cy1 <= op.cy0 || 8 * depth * wl > A.lds_words) {   // uniform; not expected
    tile_sample<IST_FILTER_BILINEAR>(A, op, bg, X0, Y0, X1, Y1);
    return;
  }
  const size_t sp = A.pitch[op.image];
  const uint8_t* src = A.src[op.image];
  const RowTaps rows = row_taps(op.ky, op.oy, Y0, Y1, op.cy0, op.cy1);   // all 64 lanes active (readlane source)
  const int nrows = (Y1 - Y0 - wave + 3) >> 2;    // output rows of this wave
  uint32_t* ring = lds + wave * (2 * depth * wl);
  if (!fresh) __syncthreads();                    // (grid-stride form) every wave is done with the previous tile's LDS
  const int chunks = wl >> 2;
  const int ni = (chunks + 63) >> 6;              // DMA instructions per source row
  const int spr = min(NP, (X1 - X0 + 63) >> 6);   // store instructions per output row
  bool counted = true;                            // false once a row went through registers (its operation count differs)
  // the two source rows of output row j -> ring slot `slot`
  auto issue = [&](int j, int slot) {
    const Tap ty = row_tap(rows, wave + 4 * j);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int sy = ty.base + h;
      const uint8_t* grow = src + static_cast<size_t>(sy) * sp;
      uint32_t* lrow = ring + (2 * slot + h) * wl;
      // reading up to 12 B past the last sampled column is harmless (next row of the same bitmap) except on the last
      // sampled row, where it could leave the allocation: that row's edge pass goes through registers instead
      const bool last_row = sy >= op.cy1;
      for (int c0 = 0; c0 < chunks; c0 += 64) {
        const int c = c0 + lane;
        const int col = fx0 + 4 * c;
        const bool edge = last_row && (fx0 + 4 * min(c0 + 63, chunks - 1) + 3 > op.cx1);    // wave-uniform
        if (!edge) {
          if (c < chunks)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) uint32_t*)(grow + static_cast<size_t>(col) * 4),
                                             (__attribute__((address_space(3))) uint32_t*)(lrow + 4 * c0), 16, 0, 0);
        } else {
          counted = false;
          if (c < chunks) {
            u32x4 v;
            if (col + 3 <= op.cx1) v = ld16(grow + static_cast<size_t>(col) * 4);
            else {
              v.x = ld4(grow + static_cast<size_t>(min(col, op.cx1)) * 4);
              v.y = ld4(grow + static_cast<size_t>(min(col + 1, op.cx1)) * 4);
              v.z = ld4(grow + static_cast<size_t>(min(col + 2, op.cx1)) * 4);
              v.w = ld4(grow + static_cast<size_t>(min(col + 3, op.cx1)) * 4);
            }
            *reinterpret_cast<u32x4*>(lrow + 4 * c) = v;
          }
        }
      }
    }
  };
  // prologue: the first depth-1 row pairs go out BEFORE the per-lane set-up (four fp64 column taps per lane)
  const int ahead = depth - 1;
  for (int j = 0; j < min(ahead, nrows); ++j) issue(j, j);
  int Xl = X0 + lane;
  asm volatile("" : "+v"(Xl));                            // (keeps the tap arithmetic below the loads issued above)
  int lx[NP]; float wx[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const Tap t = bilinear_tap(op.kx, op.ox, min(Xl + 64 * p, X1 - 1), op.cx0, op.cx1);
    lx[p] = t.base - fx0; wx[p] = t.t;
  }
  uint8_t* d = A.dst + static_cast<size_t>(Xl) * 4;
  const bool opaque = (op.flags & OPF_OPAQUE) != 0;
  int slot = 0, nslot = ahead;                            // ring slot of row j / of row j + depth - 1
  for (int j = 0; j < nrows; ++j) {
    // operations younger than row j's loads: the loads of the rows ahead of it and the stores of the rows since its issue
    if (j + ahead < nrows) {
      issue(j + ahead, nslot);
      wait_vm_upto(counted ? ahead * 2 * ni + min(j, ahead) * spr : 0);
    } else {
      wait_vm_upto(counted ? (nrows - 1 - j) * 2 * ni : 0);
    }
    const int Y = Y0 + wave + 4 * j;
    const Tap ty = row_tap(rows, wave + 4 * j);
    const uint32_t* r0 = ring + 2 * slot * wl;
    const uint32_t* r1 = r0 + wl;
    uint32_t o[NP];
    uint32_t p00[NP], p01[NP], p10[NP], p11[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) { p00[p] = r0[lx[p]]; p01[p] = r0[lx[p] + 1]; p10[p] = r1[lx[p]]; p11[p] = r1[lx[p] + 1]; }
    uint32_t all = 0xFFFFFFFFu;
#pragma unroll
    for (int p = 0; p < NP; ++p) all &= p00[p] & p01[p] & p10[p] & p11[p];
    if (opaque || (all >> 24) == 255u) bilerpN_opaque<NP>(p00, p01, p10, p11, wx, ty.t, o);
    else {
#pragma unroll
      for (int p = 0; p < NP; ++p) o[p] = bilerp_over(p00[p], p01[p], p10[p], p11[p], wx[p], ty.t, bg, false);
    }
    uint8_t* dp = d + static_cast<size_t>(Y) * A.dst_pitch;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      if (Xl + 64 * p < X1) st4(dp + 256 * p, o[p]);
    slot = slot + 1 == depth ? 0 : slot + 1;
    nslot = nslot + 1 == depth ? 0 : nslot + 1;
  }
}

// ------------------------------------------------------------------------------------------------ AREA, streamed
// IST_FILTER_AREA on an axis-aligned draw that shrinks: the sample of a canvas pixel is the mean of the source over the
// pixel's footprint, a box of max(1, |kx|) x max(1, |ky|) source pixels, every source pixel weighted by its overlap (at
// |k| <= 1 the box is the bilinear pair).  Unlike point-sampled bilinear, this rule needs EVERY source byte, so the bytes
// the memory system must move anyway (whole 32-byte sectors of the rows a shrink touches) are all useful work.
// Separable, no workgroup barrier: a wave owns output rows Y0 + wave + 4 j.  For one output row it
//   1. reads the <= ceil(|ky|) + 1 source rows of the row's box with coalesced 16-byte loads (lane = 4 neighbouring source
//      pixels; up to 8 rows in flight per lane) and sums them in registers with the rows' overlap weights (packed fp32),
//   2. writes the column sums (one float4 per source pixel) to its private LDS row,
//   3. lane = canvas pixel: sums the <= ceil(|kx|) + 1 float4 of its box from LDS with the columns' overlap weights,
//      normalises, composites over the background and stores 256 contiguous bytes per wave.
// A wave's LDS operations execute in order, so the phases need no barrier; the x taps of a lane are the same for every
// row of the tile and are computed once (fp64, as the oracle does).  The host sizes the tile width so that the x footprint
// is a whole number of 64-lane passes (ist_compile.cpp): a 256-pixel tile at 2.2x left a third of the lanes of step 1 idle.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kAreaRows = 4;             // source rows in flight per lane in step 1

template <int NP, bool OPAQUE>
IST_DEV void tile_area_stream(const LaunchArgs& A, const DevOp op, uint32_t bg, int X0, int Y0, int X1, int Y1, uint32_t* lds) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const double bw = fmax(fabs(op.kx), 1.0), bh = fmax(fabs(op.ky), 1.0);
  // x footprint of the tile (wave-uniform): the boxes of its first and last column bound it
  const double ca = op.kx * (static_cast<double>(X0) + 0.5) + op.ox, cb = op.kx * (static_cast<double>(X1 - 1) + 0.5) + op.ox;
  const double fl = fmin(ca, cb) - 0.5 * bw, fh = fmax(ca, cb) + 0.5 * bw;
  const int fx0 = __builtin_amdgcn_readfirstlane(static_cast<int>(fmin(fmax(floor(fl), -2.0e9), 2.0e9)));
  const int fx1 = __builtin_amdgcn_readfirstlane(static_cast<int>(fmin(fmax(ceil(fh), -2.0e9), 2.0e9)) - 1);
  const int wl = (fx1 - fx0 + 1 + 3) & ~3;                  // source pixels per LDS row
  const int chunks = wl >> 2;
  float* row = reinterpret_cast<float*>(lds) + static_cast<size_t>(wave) * (4 * wl);
  const size_t sp = A.pitch[op.image];
  const uint8_t* src = A.src[op.image];
  // per lane, once per tile: the box of each of its canvas pixels on the x axis
  const int Xl = X0 + lane;
  int tap[NP]; float wf[NP], wb[NP];                      // (first tap's LDS index << 8) | taps; weight of the first / last tap
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const double sxc = op.kx * (static_cast<double>(min(Xl + 64 * p, X1 - 1)) + 0.5) + op.ox;
    const double xlo = sxc - 0.5 * bw, xhi = sxc + 0.5 * bw;
    const int ix0 = static_cast<int>(fmin(fmax(floor(xlo), -2.0e9), 2.0e9)), ix1 = static_cast<int>(fmin(fmax(ceil(xhi), -2.0e9), 2.0e9)) - 1;
    tap[p] = ((ix0 - fx0) << 8) | min(ix1 - ix0 + 1, 255);
    wf[p] = static_cast<float>(fmin(static_cast<double>(ix0) + 1.0, xhi) - fmax(static_cast<double>(ix0), xlo));
    wb[p] = static_cast<float>(fmin(static_cast<double>(ix1) + 1.0, xhi) - fmax(static_cast<double>(ix1), xlo));
    __builtin_amdgcn_sched_barrier(0);                      // (one pixel's fp64 temporaries at a time)
  }
  const float normf = static_cast<float>(1.0 / (bw * bh));
  uint8_t* d = A.dst + static_cast<size_t>(Xl) * 4;
  for (int Y = Y0 + wave; Y < Y1; Y += 4) {
    // the rows of this output row's box (wave-uniform)
    const double syc = op.ky * (static_cast<double>(Y) + 0.5) + op.oy;
    const double ylo = syc - 0.5 * bh, yhi = syc + 0.5 * bh;
    const int iy0 = __builtin_amdgcn_readfirstlane(static_cast<int>(fmin(fmax(floor(ylo), -2.0e9), 2.0e9)));
    const int iy1 = __builtin_amdgcn_readfirstlane(static_cast<int>(fmin(fmax(ceil(yhi), -2.0e9), 2.0e9)) - 1);
    // 1 + 2: column sums of the box's rows -> LDS
    for (int c0 = 0; c0 < chunks; c0 += 64) {
      const int c = c0 + lane;
      const bool mine = c < chunks;
      const int xx = fx0 + 4 * min(c, chunks - 1);
      const bool inside = xx >= op.cx0 && xx + 3 <= op.cx1;
      f32x2 acc[4][2];                                      // per source pixel: (r, g), (b, a)
#pragma unroll
      for (int q = 0; q < 4; ++q) { acc[q][0] = f32x2{0.f, 0.f}; acc[q][1] = f32x2{0.f, 0.f}; }
      auto add_px = [&](int q, uint32_t px, f32x2 w2) {
        f32x2 lo = {static_cast<float>(ch(px, 0)), static_cast<float>(ch(px, 1))};
        f32x2 hi = {static_cast<float>(ch(px, 2)), static_cast<float>(px >> 24)};
        if (!OPAQUE) { const f32x2 a2 = {hi.y, hi.y}; lo = lo * a2; hi.x = hi.x * hi.y; }
        acc[q][0] = __builtin_elementwise_fma(w2, lo, acc[q][0]);
        acc[q][1] = __builtin_elementwise_fma(w2, hi, acc[q][1]);
      };
      auto weight_of = [&](int y) {                         // overlap of source row y with the box (wave-uniform)
        const float w = static_cast<float>(fmin(static_cast<double>(y) + 1.0, yhi) - fmax(static_cast<double>(y), ylo));
        const float u = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w)));
        return f32x2{u, u};
      };
      if (inside) {
        for (int yy = iy0; yy <= iy1; yy += kAreaRows) {
          u32x4 v[kAreaRows];
#pragma unroll
          for (int u = 0; u < kAreaRows; ++u) {
            if (yy + u > iy1) break;                        // (wave-uniform)
            // (plain, not non-temporal: the rows at the ends of the box are read again by the wave that owns the next output row)
            v[u] = ld16_plain(src + static_cast<size_t>(min(max(yy + u, op.cy0), op.cy1)) * sp + static_cast<size_t>(xx) * 4);
          }
#pragma unroll
          for (int u = 0; u < kAreaRows; ++u) {
            if (yy + u > iy1) break;
            const f32x2 w2 = weight_of(yy + u);
            add_px(0, v[u].x, w2); add_px(1, v[u].y, w2); add_px(2, v[u].z, w2); add_px(3, v[u].w, w2);
          }
        }
      } else {                                              // a chunk that straddles the source's edge (rare): pixel by pixel, clamped
#pragma unroll 1
        for (int y = iy0; y <= iy1; ++y) {
          const f32x2 w2 = weight_of(y);
          const uint8_t* g = src + static_cast<size_t>(min(max(y, op.cy0), op.cy1)) * sp;
          add_px(0, ld4(g + static_cast<size_t>(min(max(xx, op.cx0), op.cx1)) * 4), w2);
          __builtin_amdgcn_sched_barrier(0);
          add_px(1, ld4(g + static_cast<size_t>(min(max(xx + 1, op.cx0), op.cx1)) * 4), w2);
          __builtin_amdgcn_sched_barrier(0);
          add_px(2, ld4(g + static_cast<size_t>(min(max(xx + 2, op.cx0), op.cx1)) * 4), w2);
          __builtin_amdgcn_sched_barrier(0);
          add_px(3, ld4(g + static_cast<size_t>(min(max(xx + 3, op.cx0), op.cx1)) * 4), w2);
        }
      }
      if (mine) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t = {acc[q][0].x, acc[q][0].y, acc[q][1].x, acc[q][1].y};
          *reinterpret_cast<f32x4*>(row + 4 * (4 * c + q)) = t;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // 3: the box of every canvas pixel along x, from LDS
    uint8_t* dp = d + static_cast<size_t>(Y) * A.dst_pitch;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (X0 + 64 * p >= X1) break;                         // (wave-uniform: the tile is narrower than 64 * NP)
      const f32x4* t = reinterpret_cast<const f32x4*>(row) + (tap[p] >> 8);
      f32x4 s = t[0] * wf[p];
      const int nt = tap[p] & 255;
      for (int k = 1; k < nt - 1; ++k) s += t[k];
      if (nt > 1) s += t[nt - 1] * wb[p];
      uint32_t o;
      if (OPAQUE) {                                         // opaque source: the mean replaces the destination
        o = 0xFF000000u | to_u8(fminf(s.x * normf, 255.f)) | (to_u8(fminf(s.y * normf, 255.f)) << 8) | (to_u8(fminf(s.z * normf, 255.f)) << 16);
      } else {                                              // premultiplied mean, source-over on the background, one rounding
        const float Aa = s.w * normf;
        const float keep = 1.0f - Aa * (1.0f / 255.0f);
        o = to_u8(fminf(s.x * normf * (1.0f / 255.0f) + static_cast<float>(ch(bg, 0)) * keep, 255.f)) |
            (to_u8(fminf(s.y * normf * (1.0f / 255.0f) + static_cast<float>(ch(bg, 1)) * keep, 255.f)) << 8) |
            (to_u8(fminf(s.z * normf * (1.0f / 255.0f) + static_cast<float>(ch(bg, 2)) * keep, 255.f)) << 16) |
            (to_u8(fminf(Aa + static_cast<float>(bg >> 24) * keep, 255.f)) << 24);
      }
      if (Xl + 64 * p < X1) st4(dp + 256 * p, o);
      __builtin_amdgcn_sched_barrier(0);                    // (one pixel's taps at a time)
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------ SWAP via LDS
// One quarter-turned draw (EXIF 5-8: source x is driven by canvas Y, source y by canvas X), bilinear.  A 64 x th
// canvas tile needs a (th*|kx|+2)-column x (64*|ky|+2)-row source patch.  The patch is read row by row with coalesced
// 16-B loads and written TRANSPOSED into LDS (T[source col][source row], odd pitch -> conflict-free), so that at
// sampling time the 64 lanes of a canvas row read consecutive LDS words and the store is 256 contiguous bytes.
// Returns false (before touching LDS or any barrier, wave-uniformly) when the patch does not fit: the caller then
// renders the tile through the general path.
IST_DEV bool tile_swap_lds(const LaunchArgs& A, const DevOp op, uint32_t bg, int X0, int Y0, int X1, int Y1, uint32_t* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const Tap ca = bilinear_tap(op.kx, op.ox, Y0, op.cx0, op.cx1), cb = bilinear_tap(op.kx, op.ox, Y1 - 1, op.cx0, op.cx1);
  const Tap ra = bilinear_tap(op.ky, op.oy, X0, op.cy0, op.cy1), rb = bilinear_tap(op.ky, op.oy, X1 - 1, op.cy0, op.cy1);
  const int fx0 = __builtin_amdgcn_readfirstlane(min(ca.base, cb.base)), fx1 = __builtin_amdgcn_readfirstlane(max(ca.base, cb.base) + 1);
  const int fy0 = __builtin_amdgcn_readfirstlane(min(ra.base, rb.base)), fy1 = __builtin_amdgcn_readfirstlane(max(ra.base, rb.base) + 1);
  const int fw = fx1 - fx0 + 1, fh = fy1 - fy0 + 1;
  const int pitch = fh | 1;
  if (fw * pitch > A.lds_words || op.cx1 <= op.cx0 || op.cy1 <= op.cy0) return false;
  const size_t sp = A.pitch[op.image];
  const uint8_t* src = A.src[op.image];
  __syncthreads();
  // ---- stage transposed: work item = (source row r, 4-pixel chunk c)
  const int chunks = (fw + 3) >> 2;
  const int total = chunks * fh;
  constexpr int SU = 8;                  // 16-B loads in flight per lane: a 64 x 64 tile's patch (<= 2048 chunks) is ONE round
  for (int i0 = 0; i0 < total; i0 += 256 * SU) {
    u32x4 v[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = i0 + u * 256 + tid;
      if (i < total) {
        const int r = i / chunks, c = i - r * chunks;
        const int col = fx0 + 4 * c;
        const uint8_t* grow = src + static_cast<size_t>(fy0 + r) * sp;
        if (col + 3 <= op.cx1) v[u] = ld16(grow + static_cast<size_t>(col) * 4);
        else {
          v[u].x = ld4(grow + static_cast<size_t>(min(col, op.cx1)) * 4);
          v[u].y = ld4(grow + static_cast<size_t>(min(col + 1, op.cx1)) * 4);
          v[u].z = ld4(grow + static_cast<size_t>(min(col + 2, op.cx1)) * 4);
          v[u].w = ld4(grow + static_cast<size_t>(min(col + 3, op.cx1)) * 4);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = i0 + u * 256 + tid;
      if (i < total) {
        const int r = i / chunks, c = i - r * chunks;
        uint32_t* t = lds + (4 * c) * pitch + r;
        t[0] = v[u].x;
        if (4 * c + 1 < fw) t[pitch] = v[u].y;
        if (4 * c + 2 < fw) t[2 * pitch] = v[u].z;
        if (4 * c + 3 < fw) t[3 * pitch] = v[u].w;
      }
    }
  }
  __syncthreads();
  // ---- sample: lane = canvas X (drives the source ROW), loop over canvas Y (drives the source COLUMN, wave-uniform)
  const RowTaps cols = row_taps(op.kx, op.ox, Y0, Y1, op.cx0, op.cx1);   // all 64 lanes still active here (readlane source)
  const int X = X0 + lane;
  if (X >= X1) return true;
  const Tap tr = bilinear_tap(op.ky, op.oy, X, op.cy0, op.cy1);
  const uint32_t* col0 = lds + (tr.base - fy0);
  const bool opaque = (op.flags & OPF_OPAQUE) != 0;
  uint8_t* dcol = A.dst + static_cast<size_t>(X) * 4;
  if (op.flags & OPF_UNIT_SWAP) {      // pure quarter turn at 1:1: every canvas pixel IS one source pixel
    // exact integer indices (the bilinear tap pair clamps its base at the last column/row and would need t = 1 there)
    const int sy = nearest_tap(op.ky, op.oy, X, op.cy0, op.cy1) - fy0;
    const int step = op.kx > 0.0 ? pitch : -pitch;
    const uint32_t* a = lds + sy + (nearest_tap(op.kx, op.ox, Y0 + wave, op.cx0, op.cx1) - fx0) * pitch;
    for (int Y = Y0 + wave; Y < Y1; Y += 4, a += 4 * step) {
      uint32_t px = a[0];
      if (!opaque) px = over_int(px, bg);
      st4(dcol + static_cast<size_t>(Y) * A.dst_pitch, px);
    }
    return true;
  }
  // two canvas rows per step (Y and Y + 4): they share the row weight tr.t, so the packed blend of the SAMPLE_LDS path
  // applies (two pixels per v_pk_fma_f32 chain instead of one)
  int Y = Y0 + wave;
  for (; Y + 4 < Y1; Y += 8) {
    const Tap tc0 = row_tap(cols, Y - Y0), tc1 = row_tap(cols, Y + 4 - Y0);
    const uint32_t* a0 = col0 + (tc0.base - fx0) * pitch;   // T[sx][sy], T[sx][sy+1]
    const uint32_t* b0 = a0 + pitch;                         // T[sx+1][...]
    const uint32_t* a1 = col0 + (tc1.base - fx0) * pitch;
    const uint32_t* b1 = a1 + pitch;
    // (p00, p01, p10, p11): p01 = next source column, p10 = next source row
    const uint32_t p00[2] = {a0[0], a1[0]}, p01[2] = {b0[0], b1[0]}, p10[2] = {a0[1], a1[1]}, p11[2] = {b0[1], b1[1]};
    uint32_t o[2];
    if (opaque || ((p00[0] & p01[0] & p10[0] & p11[0] & p00[1] & p01[1] & p10[1] & p11[1]) >> 24) == 255u) {
      const float tx[2] = {tc0.t, tc1.t};
      bilerpN_opaque<2>(p00, p01, p10, p11, tx, tr.t, o);
    } else {
      o[0] = bilerp_over(p00[0], p01[0], p10[0], p11[0], tc0.t, tr.t, bg, false);
      o[1] = bilerp_over(p00[1], p01[1], p10[1], p11[1], tc1.t, tr.t, bg, false);
    }
    st4(dcol + static_cast<size_t>(Y) * A.dst_pitch, o[0]);
    st4(dcol + static_cast<size_t>(Y + 4) * A.dst_pitch, o[1]);
  }
  for (; Y < Y1; Y += 4) {
    const Tap tc = row_tap(cols, Y - Y0);
    const uint32_t* a = col0 + (tc.base - fx0) * pitch;
    const uint32_t* b = a + pitch;
    st4(dcol + static_cast<size_t>(Y) * A.dst_pitch, bilerp_over(a[0], b[0], a[1], b[1], tc.t, tr.t, bg, opaque));
  }
  return true;
}

// ------------------------------------------------------------------------------------------------ GENERAL
// one pixel through the whole paint stack, in canvas order, on a premultiplied 8-bit destination (what an
// immediate-mode Canvas with 8-bit premultiplied backing store does call by call)
IST_DEV uint32_t pixel_general(const LaunchArgs& A, const DevCell c, int X, int Y) {
  uint32_t d = c.bg;
  const bool nearest = (A.filter & 0xFF) == IST_FILTER_NEAREST;
  const bool area = (A.filter & 0xFF) == IST_FILTER_AREA;
  const bool edge_aa = (A.filter & IST_FILTER_EDGE_AA) != 0;
  for (int k = 0; k < c.stack_len; ++k) {
    const DevOp op = A.ops[A.stacks[c.stack_off + k]];
    // area of this pixel inside the destination rectangle (1 unless edge anti-aliasing is on and the edge is fractional)
    double cov = 1.0;
    if (edge_aa) {
      const double cx = fmin(static_cast<double>(X) + 1.0, op.xh) - fmax(static_cast<double>(X), op.xl);
      const double cy = fmin(static_cast<double>(Y) + 1.0, op.yh) - fmax(static_cast<double>(Y), op.yl);
      cov = fmin(fmax(cx, 0.0), 1.0) * fmin(fmax(cy, 0.0), 1.0);
      if (cov <= 0.0) continue;
    }
    const bool sw = (op.flags & OPF_SWAP) != 0;
    const int wx = sw ? Y : X, wy = sw ? X : Y;
    const uint8_t* src = A.src[op.image];
    const size_t sp = A.pitch[op.image];
    if (nearest || (op.flags & OPF_IDENTITY)) {
      const int ix = nearest_tap(op.kx, op.ox, wx, op.cx0, op.cx1);
      const int iy = nearest_tap(op.ky, op.oy, wy, op.cy0, op.cy1);
      const uint32_t s = ld4(src + static_cast<size_t>(iy) * sp + 4 * static_cast<size_t>(ix));
      if (cov >= 1.0) { d = over_int(s, d); continue; }
      // fractional edge: coverage-weighted source-over in fp64, the same operations in the same order as the oracle
      const uint32_t a = s >> 24;
      const double keep = 1.0 - cov * (static_cast<double>(a) / 255.0);
      uint32_t o = 0;
#pragma unroll
      for (int ch_ = 0; ch_ < 3; ++ch_) {
        const double P = static_cast<double>(ch(s, ch_) * a) / 255.0;
        const double v = floor(P * cov + static_cast<double>(ch(d, ch_)) * keep + 0.5);
        o |= static_cast<uint32_t>(fmin(fmax(v, 0.0), 255.0)) << (8 * ch_);
      }
      const double va = floor(static_cast<double>(a) * cov + static_cast<double>(d >> 24) * keep + 0.5);
      d = o | (static_cast<uint32_t>(fmin(fmax(va, 0.0), 255.0)) << 24);
    } else if (area && (fabs(op.kx) > 1.0 || fabs(op.ky) > 1.0)) {
      // IST_FILTER_AREA on a minifying draw: a box of width max(1, |k|) per axis around the sample position, source pixels
      // weighted by their overlap (fp32 sums of premultiplied taps; the oracle does the same sums in fp64)
      const double sxc = op.kx * (static_cast<double>(wx) + 0.5) + op.ox, syc = op.ky * (static_cast<double>(wy) + 0.5) + op.oy;
      const double bw = fmax(fabs(op.kx), 1.0), bh = fmax(fabs(op.ky), 1.0);
      const double xlo = sxc - 0.5 * bw, xhi = sxc + 0.5 * bw, ylo = syc - 0.5 * bh, yhi = syc + 0.5 * bh;
      const int ix0 = static_cast<int>(fmin(fmax(floor(xlo), -2.0e9), 2.0e9)), ix1 = static_cast<int>(fmin(fmax(ceil(xhi), -2.0e9), 2.0e9)) - 1;
      const int iy0 = static_cast<int>(fmin(fmax(floor(ylo), -2.0e9), 2.0e9)), iy1 = static_cast<int>(fmin(fmax(ceil(yhi), -2.0e9), 2.0e9)) - 1;
      float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
      for (int yy = iy0; yy <= iy1; ++yy) {
        const float oyw = static_cast<float>(fmin(static_cast<double>(yy) + 1.0, yhi) - fmax(static_cast<double>(yy), ylo));
        if (oyw <= 0.f) continue;
        const uint8_t* srow = src + static_cast<size_t>(min(max(yy, op.cy0), op.cy1)) * sp;
        float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
        for (int xx = ix0; xx <= ix1; ++xx) {
          const float oxw = static_cast<float>(fmin(static_cast<double>(xx) + 1.0, xhi) - fmax(static_cast<double>(xx), xlo));
          if (oxw <= 0.f) continue;
          const uint32_t s = ld4(srow + 4 * static_cast<size_t>(min(max(xx, op.cx0), op.cx1)));
          const float a = static_cast<float>(s >> 24);
          r0 += oxw * (static_cast<float>(ch(s, 0)) * a); r1 += oxw * (static_cast<float>(ch(s, 1)) * a);
          r2 += oxw * (static_cast<float>(ch(s, 2)) * a); r3 += oxw * a;
        }
        acc0 += oyw * r0; acc1 += oyw * r1; acc2 += oyw * r2; acc3 += oyw * r3;
      }
      const double norm = 1.0 / (bw * bh);
      const double Aa = static_cast<double>(acc3) * norm;
      const double keep = 1.0 - cov * (Aa / 255.0);
      const float accs[3] = {acc0, acc1, acc2};
      uint32_t o = 0;
#pragma unroll
      for (int ch_ = 0; ch_ < 3; ++ch_) {
        const double P = static_cast<double>(accs[ch_]) * norm / 255.0;
        const double v = floor(P * cov + static_cast<double>(ch(d, ch_)) * keep + 0.5);
        o |= static_cast<uint32_t>(fmin(fmax(v, 0.0), 255.0)) << (8 * ch_);
      }
      const double va = floor(Aa * cov + static_cast<double>(d >> 24) * keep + 0.5);
      d = o | (static_cast<uint32_t>(fmin(fmax(va, 0.0), 255.0)) << 24);
    } else {
      const Tap tx = bilinear_tap(op.kx, op.ox, wx, op.cx0, op.cx1);
      const Tap ty = bilinear_tap(op.ky, op.oy, wy, op.cy0, op.cy1);
      const uint8_t* r0 = src + static_cast<size_t>(ty.base) * sp + 4 * static_cast<size_t>(tx.base);
      const uint8_t* r1 = r0 + (op.cy1 > op.cy0 ? sp : 0);
      const size_t nx = op.cx1 > op.cx0 ? 4 : 0;
      const uint32_t p00 = ld4(r0), p01 = ld4(r0 + nx), p10 = ld4(r1), p11 = ld4(r1 + nx);
      if (cov >= 1.0) { d = bilerp_over(p00, p01, p10, p11, tx.t, ty.t, d); continue; }
      const float a00 = static_cast<float>(p00 >> 24), a01 = static_cast<float>(p01 >> 24);
      const float a10 = static_cast<float>(p10 >> 24), a11 = static_cast<float>(p11 >> 24);
      const double Aa = static_cast<double>(lerpf(lerpf(a00, a01, tx.t), lerpf(a10, a11, tx.t), ty.t));
      const double keep = 1.0 - cov * (Aa / 255.0);
      uint32_t o = 0;
#pragma unroll
      for (int ch_ = 0; ch_ < 3; ++ch_) {
        const float top = lerpf(static_cast<float>(ch(p00, ch_)) * a00, static_cast<float>(ch(p01, ch_)) * a01, tx.t);
        const float bot = lerpf(static_cast<float>(ch(p10, ch_)) * a10, static_cast<float>(ch(p11, ch_)) * a11, tx.t);
        const double P = static_cast<double>(lerpf(top, bot, ty.t)) / 255.0;
        const double v = floor(P * cov + static_cast<double>(ch(d, ch_)) * keep + 0.5);
        o |= static_cast<uint32_t>(fmin(fmax(v, 0.0), 255.0)) << (8 * ch_);
      }
      const double va = floor(Aa * cov + static_cast<double>(d >> 24) * keep + 0.5);
      d = o | (static_cast<uint32_t>(fmin(fmax(va, 0.0), 255.0)) << 24);
    }
  }
  // readback is straight alpha (ImageData): un-premultiply
  const uint32_t a = d >> 24;
  if (a == 255u) return d;
  if (a == 0u) return 0u;
  const uint32_t r = min(255u, (ch(d, 0) * 255u + a / 2u) / a);
  const uint32_t g = min(255u, (ch(d, 1) * 255u + a / 2u) / a);
  const uint32_t b = min(255u, (ch(d, 2) * 255u + a / 2u) / a);
  return r | (g << 8) | (b << 16) | (a << 24);
}

IST_DEV void tile_general(const LaunchArgs& A, const DevCell c, int X0, int Y0, int X1, int Y1) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int X = X0 + lane;
  if (X >= X1) return;
  for (int Y = Y0 + wave; Y < Y1; Y += 4)
    st4(A.dst + static_cast<size_t>(Y) * A.dst_pitch + static_cast<size_t>(X) * 4, pixel_general(A, c, X, Y));
}

// ------------------------------------------------------------------------------------------------ kernel
enum : int { HAS_FILL = 1, HAS_COPY = 2, HAS_SAMPLE = 4, HAS_GENERAL = 8, HAS_SWAP = 16, HAS_AREA = 32 };

template <int PATHS, int V>
IST_DEV void run_tile(const LaunchArgs& A, int64_t tile, bool fresh) {
  int ci, oi, X0, Y0;
  if (A.tiles) {                       // one 16-byte scalar load; the cell and the op are then fetched side by side
    const DevTile t = A.tiles[tile];
    ci = t.cell; oi = t.op; X0 = t.X0; Y0 = t.Y0;
  } else {
    // gigapixel jobs: binary search on the tile prefixes of the bands, then of the band's cells (wave-uniform loads)
    int lo = 0, hi = A.n_bands - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (A.bands[mid].tile_begin <= tile) lo = mid; else hi = mid - 1;
    }
    const DevBand b = A.bands[lo];
    const int local = static_cast<int>(tile - b.tile_begin);
    const int trow = local / b.tiles_per_row, rem = local - trow * b.tiles_per_row;
    lo = b.first_cell; hi = b.first_cell + b.n_cells - 1;
    while (lo < hi) {                      // the cell of this band that holds tile column `rem`
      const int mid = (lo + hi + 1) >> 1;
      if (A.cells[mid].band_x <= rem) lo = mid; else hi = mid - 1;
    }
    ci = lo;
    const DevCell c0 = A.cells[lo];
    oi = c0.op;
    X0 = c0.X0 + (rem - c0.band_x) * c0.tile_w; Y0 = c0.Y0 + trow * c0.tile_h;
  }
  const DevCell c = A.cells[ci];       // by value: scalar loads once; a reference would be re-read after every store
  DevOp op_;                           // fetched together with the cell (both indices come from the tile entry)
  if (oi >= 0) op_ = A.ops[oi]; else __builtin_memset(&op_, 0, sizeof(op_));
  const int X1 = min(X0 + c.tile_w, c.X1), Y1 = min(Y0 + c.tile_h, c.Y1);
  const int lg = c.tile_w >= 256 ? 31 - __builtin_clz(c.tile_w >> 8) : 0;       // FILL / COPY tiles are 256 << lg wide
  const int path = c.path;
  if ((PATHS & HAS_COPY) && path == PATH_COPY) {
    const DevOp op = op_;
    // V = 0 is what ships; 1..3 are kept as measured alternatives for tools/sweep_variants.py (IST_TUNING=1)
    if (V == 0) tile_copy<2, true, true, true>(A, op, c.bg, lg, X0, Y0, X1, Y1);        // 256x8 tile: 2 rows per wave in flight
    else if (V == 1) tile_copy<8, true, true, true>(A, op, c.bg, lg, X0, Y0, X1, Y1);   // 8 rows per wave in flight
    else if (V == 2) tile_copy<8, false, true, true>(A, op, c.bg, lg, X0, Y0, X1, Y1);  // a wave owns consecutive rows
    else tile_copy<8, true, false, false>(A, op, c.bg, lg, X0, Y0, X1, Y1);             // no non-temporal hints
  } else if ((PATHS & HAS_FILL) && path == PATH_FILL) {
    tile_fill(A, c.bg, lg, X0, Y0, X1, Y1);
  } else if ((PATHS & HAS_SAMPLE) && path == PATH_SAMPLE_LDS) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (c.tile_w == 256) tile_sample_lds<4>(A, op_, c.bg, X0, Y0, X1, Y1, c.sub_h, lds, fresh);
    else if (c.tile_w == 128) tile_sample_lds<2>(A, op_, c.bg, X0, Y0, X1, Y1, c.sub_h, lds, fresh);
    else tile_sample_lds<1>(A, op_, c.bg, X0, Y0, X1, Y1, c.sub_h, lds, fresh);
  } else if ((PATHS & HAS_SAMPLE) && path == PATH_SAMPLE_STREAM) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (c.tile_w == 256) tile_sample_stream<4>(A, op_, c.bg, X0, Y0, X1, Y1, c.sub_h, lds, fresh);
    else if (c.tile_w == 128) tile_sample_stream<2>(A, op_, c.bg, X0, Y0, X1, Y1, c.sub_h, lds, fresh);
    else tile_sample_stream<1>(A, op_, c.bg, X0, Y0, X1, Y1, c.sub_h, lds, fresh);
  } else if ((PATHS & HAS_AREA) && path == PATH_AREA_STREAM) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const bool opq = (op_.flags & OPF_OPAQUE) != 0;
    // (tiles are at most 128 canvas pixels wide: every further 64 pixels per lane cost ~15 VGPRs, and at 4 the kernel fell to 3 waves per SIMD)
    if (c.tile_w > 64) { if (opq) tile_area_stream<2, true>(A, op_, c.bg, X0, Y0, X1, Y1, lds); else tile_area_stream<2, false>(A, op_, c.bg, X0, Y0, X1, Y1, lds); }
    else { if (opq) tile_area_stream<1, true>(A, op_, c.bg, X0, Y0, X1, Y1, lds); else tile_area_stream<1, false>(A, op_, c.bg, X0, Y0, X1, Y1, lds); }
  } else if ((PATHS & HAS_SWAP) && path == PATH_SWAP_LDS) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (!tile_swap_lds(A, op_, c.bg, X0, Y0, X1, Y1, lds)) tile_general(A, c, X0, Y0, X1, Y1);
  } else if ((PATHS & HAS_SAMPLE) && path == PATH_SAMPLE) {
    if ((A.filter & 0xFF) == IST_FILTER_NEAREST) tile_sample<IST_FILTER_NEAREST>(A, op_, c.bg, X0, Y0, X1, Y1);
    else tile_sample<IST_FILTER_BILINEAR>(A, op_, c.bg, X0, Y0, X1, Y1);
  } else if (PATHS & HAS_GENERAL) {
    tile_general(A, c, X0, Y0, X1, Y1);
  }
}

// PATHS: which cell kinds this instantiation can render (a job with only fill/copy cells gets the lean one: fewer
// VGPRs); PERSIST: one block per tile (false) or a fixed grid striding over the tiles (true)
template <int PATHS, int V, bool PERSIST>
__global__ __launch_bounds__(256) void ist_stitch_kernel(const LaunchArgs A, const int64_t n_tiles) {
  if (PERSIST) {
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) run_tile<PATHS, V>(A, t, false);
  } else {
    run_tile<PATHS, V>(A, static_cast<int64_t>(blockIdx.x), true);
  }
}

// the instantiation that carries the streamed box filter: held to 5 waves per SIMD (it compiled to 103 VGPRs, one register
// past that step)
#ifndef IST_AREA_WAVES          // (compile-time experiment switch: -DIST_AREA_WAVES=6 rebuilds the variant LAB_NOTES.md quotes)
#define IST_AREA_WAVES 5
#endif
template <int PATHS, int V, bool PERSIST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(IST_AREA_WAVES))) void ist_stitch_area_kernel(const LaunchArgs A, const int64_t n_tiles) {
  if (PERSIST) {
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) run_tile<PATHS, V>(A, t, false);
  } else {
    run_tile<PATHS, V>(A, static_cast<int64_t>(blockIdx.x), true);
  }
}

template <int PATHS, int V, bool PERSIST>
static void launch_one(const LaunchArgs& args, int64_t n_tiles, hipStream_t stream, int persist_blocks, unsigned dyn_lds) {
  const unsigned grid = PERSIST ? static_cast<unsigned>(std::min<int64_t>(n_tiles, persist_blocks)) : static_cast<unsigned>(n_tiles);
  const unsigned dyn = std::max(dyn_lds, static_cast<unsigned>(args.lds_words) * 4u);   // dyn_lds: IST_DYN_LDS tuning knob (unused LDS caps the workgroups per CU)
  if constexpr ((PATHS & HAS_AREA) != 0) hipLaunchKernelGGL((ist_stitch_area_kernel<PATHS, V, PERSIST>), dim3(grid), dim3(256), dyn, stream, args, n_tiles);
  else hipLaunchKernelGGL((ist_stitch_kernel<PATHS, V, PERSIST>), dim3(grid), dim3(256), dyn, stream, args, n_tiles);
}

template <int PATHS>
static void launch_variant(int v, bool persist, const LaunchArgs& a, int64_t n, hipStream_t s, int pb, unsigned dl) {
  if (persist) { launch_one<PATHS, 0, true>(a, n, s, pb, dl); return; }      // grid-stride form of the shipped variant
  switch (v) {
    case 1: launch_one<PATHS, 1, false>(a, n, s, pb, dl); break;
    case 2: launch_one<PATHS, 2, false>(a, n, s, pb, dl); break;
    case 3: launch_one<PATHS, 3, false>(a, n, s, pb, dl); break;
    default: launch_one<PATHS, 0, false>(a, n, s, pb, dl); break;
  }
}

int launch_stitch(const LaunchArgs& args, int64_t n_tiles, int kind, void* stream) {
  if (n_tiles <= 0) return IST_OK;
  // tuning knobs, read only when the process was started with IST_TUNING=1 (tools/sweep_*.py): IST_VARIANT = copy variant
  // + 100 * persistent; IST_PERSIST_BLOCKS; IST_FULL_KERNEL; IST_DYN_LDS.  Production launches touch no environment.
  const bool tuning = tuning_mode();
  const int knob = tuning && std::getenv("IST_VARIANT") ? std::atoi(std::getenv("IST_VARIANT")) : 0;
  const int pb = tuning && std::getenv("IST_PERSIST_BLOCKS") ? std::atoi(std::getenv("IST_PERSIST_BLOCKS")) : 2048;
  const bool full = tuning && std::getenv("IST_FULL_KERNEL") != nullptr;
  const unsigned dl = tuning && std::getenv("IST_DYN_LDS") ? static_cast<unsigned>(std::atoi(std::getenv("IST_DYN_LDS"))) : 0u;
  const int v = knob % 100;
  const bool persist = knob >= 100;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // three instantiations of one template, by what the job's cells need: the fewer paths, the fewer registers and the
  // more workgroups per CU (fill/copy: 28 VGPRs; + axis-aligned resampling; + quarter turns and the per-pixel stack)
  // (the streamed box filter keeps up to 8 source rows per lane in flight: it has its own instantiation so that its register
  // count does not lower the occupancy of the jobs that never use it)
  if (kind == 0 && !full) launch_variant<HAS_FILL | HAS_COPY>(v, persist, args, n_tiles, s, pb, dl);
  else if (kind == 1 && !full) launch_variant<HAS_FILL | HAS_COPY | HAS_SAMPLE>(v, persist, args, n_tiles, s, pb, dl);
  else if (kind == 3 && !full) launch_variant<HAS_FILL | HAS_COPY | HAS_SAMPLE | HAS_AREA>(v, persist, args, n_tiles, s, pb, dl);
  else if (kind == 4 || full) launch_variant<HAS_FILL | HAS_COPY | HAS_SAMPLE | HAS_SWAP | HAS_GENERAL | HAS_AREA>(v, persist, args, n_tiles, s, pb, dl);
  else launch_variant<HAS_FILL | HAS_COPY | HAS_SAMPLE | HAS_SWAP | HAS_GENERAL>(v, persist, args, n_tiles, s, pb, dl);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(IST_E_HIP, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return IST_OK;
}

}  // namespace ist
