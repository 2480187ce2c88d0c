// ist_kernels.hip — the fused resample + blit kernel for gfx950 (MI355X, CDNA4).
//
// Stands in for the raster work the WeChat Canvas performs for the calls onStitch issues
// (reference: pages/index/index.js:1423-1424 fillRect, utils/canvas.js:153-202 drawImage under a CTM,
// index.js:1577-1579 same-size readback).  ONE launch writes every canvas pixel exactly once:
//
//   grid      one 256-thread workgroup (4 wave64) per output tile; tiles enumerate the cells of ist_compile.cpp
//   tile      256 px x 32 rows for fill / copy / sample cells: a wave row is 64 lanes x 16 B = 1 KiB contiguous
//             (one global_store_dwordx4 per lane); 64 x 64 for general cells
//   paths     FILL    constant colour                                   (gaps, centring margins, rounding slack)
//             COPY    1:1 rect: 16-B loads -> 16-B stores, 8 rows in flight per wave (the BASELINE configs)
//             SAMPLE  nearest / bilinear resample, source x driven by canvas x: per-lane column taps computed
//                     once per tile in fp64 (bit-identical to the oracle), rows streamed, fp32 lerp
//             GENERAL paint stack evaluated per pixel in canvas order (EXIF quarter turns, overlapping draws,
//                     translucent canvas)
//
// HBM-bound byte movement: no MFMA, no LDS staging in this revision (taps come through the vector L1).
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (the fp64 coordinate math must not be fused).
#include <hip/hip_runtime.h>

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

IST_DEV uint32_t to_u8(float v) {
  v = floorf(v + 0.5f);
  v = fminf(fmaxf(v, 0.0f), 255.0f);
  return static_cast<uint32_t>(v);
}

// bilinear blend of four straight-alpha taps, composited over a premultiplied destination pixel
IST_DEV uint32_t bilerp_over(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11, float tx, float ty, uint32_t d) {
  const uint32_t amin = (p00 & p01 & p10 & p11) >> 24;
  if (amin == 255u) {   // all taps opaque: plain bilinear, result replaces the destination
    uint32_t o = 0xFF000000u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float top = lerpf(static_cast<float>(ch(p00, c)), static_cast<float>(ch(p01, c)), tx);
      const float bot = lerpf(static_cast<float>(ch(p10, c)), static_cast<float>(ch(p11, c)), tx);
      o |= to_u8(lerpf(top, bot, ty)) << (8 * c);
    }
    return o;
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

// ------------------------------------------------------------------------------------------------ FILL
IST_DEV void tile_fill(const LaunchArgs& A, uint32_t colour, int X0, int Y0, int X1, int Y1) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int X = X0 + lane * 4;
  const int nv = X1 - X;
  if (nv <= 0) return;
  const u32x4 v = {colour, colour, colour, colour};
  for (int Y = Y0 + wave; Y < Y1; Y += 4) {
    uint8_t* d = A.dst + static_cast<size_t>(Y) * A.dst_pitch + static_cast<size_t>(X) * 4;
    if (nv >= 4) st16(d, v);
    else for (int p = 0; p < nv; ++p) st4(d + 4 * p, colour);
  }
}

// ------------------------------------------------------------------------------------------------ COPY
template <int U>
IST_DEV void tile_copy(const LaunchArgs& A, const DevOp& op, uint32_t bg, int X0, int Y0, int X1, int Y1) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int X = X0 + lane * 4;
  const int nv = X1 - X;
  if (nv <= 0) return;
  const size_t sp = A.pitch[op.image];
  const uint8_t* s = A.src[op.image] + (static_cast<int64_t>(X) + static_cast<int64_t>(op.ox)) * 4 +
                     static_cast<int64_t>(op.oy) * static_cast<int64_t>(sp);
  uint8_t* d = A.dst + static_cast<size_t>(X) * 4;
  const bool opaque = (op.flags & OPF_OPAQUE) != 0;
  if (nv >= 4) {
    for (int Y = Y0 + wave; Y < Y1; Y += 4 * U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int y = Y + 4 * u;
        if (y < Y1) v[u] = ld16(s + static_cast<size_t>(y) * sp);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int y = Y + 4 * u;
        if (y < Y1) {
          u32x4 o = v[u];
          if (!opaque && ((o.x & o.y & o.z & o.w) >> 24) != 255u) {
            o.x = over_int(o.x, bg); o.y = over_int(o.y, bg); o.z = over_int(o.z, bg); o.w = over_int(o.w, bg);
          }
          st16(d + static_cast<size_t>(y) * A.dst_pitch, o);
        }
      }
    }
  } else {   // ragged right edge of the cell: at most one lane per row segment
    for (int Y = Y0 + wave; Y < Y1; Y += 4)
      for (int p = 0; p < nv; ++p) {
        uint32_t v = ld4(s + static_cast<size_t>(Y) * sp + 4 * p);
        if (!opaque) v = over_int(v, bg);
        st4(d + static_cast<size_t>(Y) * A.dst_pitch + 4 * p, v);
      }
  }
}

// ------------------------------------------------------------------------------------------------ SAMPLE
template <int FILTER>
IST_DEV void tile_sample(const LaunchArgs& A, const DevOp& op, uint32_t bg, int X0, int Y0, int X1, int Y1) {
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
    uint32_t o[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      uint32_t p00, p01, p10, p11;
      const size_t off = 4 * static_cast<size_t>(tx[p].base);
      if (pair_x) {
        const u32x2 a = ld8(r0 + off), b = ld8(r1 + off);
        p00 = a.x; p01 = a.y; p10 = b.x; p11 = b.y;
      } else {
        p00 = p01 = ld4(r0 + off); p10 = p11 = ld4(r1 + off);
      }
      o[p] = bilerp_over(p00, p01, p10, p11, tx[p].t, ty.t, bg);
    }
    uint8_t* dp = d + static_cast<size_t>(Y) * A.dst_pitch;
    if (nv >= 4) { const u32x4 v = {o[0], o[1], o[2], o[3]}; st16(dp, v); }
    else { st4(dp, o[0]); if (nv > 1) st4(dp + 4, o[1]); if (nv > 2) st4(dp + 8, o[2]); }
  }
}

// ------------------------------------------------------------------------------------------------ GENERAL
// one pixel through the whole paint stack, in canvas order, on a premultiplied 8-bit destination (what an
// immediate-mode Canvas with 8-bit premultiplied backing store does call by call)
IST_DEV uint32_t pixel_general(const LaunchArgs& A, const DevCell& c, int X, int Y) {
  uint32_t d = c.bg;
  for (int k = 0; k < c.stack_len; ++k) {
    const DevOp& op = A.ops[A.stacks[c.stack_off + k]];
    const bool sw = (op.flags & OPF_SWAP) != 0;
    const int wx = sw ? Y : X, wy = sw ? X : Y;
    const uint8_t* src = A.src[op.image];
    const size_t sp = A.pitch[op.image];
    if (A.filter == IST_FILTER_NEAREST || (op.flags & OPF_IDENTITY)) {
      const int ix = nearest_tap(op.kx, op.ox, wx, op.cx0, op.cx1);
      const int iy = nearest_tap(op.ky, op.oy, wy, op.cy0, op.cy1);
      d = over_int(ld4(src + static_cast<size_t>(iy) * sp + 4 * static_cast<size_t>(ix)), d);
    } else {
      const Tap tx = bilinear_tap(op.kx, op.ox, wx, op.cx0, op.cx1);
      const Tap ty = bilinear_tap(op.ky, op.oy, wy, op.cy0, op.cy1);
      const uint8_t* r0 = src + static_cast<size_t>(ty.base) * sp + 4 * static_cast<size_t>(tx.base);
      const uint8_t* r1 = r0 + (op.cy1 > op.cy0 ? sp : 0);
      const size_t nx = op.cx1 > op.cx0 ? 4 : 0;
      d = bilerp_over(ld4(r0), ld4(r0 + nx), ld4(r1), ld4(r1 + nx), tx.t, ty.t, d);
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

IST_DEV void tile_general(const LaunchArgs& A, const DevCell& c, int X0, int Y0, int X1, int Y1) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int X = X0 + lane;
  if (X >= X1) return;
  for (int Y = Y0 + wave; Y < Y1; Y += 4)
    st4(A.dst + static_cast<size_t>(Y) * A.dst_pitch + static_cast<size_t>(X) * 4, pixel_general(A, c, X, Y));
}

// ------------------------------------------------------------------------------------------------ kernel
__global__ __launch_bounds__(256) void ist_stitch_kernel(const LaunchArgs A) {
  const int64_t tile = static_cast<int64_t>(blockIdx.x);
  // cells are few (tens): binary search on the tile prefix with wave-uniform (scalar) loads
  int lo = 0, hi = A.n_cells - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (A.cells[mid].tile_begin <= tile) lo = mid; else hi = mid - 1;
  }
  const DevCell& c = A.cells[lo];
  const int local = static_cast<int>(tile - c.tile_begin);
  const int trow = local / c.tiles_x, tcol = local - trow * c.tiles_x;
  const int X0 = c.X0 + tcol * c.tile_w, Y0 = c.Y0 + trow * c.tile_h;
  const int X1 = min(X0 + c.tile_w, c.X1), Y1 = min(Y0 + c.tile_h, c.Y1);
  switch (c.path) {
    case PATH_FILL: tile_fill(A, c.bg, X0, Y0, X1, Y1); break;
    case PATH_COPY: tile_copy<8>(A, A.ops[c.op], c.bg, X0, Y0, X1, Y1); break;
    case PATH_SAMPLE:
      if (A.filter == IST_FILTER_NEAREST) tile_sample<IST_FILTER_NEAREST>(A, A.ops[c.op], c.bg, X0, Y0, X1, Y1);
      else tile_sample<IST_FILTER_BILINEAR>(A, A.ops[c.op], c.bg, X0, Y0, X1, Y1);
      break;
    default: tile_general(A, c, X0, Y0, X1, Y1); break;
  }
}

int launch_stitch(const LaunchArgs& args, int64_t n_tiles, void* stream) {
  if (n_tiles <= 0) return IST_OK;
  hipLaunchKernelGGL(ist_stitch_kernel, dim3(static_cast<unsigned>(n_tiles)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), args);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(IST_E_HIP, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return IST_OK;
}

}  // namespace ist
