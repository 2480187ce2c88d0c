// ist_jpeg_kernels.hip — the data-parallel half of JPEG decode on gfx950: dequantise + 8x8 inverse DCT per block, then
// chroma upsampling + YCbCr->RGB per pixel, from coefficient planes produced by the host entropy decoder (ist_jpeg.cpp).
//
// The arithmetic is the public "slow-but-accurate" integer IDCT (Loeffler-Ligtenberg-Moschytz, 13-bit constants,
// 2-bit pass-1 scaling), triangle-filter ("fancy") chroma upsampling and the 16-bit fixed-point YCbCr->RGB tables that
// the IJG / libjpeg-turbo decoders use by default, so that the pixels agree with the witness the tests have (PIL).
#include <hip/hip_runtime.h>

#include <cstring>

#include "ist_internal.h"
#include "ist_jpeg.h"

namespace ist {

namespace {

constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270,
              F_0_899976223 = 7373, F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137,
              F_1_961570560 = 16069, F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 1-D pass of the LLM inverse DCT on 8 values
__device__ __forceinline__ void idct8(const int in[8], int out[8], int shift, bool pass1) {
  int z2 = in[2], z3 = in[6];
  int z1 = (z2 + z3) * F_0_541196100;
  int tmp2 = z1 + z3 * (-F_1_847759065);
  int tmp3 = z1 + z2 * F_0_765366865;
  z2 = in[0]; z3 = in[4];
  int tmp0 = (z2 + z3) << CONST_BITS;
  int tmp1 = (z2 - z3) << CONST_BITS;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  const int z5 = (z3 + z4) * F_1_175875602;
  tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
  z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  (void)pass1;
  out[0] = descale(tmp10 + tmp3, shift); out[7] = descale(tmp10 - tmp3, shift);
  out[1] = descale(tmp11 + tmp2, shift); out[6] = descale(tmp11 - tmp2, shift);
  out[2] = descale(tmp12 + tmp1, shift); out[5] = descale(tmp12 - tmp1, shift);
  out[3] = descale(tmp13 + tmp0, shift); out[4] = descale(tmp13 - tmp0, shift);
}

// (the quantisation table travels BY VALUE in the kernel arguments: 128 bytes, read with scalar loads - as a 128-byte upload per
// component it was three 5 us blit copies in front of every image's reconstruction)
// ALL components of an image in ONE launch (a workgroup belongs to one component: wg0[c] = its first workgroup): as a launch
// per component the reconstruction of nine photos was 27 launches on the file pipeline's critical submission path.
struct IdctComp { const int16_t* coef; uint8_t* plane; int blocks_x, n_blocks; uint16_t q[64]; };
struct IdctArgs { IdctComp comp[3]; int wg0[3]; };

// one thread per 8x8 block: 128 B of coefficients in, 64 samples out (plane row pitch = blocks_x * 8)
__global__ __launch_bounds__(128) void ist_jpeg_idct_kernel(const IdctArgs P) {
  const int wg = static_cast<int>(blockIdx.x);
  const int ci = wg >= P.wg0[2] ? 2 : (wg >= P.wg0[1] ? 1 : 0);             // (workgroup-uniform)
  const IdctComp& A = P.comp[ci];
  const int b = (wg - P.wg0[ci]) * static_cast<int>(blockDim.x) + static_cast<int>(threadIdx.x);
  if (b >= A.n_blocks) return;
  const int by = b / A.blocks_x, bx = b - by * A.blocks_x;
  const int16_t* c = A.coef + static_cast<size_t>(b) * 64;
  int ws[64];
  // pass 1: columns
#pragma unroll
  for (int x = 0; x < 8; ++x) {
    int in[8], o[8];
#pragma unroll
    for (int y = 0; y < 8; ++y) in[y] = static_cast<int>(c[y * 8 + x]) * static_cast<int>(A.q[y * 8 + x]);
    idct8(in, o, CONST_BITS - PASS1_BITS, true);
#pragma unroll
    for (int y = 0; y < 8; ++y) ws[y * 8 + x] = o[y];
  }
  // pass 2: rows, level shift + clamp
  uint8_t* dst = A.plane + (static_cast<size_t>(by) * 8) * (static_cast<size_t>(A.blocks_x) * 8) + static_cast<size_t>(bx) * 8;
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    int in[8], o[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) in[x] = ws[y * 8 + x];
    idct8(in, o, CONST_BITS + PASS1_BITS + 3, false);
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      lo |= static_cast<uint32_t>(min(max(o[x] + 128, 0), 255)) << (8 * x);
      hi |= static_cast<uint32_t>(min(max(o[x + 4] + 128, 0), 255)) << (8 * x);
    }
    uint2 v; v.x = lo; v.y = hi;
    *reinterpret_cast<uint2*>(dst + static_cast<size_t>(y) * A.blocks_x * 8) = v;
  }
}

// sparse -> dense: one thread per block writes its non-zero coefficients into the (zeroed) plane
__global__ __launch_bounds__(256) void ist_jpeg_scatter_kernel(const uint32_t* ent, const uint32_t* start, const uint8_t* cnt, int16_t* coef, int n_blocks) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  const uint32_t s0 = start[b], n = cnt[b];
  int16_t* c = coef + static_cast<size_t>(b) * 64;
  for (uint32_t k = 0; k < n; ++k) { const uint32_t e = ent[s0 + k]; c[(e >> 16) & 63u] = static_cast<int16_t>(e & 0xFFFFu); }
}

struct ColorArgs {
  const uint8_t* Y; const uint8_t* Cb; const uint8_t* Cr;
  int pitch_y, pitch_c;            // plane row pitches
  int width, height;               // image size
  int cw, chh;                     // true chroma plane size (ceil(width*1/hmax), ceil(height*1/vmax))
  int hs, vs;                      // luma-to-chroma ratios (1 or 2)
  int ncomp;
  uint8_t* out; size_t out_pitch;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// triangle-filter upsampling of one chroma sample at full-resolution position (x, y)
__device__ __forceinline__ int chroma_at(const uint8_t* P, int pitch, int cw, int chh, int hs, int vs, int x, int y) {
  if (hs == 1 && vs == 1) return P[static_cast<size_t>(y) * pitch + x];
  if (hs == 2 && vs == 1) {                       // h2v1: 3/4 nearer + 1/4 further column
    const int i = x >> 1;
    const uint8_t* r = P + static_cast<size_t>(y) * pitch;
    if (x & 1) return i == cw - 1 ? r[i] : (3 * r[i] + r[i + 1] + 2) >> 2;
    return i == 0 ? r[i] : (3 * r[i] + r[i - 1] + 1) >> 2;
  }
  if (hs == 1 && vs == 2) {                       // h1v2: 3/4 nearer + 1/4 further row
    const int j = y >> 1;
    const int jn = (y & 1) ? min(j + 1, chh - 1) : max(j - 1, 0);
    const int a = P[static_cast<size_t>(j) * pitch + x], b = P[static_cast<size_t>(jn) * pitch + x];
    return (3 * a + b + ((y & 1) ? 2 : 1)) >> 2;
  }
  // h2v2: vertical 3:1 first (unscaled), then horizontal 3:1, one final shift by 4
  const int j = y >> 1, i = x >> 1;
  const int jn = (y & 1) ? min(j + 1, chh - 1) : max(j - 1, 0);
  const uint8_t* r0 = P + static_cast<size_t>(j) * pitch;
  const uint8_t* r1 = P + static_cast<size_t>(jn) * pitch;
  const int cur = 3 * r0[i] + r1[i];
  if (x & 1) {
    if (i == cw - 1) return (cur * 4 + 7) >> 4;
    return (cur * 3 + (3 * r0[i + 1] + r1[i + 1]) + 7) >> 4;
  }
  if (i == 0) return (cur * 4 + 8) >> 4;
  return (cur * 3 + (3 * r0[i - 1] + r1[i - 1]) + 8) >> 4;
}

__device__ __forceinline__ uint32_t ycc_to_rgba(int Yv, int cb, int cr) {
  // 16-bit fixed point: FIX(1.40200)=91881, FIX(1.77200)=116130, FIX(0.71414)=46802, FIX(0.34414)=22554
  const int r = Yv + ((91881 * cr + 32768) >> 16);
  const int b = Yv + ((116130 * cb + 32768) >> 16);
  const int g = Yv + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
  return static_cast<uint32_t>(clampi(r, 0, 255)) | (static_cast<uint32_t>(clampi(g, 0, 255)) << 8) |
         (static_cast<uint32_t>(clampi(b, 0, 255)) << 16) | 0xFF000000u;
}

// 4:2:0 (the photo case), one thread per 4 output pixels x0 .. x0+3 of one row: the four pixels lie over chroma columns
// i0 and i0+1 and blend with i0-1 and i0+2, so the thread reads 4 columns x 2 rows per chroma plane ONCE (the generic
// path below reads 4 bytes per plane per PIXEL) and blends them vertically once.  Clamping a neighbour's column index to
// the plane reproduces the edge rules of chroma_at exactly: (4 c + 8) >> 4 = (3 c + c + 8) >> 4.
__device__ __forceinline__ void color4_h2v2(const ColorArgs& A, int x0, int y, uint32_t px[4]) {
  const int j = y >> 1, i0 = x0 >> 1;
  const int jn = (y & 1) ? min(j + 1, A.chh - 1) : max(j - 1, 0);
  const int col[4] = {max(i0 - 1, 0), i0, min(i0 + 1, A.cw - 1), min(i0 + 2, A.cw - 1)};
  int cb[4], cr[4];
  {
    const uint8_t* b0 = A.Cb + static_cast<size_t>(j) * A.pitch_c; const uint8_t* b1 = A.Cb + static_cast<size_t>(jn) * A.pitch_c;
    const uint8_t* r0 = A.Cr + static_cast<size_t>(j) * A.pitch_c; const uint8_t* r1 = A.Cr + static_cast<size_t>(jn) * A.pitch_c;
#pragma unroll
    for (int k = 0; k < 4; ++k) { cb[k] = 3 * b0[col[k]] + b1[col[k]]; cr[k] = 3 * r0[col[k]] + r1[col[k]]; }
  }
  const uint32_t yy = *reinterpret_cast<const uint32_t*>(A.Y + static_cast<size_t>(y) * A.pitch_y + x0);   // (x0 % 4 == 0, pitch % 8 == 0)
  // pixel k: own column own[k], neighbour nb[k], rounding 8 (even x) / 7 (odd x)
  const int own[4] = {1, 1, 2, 2}, nb[4] = {0, 2, 1, 3}, rnd[4] = {8, 7, 8, 7};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int u = (cb[own[k]] * 3 + cb[nb[k]] + rnd[k]) >> 4, v = (cr[own[k]] * 3 + cr[nb[k]] + rnd[k]) >> 4;
    px[k] = ycc_to_rgba(static_cast<int>((yy >> (8 * k)) & 255u), u - 128, v - 128);
  }
}

// one thread per 4 output pixels
__global__ __launch_bounds__(256) void ist_jpeg_color_kernel(const ColorArgs A) {
  const int gx = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int y = blockIdx.y;
  if (gx >= A.width) return;
  uint32_t px[4];
  if (A.ncomp == 3 && A.hs == 2 && A.vs == 2) color4_h2v2(A, gx, y, px);
  else
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = min(gx + k, A.width - 1);
    const int Yv = A.Y[static_cast<size_t>(y) * A.pitch_y + x];
    int r = Yv, g = Yv, b = Yv;
    if (A.ncomp == 3) {
      const int cb = chroma_at(A.Cb, A.pitch_c, A.cw, A.chh, A.hs, A.vs, x, y) - 128;
      const int cr = chroma_at(A.Cr, A.pitch_c, A.cw, A.chh, A.hs, A.vs, x, y) - 128;
      // 16-bit fixed point: FIX(1.40200)=91881, FIX(1.77200)=116130, FIX(0.71414)=46802, FIX(0.34414)=22554
      r = Yv + ((91881 * cr + 32768) >> 16);
      b = Yv + ((116130 * cb + 32768) >> 16);
      g = Yv + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    }
    px[k] = static_cast<uint32_t>(clampi(r, 0, 255)) | (static_cast<uint32_t>(clampi(g, 0, 255)) << 8) |
            (static_cast<uint32_t>(clampi(b, 0, 255)) << 16) | 0xFF000000u;
  }
  uint8_t* o = A.out + static_cast<size_t>(y) * A.out_pitch + static_cast<size_t>(gx) * 4;
  const int nv = A.width - gx;
  if (nv >= 4 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
    uint4 v; v.x = px[0]; v.y = px[1]; v.z = px[2]; v.w = px[3];
    *reinterpret_cast<uint4*>(o) = v;
  } else {
    for (int k = 0; k < 4 && k < nv; ++k) *reinterpret_cast<uint32_t*>(o + 4 * k) = px[k];
  }
}

}  // namespace

int jpeg_launch_scatter(const uint32_t* d_ent, const uint32_t* d_start, const uint8_t* d_cnt, int16_t* d_coef, int n_blocks, void* stream_) {
  if (n_blocks <= 0) return IST_OK;
  hipLaunchKernelGGL(ist_jpeg_scatter_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_), d_ent, d_start, d_cnt, d_coef, n_blocks);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(IST_E_HIP, std::string("JPEG scatter launch failed: ") + hipGetErrorString(e));
  return IST_OK;
}

int jpeg_launch_reconstruct(const JpegDeviceJob& J, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  {
    IdctArgs a;
    std::memset(&a, 0, sizeof a);
    int wgs = 0;
    for (int c = 0; c < 3; ++c) {
      a.wg0[c] = wgs;                                  // (a component the image does not have: no workgroups, never selected)
      if (c >= J.ncomp) { a.wg0[c] = 0x7fffffff; continue; }
      IdctComp& C = a.comp[c];
      C.coef = J.d_coef[c]; C.plane = J.d_plane[c]; C.blocks_x = J.blocks_x[c]; C.n_blocks = J.blocks_x[c] * J.blocks_y[c];
      for (int k = 0; k < 64; ++k) C.q[k] = J.q_host[c] ? J.q_host[c][k] : 1;
      wgs += (C.n_blocks + 127) / 128;
    }
    if (wgs > 0) hipLaunchKernelGGL(ist_jpeg_idct_kernel, dim3(static_cast<unsigned>(wgs)), dim3(128), 0, stream, a);
  }
  ColorArgs ca;
  ca.Y = J.d_plane[0]; ca.Cb = J.ncomp == 3 ? J.d_plane[1] : nullptr; ca.Cr = J.ncomp == 3 ? J.d_plane[2] : nullptr;
  ca.pitch_y = J.blocks_x[0] * 8; ca.pitch_c = J.ncomp == 3 ? J.blocks_x[1] * 8 : 0;
  ca.width = J.width; ca.height = J.height;
  ca.hs = J.hmax; ca.vs = J.vmax;
  ca.cw = (J.width + J.hmax - 1) / J.hmax; ca.chh = (J.height + J.vmax - 1) / J.vmax;
  ca.ncomp = J.ncomp; ca.out = J.out; ca.out_pitch = J.out_pitch;
  hipLaunchKernelGGL(ist_jpeg_color_kernel, dim3((J.width + 1023) / 1024, J.height), dim3(256), 0, stream, ca);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(IST_E_HIP, std::string("JPEG reconstruct launch failed: ") + hipGetErrorString(e));
  return IST_OK;
}

}  // namespace ist
