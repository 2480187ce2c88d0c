// ist_jpeg_kernels.hip — the data-parallel half of JPEG decode on gfx950: dequantise + 8x8 inverse DCT per block, then
// chroma upsampling + YCbCr->RGB per pixel, from coefficient planes produced by the entropy decoders (GPU: ist_jpeg_gpu.hip,
// host: ist_jpeg.cpp).  Reference anchor: the Image.src decode of loadImageFrom (utils/canvas.js:27-121; pages/index/index.js:1441-1463).
//
// The arithmetic is the public "slow-but-accurate" integer IDCT (Loeffler-Ligtenberg-Moschytz, 13-bit constants,
// 2-bit pass-1 scaling), triangle-filter ("fancy") chroma upsampling and the 16-bit fixed-point YCbCr->RGB tables that
// the IJG / libjpeg-turbo decoders use by default, so that the pixels agree with the witness the tests have (PIL).
#include <hip/hip_runtime.h>

#include <cstdlib>
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

// ---- the 8x8 inverse DCT of 32 blocks by one 256-thread workgroup: 8 lanes per block -------------------------------------------
// (round 4; the first version gave every thread a whole block: 64 two-byte loads per thread at a lane stride of 128 bytes - a wave
// touched 64 cache lines per load instruction - and 64 live ints per thread.)  Thread t = 8 * b + i.  Lane (b, i):
//   1. reads ROW i of its block with one 16-byte load (a wave reads eight whole blocks = 1 KiB contiguous), dequantises it and
//      writes it to the block's LDS image;
//   2. reads COLUMN i of the image, runs pass 1 (columns, as the reference IDCT does first) and writes the column back;
//   3. reads ROW i, runs pass 2, level-shifts and clamps: eight samples of row i of the block.
// LDS image of a block: 72 dwords, element (y, x) at 9 * y + x.  With that pitch the row accesses and the column accesses of the 32
// lanes that share an LDS cycle (4 blocks x 8 lanes, ds_read_b32 / ds_write_b32: bank = dword address mod 32) all fall on 32 different
// banks.  The 8 lanes of a block sit in one wave, whose LDS operations execute in order: no barrier between the three steps.
constexpr int kIdctBlocks = 32, kIdctPitch = 72;

// q9: the quantisation table as int at [9 * y + x] (LDS).  Returns the eight samples of row i as two packed dwords.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// row i (= t & 7) of a block's coefficients: issued as early as possible, consumed by idct_rows_of_32_blocks
__device__ __forceinline__ u32x4 load_coef_row(const int16_t* blk, int t) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if (blk) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(blk) + (t & 7));      // (read once)
  return v;
}
__device__ __forceinline__ uint2 idct_rows_of_32_blocks(u32x4 v, const int* q9, int* ws, int t) {
  const int b = t >> 3, i = t & 7;
  int* W = ws + b * kIdctPitch;
  {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int c = static_cast<int>(static_cast<int16_t>((w[x >> 1] >> (16 * (x & 1))) & 0xFFFFu));
      W[9 * i + x] = __mul24(c, q9[9 * i + x]);      // (int16 x uint16: exact in 24-bit operands; the full-rate multiply)
    }
  }
  int in[8], o[8];
#pragma unroll
  for (int y = 0; y < 8; ++y) in[y] = W[9 * y + i];
  idct8(in, o, CONST_BITS - PASS1_BITS, true);
#pragma unroll
  for (int y = 0; y < 8; ++y) W[9 * y + i] = o[y];
#pragma unroll
  for (int x = 0; x < 8; ++x) in[x] = W[9 * i + x];
  idct8(in, o, CONST_BITS + PASS1_BITS + 3, false);
  uint2 r; r.x = 0u; r.y = 0u;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    r.x |= static_cast<uint32_t>(min(max(o[x] + 128, 0), 255)) << (8 * x);
    r.y |= static_cast<uint32_t>(min(max(o[x + 4] + 128, 0), 255)) << (8 * x);
  }
  return r;
}

// (the quantisation table travels BY VALUE in the kernel arguments: 128 bytes, read with scalar loads - as a 128-byte upload per
// component it was three 5 us blit copies in front of every image's reconstruction)
// Components -> sample planes, all of them in ONE launch (a workgroup belongs to one component: wg0[c] = its first workgroup).
// The fused kernel below takes the luma blocks itself, so for a colour image this launch carries the two chroma planes only.
// Up to kIdctComps components per launch - the chroma planes of nine photos are ONE launch behind the Huffman batch instead of nine
// 9 us launches between the images' fused launches (152 bytes of arguments per component: 2.8 KB of the 4 KB a launch may carry).
constexpr int kIdctComps = 18;
struct IdctComp { const int16_t* coef; uint8_t* plane; int blocks_x, n_blocks; uint16_t q[64]; };
struct IdctArgs { IdctComp comp[kIdctComps]; int wg0[kIdctComps + 1]; };      // wg0[c] = first workgroup of component c; wg0[n] = the grid

__global__ __launch_bounds__(256) void ist_jpeg_idct_kernel(const IdctArgs P) {
  __shared__ int ws[kIdctBlocks * kIdctPitch];
  __shared__ int q9[72];
  const int wg = static_cast<int>(blockIdx.x), t = static_cast<int>(threadIdx.x);
  int ci = 0;
  while (ci + 1 < kIdctComps && wg >= P.wg0[ci + 1]) ++ci;                 // (workgroup-uniform: scalar loads of the arguments)
  const IdctComp& A = P.comp[ci];
  const int b = (wg - P.wg0[ci]) * kIdctBlocks + (t >> 3);
  const bool live = b < A.n_blocks;
  const u32x4 cf = load_coef_row(live ? A.coef + static_cast<size_t>(b) * 64 : nullptr, t);
  if (t < 64) q9[9 * (t >> 3) + (t & 7)] = A.q[t];
  __syncthreads();
  const uint2 v = idct_rows_of_32_blocks(cf, q9, ws, t);
  if (!live) return;
  const int by = b / A.blocks_x, bx = b - by * A.blocks_x;
  *reinterpret_cast<uint2*>(A.plane + (static_cast<size_t>(by) * 8 + (t & 7)) * (static_cast<size_t>(A.blocks_x) * 8) + static_cast<size_t>(bx) * 8) = v;
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
  const int16_t* coef_y; int blocks_x, blocks_y;   // luma coefficient blocks (blocks_x per block row)
  const uint8_t* Cb; const uint8_t* Cr;
  int pitch_c;                     // chroma plane row pitch
  int width, height;               // image size
  int cw, chh;                     // true chroma plane size (ceil(width*1/hmax), ceil(height*1/vmax))
  int hs, vs;                      // luma-to-chroma ratios (1 or 2)
  int ncomp;
  uint8_t* out; size_t out_pitch;
  int exp;                         // 0 in production.  IST_TUNING=1 IST_JPEG_EXP=bits: ablations for the kernel trace (1: no global loads, 2: no IDCT arithmetic, 4: no colour stage / stores) - wrong pixels, right clock
  uint16_t q[64];                  // luma quantisation table
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ uint32_t ycc_to_rgba(int Yv, int cb, int cr) {
  // 16-bit fixed point: FIX(1.40200)=91881, FIX(1.77200)=116130, FIX(0.71414)=46802, FIX(0.34414)=22554
  const int r = Yv + ((91881 * cr + 32768) >> 16);
  const int b = Yv + ((116130 * cb + 32768) >> 16);
  const int g = Yv + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
  return static_cast<uint32_t>(clampi(r, 0, 255)) | (static_cast<uint32_t>(clampi(g, 0, 255)) << 8) |
         (static_cast<uint32_t>(clampi(b, 0, 255)) << 16) | 0xFF000000u;
}

// ---- luma IDCT + chroma upsampling + colour conversion, fused per tile of one MCU row (round 4) --------------------------------
// A workgroup owns 32 luma blocks = one MCU row x 128 pixels (two block rows: v = 2) or x 256 pixels (one block row): it runs their
// inverse DCT (above) into an LDS tile - the luma plane never exists in HBM -, stages the chroma samples the tile's pixels blend
// (the tile's own + one row above and below + one column left and right, clamped to the plane: the edge rules of the triangle
// filter are clamps) from the chroma planes into LDS, and converts.  Traffic per 4:2:0 pixel: 2 B of luma coefficients + 0.5 B of
// chroma samples (+ halo, from L2) in, 4 B of RGBA out; the chroma planes cost their own 1 B + 0.5 B in the launch before: 8 B in
// all against 10 B of the unfused pair (coefficients 3, planes out and in 1.5 + 1.5, RGBA 4), and every global access is a whole
// 16-byte lane.  The arithmetic is unchanged (bit-exact against libjpeg-turbo's defaults, tests/test_gpu_jpeg.py).
constexpr bool kFusedHalfTiles = false;   // 16 instead of 32 luma blocks per workgroup of the fused kernel (IST_JPEG_NB=16|32 overrides in tuning mode)
constexpr int kChromaRows = 10;                                      // chroma tile rows j0-1 .. j0+8; its columns start at i0-4 (dword aligned origin)

template <int HS, int VS, int CP>
__device__ __forceinline__ int chroma_lds(const uint8_t* T, int cw, int i_org, int j_org, int x, int y) {
  // T[(j - j_org) * CP + (i - i_org)]: rows are clamped at load time, columns here
  auto at = [&](int j, int i) { return static_cast<int>(T[(j - j_org) * CP + (clampi(i, 0, cw - 1) - i_org)]); };
  if (HS == 1 && VS == 1) return at(y, x);
  if (HS == 2 && VS == 1) {                       // h2v1: 3/4 nearer + 1/4 further column
    const int i = x >> 1;
    if (x & 1) return i == cw - 1 ? at(y, i) : (3 * at(y, i) + at(y, i + 1) + 2) >> 2;
    return i == 0 ? at(y, i) : (3 * at(y, i) + at(y, i - 1) + 1) >> 2;
  }
  if (HS == 1 && VS == 2) {                       // h1v2: 3/4 nearer + 1/4 further row
    const int j = y >> 1, jn = (y & 1) ? j + 1 : j - 1;
    return (3 * at(j, x) + at(jn, x) + ((y & 1) ? 2 : 1)) >> 2;
  }
  // h2v2: vertical 3:1 first (unscaled), then horizontal 3:1, one final shift by 4; a clamped neighbour column reproduces the edge
  // rule: (4 c + 8) >> 4 = (3 c + c + 8) >> 4
  const int j = y >> 1, i = x >> 1, jn = (y & 1) ? j + 1 : j - 1;
  const int cur = 3 * at(j, i) + at(jn, i);
  const int in = (x & 1) ? i + 1 : i - 1;
  return (cur * 3 + (3 * at(j, in) + at(jn, in)) + ((x & 1) ? 7 : 8)) >> 4;
}

// NB = luma blocks per workgroup (8 lanes each): 32 (256 threads) or 16 (128 threads, half as wide a tile).
template <int HS, int VS, bool COLOUR, int NB>
__global__ __launch_bounds__(NB * 8) void ist_jpeg_fused_kernel(const ColorArgs A) {
  constexpr int NT = NB * 8;                                             // threads
  constexpr int TW = NB * 8 / VS, TH = 8 * VS, YP = TW + 16;             // tile; bytes per luma tile row: 16 mod 128 keeps the row writes of step 3 off each other's banks
  constexpr int BW = TW / 8;                                             // luma blocks per block row of the tile
  constexpr int CP = ((TW / HS + 8) + 15) & ~15;                         // bytes per chroma tile row (the tile's columns + 4 of halo room on each side)
  __shared__ int ws[NB * kIdctPitch];
  __shared__ int q9[72];
  __shared__ __attribute__((aligned(16))) uint8_t Ys[TH * YP];
  __shared__ __attribute__((aligned(16))) uint8_t Cs[COLOUR ? 2 * kChromaRows * CP : 16];
  const int t = static_cast<int>(threadIdx.x);
  const int x_org = static_cast<int>(blockIdx.x) * TW, y_org = static_cast<int>(blockIdx.y) * TH;
  // the tile's coefficient rows are requested first: their trip to HBM runs beside the chroma tile's (one exposed latency per
  // workgroup instead of two: 8 workgroups per CU live about as long as their loads take)
  const int blk_b = t >> 3;
  const int blk_x = x_org / 8 + (blk_b % BW), blk_y = y_org / 8 + (blk_b / BW);
  const u32x4 cf = load_coef_row(blk_x < A.blocks_x && blk_y < A.blocks_y && !(A.exp & 1) ? A.coef_y + static_cast<uint32_t>((blk_y * A.blocks_x + blk_x) * 64) : nullptr, t);
  if (t < 64) q9[9 * (t >> 3) + (t & 7)] = A.q[t];
  // the chroma tile: rows j_org .. j_org+9 (clamped to the plane), columns from i_org in dwords
  const int i_org = x_org / HS - 4, j_org = y_org / VS - 1;
  if (COLOUR) {
    // dword d of tile row pr (= plane * 10 + row): 64 lanes x 4 row slots per pass.  Columns outside the plane are REPLICATED into
    // the tile (a dword that straddles an edge is put together from clamped bytes), so the blends below never clamp.
    constexpr int DW = (TW / HS + 8) / 4;                                // dwords per row: 4 columns of halo room on each side (<= 66)
    for (int d = t & 63; d < DW; d += 64) {
      const int i = i_org + 4 * d;
      for (int pr = t >> 6; pr < 2 * kChromaRows; pr += NT / 64) {
        const int pl = pr >= kChromaRows ? 1 : 0, r = pr - pl * kChromaRows;
        const uint8_t* row = (pl ? A.Cr : A.Cb) + static_cast<uint32_t>(clampi(j_org + r, 0, A.chh - 1) * A.pitch_c);      // (a plane is < 4 GB: 32-bit offsets)
        uint32_t v;
        if (A.exp & 1) v = 0x80808080u;
        else if (i >= 0 && i + 3 <= A.cw - 1) v = *reinterpret_cast<const uint32_t*>(row + i);
        else {
          v = 0u;
#pragma unroll
          for (int k = 0; k < 4; ++k) v |= static_cast<uint32_t>(row[clampi(i + k, 0, A.cw - 1)]) << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(Cs + pr * CP + 4 * d) = v;
      }
    }
  }
  __syncthreads();
  {
    uint2 v;
    if (A.exp & 2) { v.x = cf.x ^ cf.y; v.y = cf.z ^ cf.w; }
    else v = idct_rows_of_32_blocks(cf, q9, ws, t);
    *reinterpret_cast<uint2*>(Ys + ((blk_b / BW) * 8 + (t & 7)) * YP + (blk_b % BW) * 8) = v;
  }
  __syncthreads();
  if (A.exp & 4) { if (t == 0 && Ys[5] == 77 && Cs[3] == 99) A.out[0] = 1; return; }
  if (COLOUR && HS == 2 && VS == 2) {
    // 4:2:0, the photo case: a thread converts 4 pixels x 2 rows (the rows 2j, 2j+1 that share chroma row j): per plane ONE
    // two-dword LDS read per chroma row j-1, j, j+1 gives the four columns i-1 .. i+2 the eight pixels blend; the vertical 3:1 and the
    // horizontal 3:1 run on pairs of 16-bit lanes in 32-bit registers (every term < 4096), the colour conversion on 24-bit
    // multiply-adds (chroma < 256, constants < 2^17: exact).  Same integers as chroma_lds + ycc_to_rgba, a third of the instructions
    // (the kernel is bound by instruction issue: rocprofv3 + ISA count).
    constexpr int GXF = TW / 4;                                          // groups of 4 pixels across the tile (32 or 16)
    const int gx = t & (GXF - 1), jj = t / GXF;
    const int lx = 4 * gx, ly = 2 * jj;
    const int x0 = x_org + lx, y = y_org + ly;
    if (x0 >= A.width || y >= A.height) return;
    const int cb0 = 2 * gx + 3;                                          // tile column of chroma i - 1 (i = x0 / 2; the tile starts at i_org = x_org / 2 - 4)
    const int al = cb0 & ~3, sh = 8 * (cb0 & 3);
    uint32_t pcb[4], pcr[4];                                            // per output row r = 0, 1: [2r] = pixels 0,1 and [2r+1] = pixels 2,3, as (value << 4 | fraction) in 16-bit lanes
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      uint32_t lo[3], hi[3];                                            // columns (c0, c2) and (c1, c3) of rows j-1, j, j+1 in 16-bit lanes
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(Cs + (pl * kChromaRows + jj + r) * CP + al);
        const uint32_t w = static_cast<uint32_t>((static_cast<uint64_t>(q[0]) | (static_cast<uint64_t>(q[1]) << 32)) >> sh);
        lo[r] = w & 0x00FF00FFu; hi[r] = (w >> 8) & 0x00FF00FFu;
      }
      const uint32_t l3 = lo[1] * 3u, h3 = hi[1] * 3u;                   // (lanes <= 765: no carry between them)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const uint32_t vl = l3 + lo[2 * r], vh = h3 + hi[2 * r];         // vertical: 3 * row j + row j -/+ 1 -> (c0, c2), (c1, c3), each <= 1020
        const uint32_t b1 = __umul24(vh & 0xFFFFu, 0x00030003u), b2 = __umul24(vl >> 16, 0x00030003u);      // 3 * c1, 3 * c2 in both lanes
        const uint32_t p01 = b1 + vl + 0x00070008u, p23 = b2 + vh + 0x00070008u;      // pixel 0: 3 c1 + c0 + 8, 1: 3 c1 + c2 + 7, 2: 3 c2 + c1 + 8, 3: 3 c2 + c3 + 7
        if (pl == 0) { pcb[2 * r] = p01; pcb[2 * r + 1] = p23; } else { pcr[2 * r] = p01; pcr[2 * r + 1] = p23; }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (y + r >= A.height) break;
      const uint32_t yy = *reinterpret_cast<const uint32_t*>(Ys + (ly + r) * YP + lx);
      uint32_t px[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int cb = static_cast<int>((pcb[2 * r + (k >> 1)] >> (4 + 16 * (k & 1))) & 255u);      // (>> 4: the blend's final shift)
        const int cr = static_cast<int>((pcr[2 * r + (k >> 1)] >> (4 + 16 * (k & 1))) & 255u);
        const int y16 = static_cast<int>(__builtin_amdgcn_perm(0x00008000u, yy, 0x0c000504u | (static_cast<uint32_t>(k) << 16)));   // Y << 16 + the rounding half: byte 2 = Y_k, bytes 1, 0 = 0x80, 0x00
        // Y + ((K * (c - 128) + 32768) >> 16) = (Y << 16 + 32768 + K * c - 128 K) >> 16
        // clamped BEFORE the shift (to [0, 2^24): byte 2 is the channel) and put together with a byte permute.  Written as
        // clamp(x >> 16, 0, 255) the compiler (ROCm 7.2) forms v_ashr_pk_u8_i32, whose destination keeps its upper 16 bits - it then
        // ORs the blue byte onto whatever the register held before (seen on the GPU: blue = right value | junk, red and green right).
        const uint32_t r_ = static_cast<uint32_t>(clampi(__mul24(cr, 91881) + (y16 - 128 * 91881), 0, 0x00FFFFFF));
        const uint32_t b_ = static_cast<uint32_t>(clampi(__mul24(cb, 116130) + (y16 - 128 * 116130), 0, 0x00FFFFFF));
        const uint32_t g_ = static_cast<uint32_t>(clampi(__mul24(cb, -22554) + __mul24(cr, -46802) + (y16 + 128 * (22554 + 46802)), 0, 0x00FFFFFF));
        px[k] = __builtin_amdgcn_perm(g_, r_, 0x0c0c0602u) | (b_ & 0x00FF0000u) | 0xFF000000u;      // byte 0 = r[2], byte 1 = g[2]
      }
      // (the tile's first row on the scalar unit; a row pitch is < 2^24: the rows inside the tile with a 24-bit multiply)
      uint8_t* o = A.out + static_cast<size_t>(y_org) * A.out_pitch + static_cast<uint32_t>(x0) * 4u;
      if (A.out_pitch < (size_t{1} << 24)) o += __umul24(static_cast<uint32_t>(ly + r), static_cast<uint32_t>(A.out_pitch));      // (wave-uniform choice)
      else o += static_cast<size_t>(ly + r) * A.out_pitch;
      const int nv = A.width - x0;
      if (nv >= 4 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
        uint4 v; v.x = px[0]; v.y = px[1]; v.z = px[2]; v.w = px[3];
        *reinterpret_cast<uint4*>(o) = v;
      } else {
        for (int k = 0; k < 4 && k < nv; ++k) *reinterpret_cast<uint32_t*>(o + 4 * k) = px[k];
      }
    }
    return;
  }
  // colour, the other samplings: groups of 4 pixels; a wave's 64 groups are consecutive in x (512 B - 1 KiB runs of a canvas row)
  constexpr int GX = TW / 4;
  for (int g = t; g < GX * TH; g += NT) {
    const int ly = g / GX, lx = 4 * (g % GX);
    const int x0 = x_org + lx, y = y_org + ly;
    if (x0 >= A.width || y >= A.height) continue;
    const uint32_t yy = *reinterpret_cast<const uint32_t*>(Ys + ly * YP + lx);
    uint32_t px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int Yv = static_cast<int>((yy >> (8 * k)) & 255u);
      if (COLOUR) {
        const int x = min(x0 + k, A.width - 1);
        const int cb = chroma_lds<HS, VS, CP>(Cs, A.cw, i_org, j_org, x, y) - 128;
        const int cr = chroma_lds<HS, VS, CP>(Cs + kChromaRows * CP, A.cw, i_org, j_org, x, y) - 128;
        px[k] = ycc_to_rgba(Yv, cb, cr);
      } else {
        px[k] = static_cast<uint32_t>(Yv) * 0x010101u | 0xFF000000u;
      }
    }
    uint8_t* o = A.out + static_cast<size_t>(y) * A.out_pitch + static_cast<size_t>(x0) * 4;
    const int nv = A.width - x0;
    if (nv >= 4 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
      uint4 v; v.x = px[0]; v.y = px[1]; v.z = px[2]; v.w = px[3];
      *reinterpret_cast<uint4*>(o) = v;
    } else {
      for (int k = 0; k < 4 && k < nv; ++k) *reinterpret_cast<uint32_t*>(o + 4 * k) = px[k];
    }
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

int jpeg_launch_chroma_idct(const JpegDeviceJob* jobs, int n_jobs, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  IdctArgs a;
  int nc = 0, wgs = 0;
  auto flush = [&]() -> int {
    if (nc == 0) return IST_OK;
    for (int c = nc; c <= kIdctComps; ++c) a.wg0[c] = c == nc ? wgs : 0x7fffffff;
    hipLaunchKernelGGL(ist_jpeg_idct_kernel, dim3(static_cast<unsigned>(wgs)), dim3(256), 0, stream, a);
    const hipError_t e = hipGetLastError();
    nc = 0; wgs = 0;
    if (e != hipSuccess) return fail(IST_E_HIP, std::string("JPEG chroma IDCT launch failed: ") + hipGetErrorString(e));
    return IST_OK;
  };
  std::memset(&a, 0, sizeof a);
  for (int k = 0; k < n_jobs; ++k) {
    const JpegDeviceJob& J = jobs[k];
    if (J.ncomp != 3) continue;
    for (int c = 1; c < 3; ++c) {
      const int n_blocks = J.blocks_x[c] * J.blocks_y[c];
      if (n_blocks <= 0) continue;
      IdctComp& C = a.comp[nc];
      a.wg0[nc] = wgs;
      C.coef = J.d_coef[c]; C.plane = J.d_plane[c]; C.blocks_x = J.blocks_x[c]; C.n_blocks = n_blocks;
      for (int q = 0; q < 64; ++q) C.q[q] = J.q_host[c] ? J.q_host[c][q] : 1;
      wgs += (n_blocks + kIdctBlocks - 1) / kIdctBlocks;
      if (++nc == kIdctComps) { const int rc = flush(); if (rc) return rc; }
    }
  }
  return flush();
}

int jpeg_launch_reconstruct(const JpegDeviceJob& J, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (J.ncomp == 3 && !J.chroma_done) {                // the chroma planes (the fused kernel blends their neighbours across tiles)
    const int rc = jpeg_launch_chroma_idct(&J, 1, stream_);
    if (rc) return rc;
  }
  ColorArgs ca;
  std::memset(&ca, 0, sizeof ca);
  ca.coef_y = J.d_coef[0]; ca.blocks_x = J.blocks_x[0]; ca.blocks_y = J.blocks_y[0];
  ca.Cb = J.ncomp == 3 ? J.d_plane[1] : nullptr; ca.Cr = J.ncomp == 3 ? J.d_plane[2] : nullptr;
  ca.pitch_c = J.ncomp == 3 ? J.blocks_x[1] * 8 : 0;
  ca.width = J.width; ca.height = J.height;
  ca.hs = J.hmax; ca.vs = J.vmax;
  ca.cw = (J.width + J.hmax - 1) / J.hmax; ca.chh = (J.height + J.vmax - 1) / J.vmax;
  ca.ncomp = J.ncomp; ca.out = J.out; ca.out_pitch = J.out_pitch;
  for (int k = 0; k < 64; ++k) ca.q[k] = J.q_host[0] ? J.q_host[0][k] : 1;
  static const int exp = (tuning_mode() && std::getenv("IST_JPEG_EXP")) ? std::atoi(std::getenv("IST_JPEG_EXP")) : 0;
  ca.exp = exp;
  static const int nb_knob = (tuning_mode() && std::getenv("IST_JPEG_NB")) ? std::atoi(std::getenv("IST_JPEG_NB")) : 0;      // A/B: blocks per workgroup
  const bool half = nb_knob ? nb_knob == 16 : kFusedHalfTiles;
  const int nb = half ? 16 : 32;
  const int tw = nb * 8 / J.vmax, th = 8 * J.vmax;
  const dim3 grid(static_cast<unsigned>((J.blocks_x[0] * 8 + tw - 1) / tw), static_cast<unsigned>((J.blocks_y[0] * 8 + th - 1) / th));
  if (grid.x > 0 && grid.y > 0) {
#define IST_FUSED(HS, VS, C) do { if (half) hipLaunchKernelGGL((ist_jpeg_fused_kernel<HS, VS, C, 16>), grid, dim3(128), 0, stream, ca); \
                                  else hipLaunchKernelGGL((ist_jpeg_fused_kernel<HS, VS, C, 32>), grid, dim3(256), 0, stream, ca); } while (0)
    if (J.ncomp != 3) IST_FUSED(1, 1, false);
    else if (J.hmax == 2 && J.vmax == 2) IST_FUSED(2, 2, true);
    else if (J.hmax == 2) IST_FUSED(2, 1, true);
    else if (J.vmax == 2) IST_FUSED(1, 2, true);
    else IST_FUSED(1, 1, true);
#undef IST_FUSED
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(IST_E_HIP, std::string("JPEG reconstruct launch failed: ") + hipGetErrorString(e));
  return IST_OK;
}

}  // namespace ist
