// Quarter-turn (transpose) of RGBA8 images at 1:1, tile shapes compared in isolation (a debugging aid for SWAP_LDS).
//   dst(X, Y) = src(x = Y, y = X) ... canvas X drives the source row, canvas Y the source column (EXIF 5-like; the flips
//   of 6 / 8 do not change the access pattern).  hipcc -O3 --offload-arch=gfx950 transpose.cpp -o transpose
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

// TX x TY canvas tile per workgroup of 256 threads; source patch = TX rows x TY columns.
// LOADV: pixels per load (1 or 4); STOREV: pixels per store (1 or 4)
template <int TX, int TY, int STOREV>
__global__ __launch_bounds__(256) void tr(const u32* __restrict__ src, u32* __restrict__ dst, int sw, int sh) {
  // canvas: width = sh, height = sw
  extern __shared__ u32 lds[];
  constexpr int P = TX + 1;                        // T[col][row], pitch TX + 1
  const int tiles_x = (sh + TX - 1) / TX;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int X0 = tx * TX, Y0 = ty * TY;            // source rows X0.., source cols Y0..
  const int tid = threadIdx.x;
  // stage: chunk = 4 source pixels of one source row
  constexpr int CH = TY / 4;
  for (int i = tid; i < TX * CH; i += 256) {
    const int r = i / CH, c = i - r * CH;
    const int sy = X0 + r, sx = Y0 + 4 * c;
    u32x4 v = {0, 0, 0, 0};
    if (sy < sh && sx + 3 < sw) v = *reinterpret_cast<const u32x4*>(src + (size_t)sy * sw + sx);
    lds[(4 * c) * P + r] = v.x; lds[(4 * c + 1) * P + r] = v.y; lds[(4 * c + 2) * P + r] = v.z; lds[(4 * c + 3) * P + r] = v.w;
  }
  __syncthreads();
  if (STOREV == 1) {
    for (int i = tid; i < TX * TY; i += 256) {
      const int y = i / TX, x = i - y * TX;
      const int X = X0 + x, Y = Y0 + y;
      if (X < sh && Y < sw) __builtin_nontemporal_store(lds[y * P + x], dst + (size_t)Y * sh + X);
    }
  } else {
    for (int i = tid; i < TX * TY / 4; i += 256) {
      const int y = i / (TX / 4), x = 4 * (i - y * (TX / 4));
      const int X = X0 + x, Y = Y0 + y;
      if (X + 3 < sh && Y < sw) {
        u32x4 v = {lds[y * P + x], lds[y * P + x + 1], lds[y * P + x + 2], lds[y * P + x + 3]};
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + (size_t)Y * sh + X));
      }
    }
  }
}

template <int TX, int TY, int STOREV>
void run(const char* name, const std::vector<u32*>& src, const std::vector<u32*>& dst, int sw, int sh) {
  const int tiles = ((sh + TX - 1) / TX) * ((sw + TY - 1) / TY);
  const size_t lds = (size_t)(TX + 1) * TY * 4;
  hipFuncSetAttribute((const void*)tr<TX, TY, STOREV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    for (int it = 0; it < 20; ++it)
      for (size_t k = 0; k < src.size(); ++k) hipLaunchKernelGGL((tr<TX, TY, STOREV>), dim3(tiles), dim3(256), lds, 0, src[k], dst[k], sw, sh);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    if (rep == 2) std::printf("%-28s lds %6zu B  %8.1f us per 9 images  %.2f TB/s\n", name, lds, ms * 1e3 / 20, 2.0 * src.size() * sw * sh * 4 / (ms / 20 * 1e-3) / 1e12);
  }
}

int main() {
  const int sw = 4032, sh = 3024, n = 9;
  std::vector<u32*> src(n), dst(n);
  for (int k = 0; k < n; ++k) { hipMalloc((void**)&src[k], (size_t)sw * sh * 4); hipMalloc((void**)&dst[k], (size_t)sw * sh * 4); hipMemset(src[k], k + 1, (size_t)sw * sh * 4); }
  run<64, 64, 1>("64x64 st4", src, dst, sw, sh);
  run<64, 64, 4>("64x64 st16", src, dst, sw, sh);
  run<128, 64, 1>("128x64 st4", src, dst, sw, sh);
  run<128, 64, 4>("128x64 st16", src, dst, sw, sh);
  run<64, 128, 4>("64x128 st16", src, dst, sw, sh);
  run<128, 128, 4>("128x128 st16", src, dst, sw, sh);
  run<256, 64, 4>("256x64 st16", src, dst, sw, sh);
  run<32, 32, 1>("32x32 st4", src, dst, sw, sh);
  run<64, 32, 4>("64x32 st16", src, dst, sw, sh);
  return 0;
}
