// Cache-policy bits of the copy's loads and stores on gfx950 (tools/exp: measurement aid, not product):
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/cache_bits.bin tools/exp/cache_bits.cpp && tools/exp/cache_bits.bin
// The flat copy of hbm_ceiling.cpp (eight 1 KiB wave-rows 32 KiB apart, two in flight per wave), with every combination of the sc0 / sc1 / nt
// modifiers on global_load_dwordx4 and global_store_dwordx4.  The product uses __builtin_nontemporal_load / _store (= nt on both).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define LD(name, mods) __device__ inline u32x4 name(const uint8_t* p) { u32x4 v; asm volatile("global_load_dwordx4 %0, %1, off " mods : "=v"(v) : "v"(p) : "memory"); return v; }
#define ST(name, mods) __device__ inline void name(uint8_t* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off " mods : : "v"(p), "v"(v) : "memory"); }
LD(ld0, "") LD(ld1, "nt") LD(ld2, "sc0") LD(ld3, "sc0 nt") LD(ld4, "sc1") LD(ld5, "sc1 nt") LD(ld6, "sc0 sc1") LD(ld7, "sc0 sc1 nt")
ST(st0, "") ST(st1, "nt") ST(st2, "sc0") ST(st3, "sc0 nt") ST(st4, "sc1") ST(st5, "sc1 nt") ST(st6, "sc0 sc1") ST(st7, "sc0 sc1 nt")
static const char* kMods[8] = {"(none)", "nt", "sc0", "sc0 nt", "sc1", "sc1 nt", "sc0 sc1", "sc0 sc1 nt"};
template <int L> __device__ inline u32x4 ld(const uint8_t* p) {
  if (L == 0) return ld0(p); if (L == 1) return ld1(p); if (L == 2) return ld2(p); if (L == 3) return ld3(p);
  if (L == 4) return ld4(p); if (L == 5) return ld5(p); if (L == 6) return ld6(p); return ld7(p);
}
template <int S> __device__ inline void st(uint8_t* p, u32x4 v) {
  if (S == 0) st0(p, v); else if (S == 1) st1(p, v); else if (S == 2) st2(p, v); else if (S == 3) st3(p, v);
  else if (S == 4) st4(p, v); else if (S == 5) st5(p, v); else if (S == 6) st6(p, v); else st7(p, v);
}
template <int L, int S>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t bytes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t stride = 32768, per = 32;
  const size_t block = blockIdx.x / per, col = blockIdx.x - block * per;
  u32x4 v[2];
  size_t o[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    o[u] = block * 8 * stride + static_cast<size_t>(u * 4 + wave) * stride + col * 1024 + static_cast<size_t>(lane) * 16;
    if (o[u] + 16 <= bytes) v[u] = ld<L>(src + o[u]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (o[u] + 16 <= bytes) st<S>(dst + o[u], v[u]);
}
typedef void (*Kern)(const uint8_t*, uint8_t*, size_t);
template <int L, int S> struct Fill { static void go(Kern* t) { t[L * 8 + S] = k<L, S>; Fill<L, S - 1>::go(t); } };
template <int L> struct Fill<L, -1> { static void go(Kern* t) { Fill<L - 1, 7>::go(t); } };
template <> struct Fill<-1, 7> { static void go(Kern*) {} };
int main() {
  const size_t bytes = 438939648, cap = bytes + (8u << 20);
  uint8_t *a, *b;
  CK(hipMalloc(&a, cap)); CK(hipMalloc(&b, cap));
  CK(hipMemset(a, 0x5A, cap)); CK(hipMemset(b, 0, cap));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Kern tab[64];
  Fill<7, 7>::go(tab);
  const unsigned grid = static_cast<unsigned>((bytes + 8 * 32768 - 1) / (8 * 32768) * 32);
  for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(tab[9], dim3(grid), dim3(256), 0, 0, a, b, bytes);
  CK(hipDeviceSynchronize());
  for (int L = 0; L < 8; ++L)
    for (int S = 0; S < 8; ++S) {
      std::vector<float> ts;
      for (int r = 0; r < 5; ++r) {
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(tab[L * 8 + S], dim3(grid), dim3(256), 0, 0, a, b, bytes);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(tab[L * 8 + S], dim3(grid), dim3(256), 0, 0, a, b, bytes);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms * 1000.0f / 30);
      }
      std::sort(ts.begin(), ts.end());
      std::printf("load %-10s store %-10s: %7.1f us  %.3f of 8 TB/s\n", kMods[L], kMods[S], ts[2], 2.0 * bytes / (ts[2] * 1e-6) / 8e12);
      std::fflush(stdout);
    }
  return 0;
}
