// Issue cost of 32-bit integer multiplies on gfx950 (tools/exp: measurement aid, not product):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mulrate tools/exp/mulrate.cpp && /tmp/mulrate
// Four independent dependency chains per lane keep the SIMD busy; 8 waves per SIMD; time per wave-instruction in cycles
// (assuming the clock the chip reports under this load).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 77;
  const uint32_t m = seed | 1;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) { a = a * m + 1; b = b * m + 2; c = c * m + 3; d = d * m + 4; }                                            // v_mul_lo_u32 (+ add)
      if (MODE == 1) { a = __umul24(a, m) + 1; b = __umul24(b, m) + 2; c = __umul24(c, m) + 3; d = __umul24(d, m) + 4; }         // v_mad_u32_u24
      if (MODE == 2) { a = (a + m) ^ 1; b = (b + m) ^ 2; c = (c + m) ^ 3; d = (d + m) ^ 4; }                                      // add + xor
      if (MODE == 3) { a = static_cast<uint32_t>((static_cast<uint64_t>(a) * m) >> 32) + 1; b = static_cast<uint32_t>((static_cast<uint64_t>(b) * m) >> 32) + 2;
                       c = static_cast<uint32_t>((static_cast<uint64_t>(c) * m) >> 32) + 3; d = static_cast<uint32_t>((static_cast<uint64_t>(d) * m) >> 32) + 4; }   // v_mul_hi_u32
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int MODE> double run(uint32_t* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int wgs = 256 * 8;                      // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, d, 12345u, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, d, 12345u, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: 8 waves * iters * 16 * 4 chains * (ops per step)
  return ms;
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 2000;
  const double steps = 8.0 * iters * 16 * 4;     // chain steps per SIMD
  const char* names[4] = {"v_mul_lo_u32 + v_add", "v_mad_u32_u24", "v_add + v_xor", "v_mul_hi_u32 + v_add"};
  double ms[4] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters)};
  for (int m = 0; m < 4; ++m) std::printf("%-24s %8.3f ms  = %6.2f ns per chain step per SIMD (at 2.4 GHz: %5.1f cycles)\n", names[m], ms[m], ms[m] * 1e6 / steps, ms[m] * 1e6 / steps * 2.4);
  return 0;
}
