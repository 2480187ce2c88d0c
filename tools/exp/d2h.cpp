// D2H copy rate into pinned blocks of different sizes, with and without a concurrent kernel (debugging aid).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(float* p, int iters) { float v = p[threadIdx.x]; for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
int main() {
  const size_t piece = 21u << 20;
  void* d = nullptr; float* w = nullptr;
  hipMalloc(&d, 160u << 20); hipMalloc((void**)&w, 4096);
  hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  for (size_t host_mb : {160, 460}) {
    void* h = nullptr; hipHostMalloc(&h, host_mb << 20, hipHostMallocPortable);
    for (int busy = 0; busy < 2; ++busy)
      for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        if (busy) hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, a, w, 200000);
        for (int n = 0; n < 7; ++n) hipMemcpyAsync((char*)h + n * piece, (char*)d + n * piece, piece, hipMemcpyDeviceToHost, b);
        hipStreamSynchronize(b);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        hipDeviceSynchronize();
        const double all = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rep) std::printf("pinned block %zu MiB, %s: 7 x 21 MiB in %.3f ms  %.1f GB/s (kernel + copies %.3f ms)\n", host_mb, busy ? "kernel running" : "idle GPU", ms, 7.0 * piece / ms / 1e6, all);
      }
    hipHostFree(h);
  }
  return 0;
}
