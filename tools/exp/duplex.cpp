// Can this box move data host -> device and device -> host at the same time?  (tools/exp: measurement aid, not product.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/duplex.bin tools/exp/duplex.cpp && tools/exp/duplex.bin
// 439 MB each way (the headline's bytes), pinned host blocks.  The host path (ist_stitch_rgba8) uploads everything, launches, downloads
// everything: 8.1 + 7.5 ms.  An earlier attempt to overlap the two directions with runtime copies on two streams fell to ~16 GB/s in one
// direction (ist_runtime.cpp, comment in ist_stitch_rgba8).  Here: each direction alone, both as runtime copies, and the download done by a
// KERNEL that stores into the pinned block (the block is mapped into the device's address space) while the copy engine uploads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
__global__ __launch_bounds__(256) void push(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256ull) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = 438939648, piece = 48771072;       // nine pieces, as the nine images / bands
  void *hu, *hd, *du, *dd;
  CK(hipHostMalloc(&hu, bytes, hipHostMallocPortable)); CK(hipHostMalloc(&hd, bytes, hipHostMallocPortable));
  CK(hipMalloc(&du, bytes)); CK(hipMalloc(&dd, bytes));
  std::memset(hu, 0x5A, bytes); std::memset(hd, 0, bytes);
  CK(hipMemset(dd, 0x33, bytes));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  void* hd_dev = nullptr;
  CK(hipHostGetDevicePointer(&hd_dev, hd, 0));
  auto up = [&]() { for (int k = 0; k < 9; ++k) CK(hipMemcpyAsync((char*)du + k * piece, (char*)hu + k * piece, piece, hipMemcpyHostToDevice, a)); };
  auto down = [&]() { for (int k = 0; k < 9; ++k) CK(hipMemcpyAsync((char*)hd + k * piece, (char*)dd + k * piece, piece, hipMemcpyDeviceToHost, b)); };
  auto down_kernel = [&](int grid) { for (int k = 0; k < 9; ++k) hipLaunchKernelGGL(push, dim3(grid), dim3(256), 0, b, (const u32x4*)((char*)dd + k * piece), (u32x4*)((char*)hd_dev + k * piece), piece / 16); };
  for (int rep = 0; rep < 3; ++rep) {
    double t0, t;
    CK(hipDeviceSynchronize()); t0 = now_ms(); up(); CK(hipDeviceSynchronize()); t = now_ms() - t0;
    if (rep) std::printf("upload alone (runtime copies):            %6.2f ms  %5.1f GB/s\n", t, bytes / t / 1e6);
    CK(hipDeviceSynchronize()); t0 = now_ms(); down(); CK(hipDeviceSynchronize()); t = now_ms() - t0;
    if (rep) std::printf("download alone (runtime copies):          %6.2f ms  %5.1f GB/s\n", t, bytes / t / 1e6);
    for (int grid : {64, 256, 1024}) {
      CK(hipDeviceSynchronize()); t0 = now_ms(); down_kernel(grid); CK(hipDeviceSynchronize()); t = now_ms() - t0;
      if (rep) std::printf("download alone (kernel, %4d workgroups): %6.2f ms  %5.1f GB/s\n", grid, t, bytes / t / 1e6);
    }
    CK(hipDeviceSynchronize()); t0 = now_ms(); up(); down(); CK(hipDeviceSynchronize()); t = now_ms() - t0;
    if (rep) std::printf("both, runtime copies on two streams:      %6.2f ms  %5.1f GB/s each way\n", t, bytes / t / 1e6);
    for (int grid : {64, 256, 1024}) {
      CK(hipDeviceSynchronize()); t0 = now_ms(); up(); down_kernel(grid); CK(hipDeviceSynchronize()); t = now_ms() - t0;
      if (rep) std::printf("both, upload by copies + download by kernel (%4d workgroups): %6.2f ms  %5.1f GB/s each way\n", grid, t, bytes / t / 1e6);
    }
    // the product's upload: 4 MiB chunks of a pinned ring on four lane streams (here straight from the pinned block: no CPU packing) ...
    {
      static hipStream_t lane[4]; static bool made = false;
      if (!made) { for (auto& l : lane) CK(hipStreamCreateWithFlags(&l, hipStreamNonBlocking)); made = true; }
      const size_t chunk = 4u << 20;
      auto up_chunks = [&]() { size_t o = 0; int k = 0; while (o < bytes) { const size_t n = bytes - o < chunk ? bytes - o : chunk; CK(hipMemcpyAsync((char*)du + o, (char*)hu + o, n, hipMemcpyHostToDevice, lane[k & 3])); o += n; ++k; } };
      CK(hipDeviceSynchronize()); t0 = now_ms(); up_chunks(); CK(hipDeviceSynchronize()); t = now_ms() - t0;
      if (rep) std::printf("upload alone, 4 MiB chunks on four streams: %6.2f ms  %5.1f GB/s\n", t, bytes / t / 1e6);
      CK(hipDeviceSynchronize()); t0 = now_ms(); up_chunks(); down(); CK(hipDeviceSynchronize()); t = now_ms() - t0;
      if (rep) std::printf("both, chunked upload + nine download copies: %6.2f ms  %5.1f GB/s each way\n", t, bytes / t / 1e6);
      // ... and with the CPU packing in front of it: four threads copy pageable memory into the ring's chunks, each chunk uploaded when packed
      static char* pageable = nullptr;
      if (!pageable) { pageable = static_cast<char*>(std::malloc(bytes)); std::memset(pageable, 0x77, bytes); }
      auto packed_upload = [&](bool with_down) {
        std::vector<std::thread> th;
        const size_t n_chunks = (bytes + chunk - 1) / chunk;
        std::atomic<size_t> next{0};
        if (with_down) down();
        for (int w = 0; w < 4; ++w) th.emplace_back([&, w]() {
          CK(hipSetDevice(0));
          for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= n_chunks) break;
            const size_t o = c * chunk, n = bytes - o < chunk ? bytes - o : chunk;
            std::memcpy((char*)hu + o, pageable + o, n);            // (a ring would reuse 8 chunks; a fresh region per chunk costs the same bandwidth)
            CK(hipMemcpyAsync((char*)du + o, (char*)hu + o, n, hipMemcpyHostToDevice, lane[w]));
          }
        });
        for (auto& t2 : th) t2.join();
      };
      CK(hipDeviceSynchronize()); t0 = now_ms(); packed_upload(false); CK(hipDeviceSynchronize()); t = now_ms() - t0;
      if (rep) std::printf("upload alone, packed by four threads:        %6.2f ms  %5.1f GB/s\n", t, bytes / t / 1e6);
      CK(hipDeviceSynchronize()); t0 = now_ms(); packed_upload(true); CK(hipDeviceSynchronize()); t = now_ms() - t0;
      if (rep) std::printf("both, packed upload + nine download copies:  %6.2f ms  %5.1f GB/s each way\n", t, bytes / t / 1e6);
    }
    if (rep) std::printf("--\n");
  }
  return static_cast<const unsigned char*>(hd)[12345] == 0x33 ? 0 : 1;
}
