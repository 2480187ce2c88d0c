// Upload pageable memory through TWO pinned staging pieces of X MiB (packed by four threads, ONE runtime copy per piece on one stream) while nine
// 48 MB runtime copies download on another stream: which piece size keeps both directions at full rate?  (tools/exp/duplex.cpp: 4 MiB chunks
// on four streams against concurrent downloads fall to 12.7 GB/s; big copies both ways sustain 48 GB/s each way.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/duplex2.bin tools/exp/duplex2.cpp -lpthread && tools/exp/duplex2.bin
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pack(char* dst, const char* src, size_t n, int threads) {
  std::vector<std::thread> th;
  const size_t per = (n / threads + 4095) & ~static_cast<size_t>(4095);
  for (int w = 0; w < threads; ++w) th.emplace_back([=]() { const size_t o = w * per; if (o < n) std::memcpy(dst + o, src + o, std::min(per, n - o)); });
  for (auto& t : th) t.join();
}
int main() {
  const size_t bytes = 438939648, piece = 48771072;
  void *hd, *du, *dd;
  CK(hipHostMalloc(&hd, bytes, hipHostMallocPortable));
  CK(hipMalloc(&du, bytes)); CK(hipMalloc(&dd, bytes));
  char* pageable = static_cast<char*>(std::malloc(bytes));
  std::memset(pageable, 0x77, bytes); std::memset(hd, 0, bytes);
  CK(hipMemset(dd, 0x33, bytes));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  auto down = [&]() { for (int k = 0; k < 9; ++k) CK(hipMemcpyAsync((char*)hd + k * piece, (char*)dd + k * piece, piece, hipMemcpyDeviceToHost, b)); };
  for (size_t mb : {4, 8, 16, 32, 48}) {
    const size_t X = mb << 20;
    char* stage[2]; hipEvent_t free_ev[2];
    for (int i = 0; i < 2; ++i) { CK(hipHostMalloc((void**)&stage[i], X, hipHostMallocPortable)); std::memset(stage[i], 0, X); CK(hipEventCreateWithFlags(&free_ev[i], hipEventDisableTiming)); }
    for (int threads : {4, 8}) {
      for (int with_down = 0; with_down < 2; ++with_down) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipDeviceSynchronize());
          const double t0 = now_ms();
          if (with_down) down();
          int i = 0; bool used[2] = {false, false};
          for (size_t o = 0; o < bytes; o += X, i ^= 1) {
            const size_t n = std::min(X, bytes - o);
            if (used[i]) CK(hipEventSynchronize(free_ev[i]));
            pack(stage[i], pageable + o, n, threads);
            CK(hipMemcpyAsync((char*)du + o, stage[i], n, hipMemcpyHostToDevice, a));
            CK(hipEventRecord(free_ev[i], a)); used[i] = true;
          }
          CK(hipDeviceSynchronize());
          best = std::min(best, now_ms() - t0);
        }
        std::printf("pieces of %2zu MiB, %d packing threads, %s: %6.2f ms  (%.1f GB/s %s)\n", mb, threads, with_down ? "with the downloads" : "upload alone      ", best, bytes / best / 1e6, with_down ? "each way" : "");
      }
    }
    for (int i = 0; i < 2; ++i) { CK(hipHostFree(stage[i])); CK(hipEventDestroy(free_ev[i])); }
  }
  return 0;
}
