// How often does a JPEG decoder that starts a 1024-bit subsequence in the wrong state leave it in the right one?  (CPU only.)
// The GPU entropy decoder (ist_jpeg_gpu.hip) synchronises by letting every subsequence hand its exit state to the next until
// nothing changes; the number of passes is the longest chain of subsequences that hand a WRONG state on.  This program replays
// the decoder's state evolution (bit position, MCU slot, zig-zag index - the same rules as symbol() there) on a real scan and
// counts, per subsequence: started from the true (position, index) but a wrong MCU slot, and started from the first pass's
// guess (a block starts at the boundary, slot 0) - does the exit state come out true, true except for the slot, or elsewhere?
//   g++ -std=c++17 -O2 -Iimagestitching_amd/csrc -Iinclude tools/sim_slot_sync.cpp imagestitching_amd/csrc/ist_jpeg.cpp -o /tmp/sim_slot_sync -lpthread
//   /tmp/sim_slot_sync photo.jpg            (a baseline JPEG, e.g. bench.photo_jpeg(0, 4032, 3024) written to a file)
// Nine-photo bench content (q90, 4:2:0): wrong slot -> exit true 0.62, slot wrong only 0.12, elsewhere 0.26; guess -> 0.69 / 0.10 /
// 0.21.  A link fails 38 % of the time, so among 130 000 subsequences the longest chain is ~10: the 9-10 passes the kernel trace shows.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ist_jpeg.h"
using namespace ist;
namespace ist { int fail(int code, const std::string& m) { std::fprintf(stderr, "error %d: %s\n", code, m.c_str()); return code; } }

static JpegGpuScan G;
static const uint8_t* s;
static uint32_t bits16(int64_t p) {
  uint64_t v = 0;
  for (int i = 0; i < 5; ++i) v = (v << 8) | (static_cast<size_t>(p / 8 + i) < G.stream.size() ? s[p / 8 + i] : 0);
  return static_cast<uint32_t>((v >> (24 - p % 8)) & 0xFFFF);
}
static int huff(const JpegGpuTables& T, bool isdc, uint32_t tab, uint32_t v16, uint32_t* len) {
  const uint16_t* look = reinterpret_cast<const uint16_t*>(&T);
  const uint32_t at = isdc ? (tab << kJpegDcLookBits) + (v16 >> (16 - kJpegDcLookBits))
                           : (2u << kJpegDcLookBits) + (tab << kJpegAcLookBits) + (v16 >> (16 - kJpegAcLookBits));
  const uint32_t e = look[at];
  if (e) { *len = e >> 8; return static_cast<int>(e & 0xFF); }
  const JpegHuffTail& h = T.tail[isdc ? tab : 2u + tab];
  if (v16 >= h.lim[7]) { *len = 1; return -1; }
  uint32_t k = 0;
  for (int q = 1; q <= 6; ++q) k += v16 >= h.lim[q];
  *len = 10 + k;
  return h.vals[(h.vptr[k] + ((v16 - h.lim[k]) >> (6 - k))) & 255];
}
struct St { int64_t p; int c, z; };
static St run(St S, int64_t limit) {
  while (S.p < limit) {
    const bool isdc = S.z == 0;
    const int comp = G.slot_comp[S.c];
    uint32_t len;
    const int rs = huff(G.tables, isdc, isdc ? G.dc_tab[comp] : G.ac_tab[comp], bits16(S.p), &len);
    S.p += len;
    if (rs < 0 && !isdc) continue;
    uint32_t r, sz;
    if (isdc) { r = 0; sz = (rs > 0 && rs <= 15) ? rs : 0; } else { r = rs >> 4; sz = rs & 15; }
    if (!isdc && sz == 0) { if (r == 15) S.z += 16; else S.z = 64; }
    else { S.z += r + 1; S.p += sz; }
    if (S.z >= 64) { S.z = 0; S.c = (S.c + 1) % G.slots; }
  }
  return S;
}
int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s file.jpg [subsequence bits]\n", argv[0]); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> d(64u << 20);
  const size_t n = std::fread(d.data(), 1, d.size(), f);
  std::fclose(f);
  JpegImage J;
  if (jpeg_parse_and_entropy_decode(d.data(), static_cast<int64_t>(n), &J, false, &G) || !G.eligible || !G.iv.empty()) { std::fprintf(stderr, "not a file the GPU decoder takes as one stream\n"); return 1; }
  s = G.stream.data();
  const int64_t SUB = argc > 2 ? std::atoll(argv[2]) : 1024;
  const int nsub = static_cast<int>((G.bits + SUB - 1) / SUB);
  std::vector<St> truth(static_cast<size_t>(nsub) + 1);
  St S{0, 0, 0};
  truth[0] = S;
  for (int i = 0; i < nsub; ++i) { S = run(S, std::min<int64_t>((i + 1) * SUB, G.bits)); truth[static_cast<size_t>(i) + 1] = S; }
  long same = 0, pz_only = 0, diff = 0, g_ok = 0, g_pz = 0, g_bad = 0;
  for (int i = 1; i + 1 < nsub; i += 3) {
    const int64_t lim = (i + 1) * SUB;
    const St& T1 = truth[static_cast<size_t>(i) + 1];
    for (int dl = 1; dl < G.slots; ++dl) {
      St a = truth[static_cast<size_t>(i)];
      a.c = (a.c + dl) % G.slots;
      const St e = run(a, lim);
      if (e.p == T1.p && e.z == T1.z) { if (e.c == T1.c) ++same; else ++pz_only; } else ++diff;
    }
    const St e = run(St{static_cast<int64_t>(i) * SUB, 0, 0}, lim);
    if (e.p == T1.p && e.z == T1.z) { if (e.c == T1.c) ++g_ok; else ++g_pz; } else ++g_bad;
  }
  const double t = static_cast<double>(same + pz_only + diff), tg = static_cast<double>(g_ok + g_pz + g_bad);
  std::printf("%d subsequences of %lld bits, %d blocks per MCU\n", nsub, static_cast<long long>(SUB), G.slots);
  std::printf("start = true (position, index), wrong slot:   exit true %.3f | slot wrong only %.3f | elsewhere %.3f\n", same / t, pz_only / t, diff / t);
  std::printf("start = guess (block at the boundary, slot 0): exit true %.3f | slot wrong only %.3f | elsewhere %.3f\n", g_ok / tg, g_pz / tg, g_bad / tg);
  return 0;
}
