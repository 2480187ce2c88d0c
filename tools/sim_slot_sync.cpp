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
  // ---- how many passes does a workgroup of the synchronisation kernel need?  (round 4)  Replays the kernel's iteration - 128 owned
  // subsequences + up to 8 ghost lanes in front, pass 0 from guesses, pass k from the left neighbour's exit of pass k-1, a lane
  // re-decodes only when its start state changed - for (A) the shipped scheme, one start hypothesis per lane, and (B) TWO: every lane
  // keeps the exits of two different starts (pass 0: the guess with slot 0 and with the first chroma slot; pass k: both exits of
  // its left neighbour) and the true chain is picked at the end.  A launch costs as much as its slowest workgroup.
  {
    const int WG = 128, GH = 8;
    auto eq = [](const St& a, const St& b) { return a.p == b.p && a.c == b.c && a.z == b.z; };
    int chroma_slot = 0;
    for (int k = 0; k < G.slots; ++k) if (G.slot_comp[k] != G.slot_comp[0]) { chroma_slot = k; break; }
    std::vector<int> hist_a(64, 0), hist_b(64, 0);
    int worst_a = 0, worst_b = 0;
    for (int g0 = 0; g0 < nsub; g0 += WG) {
      const int first = std::max(0, g0 - GH), last = std::min(nsub, g0 + WG);        // lanes [first, last): ghosts, then owned
      const int L = last - first;
      // (A)
      {
        std::vector<St> start(static_cast<size_t>(L)), ex(static_cast<size_t>(L));
        std::vector<char> have(static_cast<size_t>(L), 0);
        int passes = 0;
        for (int it = 0; it < 64; ++it) {
          std::vector<St> nex = ex;
          bool any = false;
          for (int e = 0; e < L; ++e) {
            const int i = first + e;
            St sp;
            if (i == 0) sp = St{0, 0, 0};
            else if (it == 0 || e == 0) sp = St{static_cast<int64_t>(i) * SUB, 0, 0};
            else sp = ex[static_cast<size_t>(e) - 1];
            if (have[static_cast<size_t>(e)] && eq(start[static_cast<size_t>(e)], sp)) continue;
            nex[static_cast<size_t>(e)] = run(sp, std::min<int64_t>((i + 1) * SUB, G.bits));
            start[static_cast<size_t>(e)] = sp; have[static_cast<size_t>(e)] = 1; any = true;
          }
          ex = nex;
          if (!any) break;
          ++passes;
        }
        hist_a[static_cast<size_t>(std::min(passes, 63))]++; worst_a = std::max(worst_a, passes);
      }
      // (B)
      {
        struct Two { St s[2], x[2]; int n = 0; };
        std::vector<Two> cur(static_cast<size_t>(L));
        int passes = 0;
        for (int it = 0; it < 64; ++it) {
          std::vector<Two> nxt = cur;
          bool any = false;
          for (int e = 0; e < L; ++e) {
            const int i = first + e;
            St cand[2]; int nc = 0;
            if (i == 0) { cand[nc++] = St{0, 0, 0}; }
            else if (it == 0 || e == 0) { cand[nc++] = St{static_cast<int64_t>(i) * SUB, 0, 0}; cand[nc++] = St{static_cast<int64_t>(i) * SUB, chroma_slot, 0}; }
            else { const Two& l = cur[static_cast<size_t>(e) - 1]; for (int k = 0; k < l.n; ++k) { bool dup = false; for (int q = 0; q < nc; ++q) dup = dup || eq(cand[q], l.x[k]); if (!dup) cand[nc++] = l.x[k]; } }
            Two t; t.n = nc;
            for (int k = 0; k < nc; ++k) {
              t.s[k] = cand[k];
              bool known = false;
              for (int q = 0; q < cur[static_cast<size_t>(e)].n; ++q) if (eq(cur[static_cast<size_t>(e)].s[q], cand[k])) { t.x[k] = cur[static_cast<size_t>(e)].x[q]; known = true; }
              if (!known) { t.x[k] = run(cand[k], std::min<int64_t>((i + 1) * SUB, G.bits)); any = true; }
            }
            nxt[static_cast<size_t>(e)] = t;
          }
          cur = nxt;
          if (!any) break;
          ++passes;
        }
        // the chain from the workgroup's first lane: is every owned lane's true exit among its candidates?  (it must be, once settled)
        hist_b[static_cast<size_t>(std::min(passes, 63))]++; worst_b = std::max(worst_b, passes);
      }
    }
    std::printf("passes a workgroup needs (first launch; %d subsequences + %d ghost lanes per workgroup): one hypothesis: worst %d, two hypotheses: worst %d\n", WG, GH, worst_a, worst_b);
    std::printf("  histogram, one hypothesis :");
    for (int k = 0; k < 64; ++k) if (hist_a[static_cast<size_t>(k)]) std::printf(" %d:%d", k, hist_a[static_cast<size_t>(k)]);
    std::printf("\n  histogram, two hypotheses:");
    for (int k = 0; k < 64; ++k) if (hist_b[static_cast<size_t>(k)]) std::printf(" %d:%d", k, hist_b[static_cast<size_t>(k)]);
    std::printf("\n");
  }
  // ---- loop trips of a counting pass if ONE look-up resolved up to two AC symbols (an index of W bits holding both codes and their magnitude
  // bits; never across a block end; the first symbol of a block - DC - always alone): trips per symbol along the true path of the scan
  {
    struct Sym { int bits; bool dc, ends; };
    std::vector<Sym> syms;
    St S{0, 0, 0};
    while (S.p < G.bits) {
      const St b = S;
      S = run(St{b.p, b.c, b.z}, b.p + 1);                          // run() stops once p has moved: exactly one symbol
      syms.push_back(Sym{static_cast<int>(S.p - b.p), b.z == 0, S.z == 0});
    }
    for (int W = 9; W <= 13; ++W) {
      long trips = 0;
      for (size_t k = 0; k < syms.size();) {
        ++trips;
        if (!syms[k].dc && !syms[k].ends && k + 1 < syms.size() && syms[k].bits + syms[k + 1].bits <= W) k += 2; else k += 1;
      }
      std::printf("two AC symbols per look-up, %2d-bit index: %.3f loop trips per symbol (%zu symbols, %.2f bits each)\n", W, static_cast<double>(trips) / syms.size(), syms.size(),
                  static_cast<double>(G.bits) / syms.size());
    }
  }
  const double t = static_cast<double>(same + pz_only + diff), tg = static_cast<double>(g_ok + g_pz + g_bad);
  std::printf("%d subsequences of %lld bits, %d blocks per MCU\n", nsub, static_cast<long long>(SUB), G.slots);
  std::printf("start = true (position, index), wrong slot:   exit true %.3f | slot wrong only %.3f | elsewhere %.3f\n", same / t, pz_only / t, diff / t);
  std::printf("start = guess (block at the boundary, slot 0): exit true %.3f | slot wrong only %.3f | elsewhere %.3f\n", g_ok / tg, g_pz / tg, g_bad / tg);
  return 0;
}
