// ist_jpeg_gpu.hip — JPEG entropy (Huffman) decoding on the GPU for baseline files: the serial half of the decode step
// (reference anchor: loadImageFrom, utils/canvas.js:27-121 — the platform decoder behind
// Image.src) made parallel.
//
// A Huffman-coded scan has no random access, but it is SELF-SYNCHRONISING: a decoder started at a wrong bit position
// falls into step with the true code boundaries after a few symbols.  The published scheme for GPUs (Weissenberger &
// Schmidt, "Massively Parallel Huffman Decoding on GPUs", ICPP 2018, and its JPEG follow-up) is restated here for this
// path:
//   * the de-stuffed scan is cut into subsequences of 1024 bits; one thread owns one subsequence (a scan with restart
//     intervals: every interval is cut on its own - it is an independent stream, T.81 E.1.4);
//   * pass 0: every thread decodes its subsequence from its first bit, pretending a block starts there, and records
//     the state in which it crosses its end (bit position, MCU slot, zig-zag index) and how many blocks it finished;
//   * pass t >= 1: every thread decodes its subsequence again, starting from the exit state its LEFT neighbour
//     recorded in pass t-1.  Subsequence 0 starts from the true state, so after pass t subsequences 0..t are exact —
//     and because of self-synchronisation almost all others are too.  When a pass changes no exit state the states are
//     a fixed point, hence (by induction from subsequence 0) all exact; a subsequence whose start state did not change
//     since it was last decoded keeps its result, so the later passes touch only the few that are still settling;
//   * the passes of one workgroup (256 neighbouring subsequences) run INSIDE one launch, exit states exchanged through
//     LDS; only the hand-over between workgroups needs another launch (ist_jpeg_sync_kernel);
//   * an exclusive scan of the per-subsequence block counts gives every thread the index of its first block; a last
//     pass decodes once more and writes the coefficients straight into the dense planes in HBM; the DC differences
//     are integrated per component in decoding order by a block-wide scan.
// Nothing of the coefficients crosses PCIe: the upload is the compressed scan (1.8 MB for a 12 MP photo instead of a
// 36 MB plane).  The result is validated (exactly the blocks the frame header promises; no invalid code, run or DC
// category on the true path); an image that fails is reported back and decoded by the host decoder (ist_jpeg.cpp),
// which also produces the error message.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ist_host.h"
#include "ist_internal.h"
#include "ist_jpeg.h"

namespace ist {

namespace {

// bits per subsequence (compile-time: it sizes the LDS staging area).  Whole call on nine 12 MP photos, round 2 kernels:
// 1024 bits 4.5 ms, 2048 bits 4.3 ms, 4096 bits 5.4 ms.  With the bitstream staged in LDS the decode is no longer paced
// by global-memory latency, and the shorter subsequence gives every SIMD two waves instead of one.
#ifndef IST_SUB_BITS          // (compile-time experiment switches: -DIST_SUB_BITS=512, -DIST_GHOSTS=2, -DIST_AC_LOOK_BITS=11 rebuild the variants DESIGN.md quotes)
#define IST_SUB_BITS 1024
#endif
constexpr int kSubBits = IST_SUB_BITS;
#ifndef IST_SYNC_THREADS
#define IST_SYNC_THREADS 128
#endif
// subsequences per workgroup of the synchronisation kernel (a multiple of kWriteThreads); a unit - an image, or ONE RESTART
// INTERVAL of it - is padded to whole workgroups.  (measured: 128 instead of 256 costs scans without restart intervals nothing,
// sync launches 0.95-0.96 vs 0.96-0.98 ms, and halves what short intervals waste: nine 12 MP photos with an interval per MCU row
// 4.17 -> 2.72 ms, per 32 MCUs 13.7 -> 6.4 ms)
constexpr int kSyncThreads = IST_SYNC_THREADS;
constexpr int kWriteThreads = 128;      // ... of the writing kernel (it also holds one 8x8 block per thread in LDS)
constexpr int kMarginWords = 64;        // staged behind a workgroup's own bits: a block can run 64 x 27 bits past its start
constexpr int kMaxPasses = 64;

__constant__ uint8_t kZig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct DevImg {
  const uint8_t* stream; int64_t bits;
  const JpegGpuTables* tables;            // the scan's two DC + two AC tables
  int32_t first_sub, n_sub;
  int32_t slots, total_blocks, mcus_x;
  int32_t mcu0;                           // the MCU its first block belongs to (non-zero: a later restart interval of a scan)
  uint8_t slot_comp[10], slot_idx[10], dc_tab[3], ac_tab[3];
  int16_t* coef[3];
  int32_t h[3], v[3], blocks_x[3];
  uint32_t* err;                          // set when the TRUE path meets what the host decoder would call corrupt data
};

struct State { uint32_t p; uint32_t c; uint32_t z; };

// (pointers read back from the LDS copy of the image record are generic to the compiler; the casts to address space 1
// make their accesses global_load / global_store instead of flat ones)
typedef const __attribute__((address_space(1))) uint8_t* GlobalBytes;
typedef const __attribute__((address_space(1))) uint32_t* GlobalWords;
typedef __attribute__((address_space(1))) int16_t* GlobalI16;
typedef __attribute__((address_space(1))) uint32_t* GlobalU32;

// What a workgroup keeps in LDS: the scan's Huffman tables and the record of its image (a workgroup never spans two
// images: the host pads every image's subsequences to whole workgroups) and ITS PART OF THE BITSTREAM, byte-swapped
// into big-endian words, plus a margin.  The decoding loop then touches global memory only to store finished blocks.
// (Round 2's first version kept all of this in global memory.  What paced it was not the dependent look-ups but the
// lock step: the 64 lanes of a wave refill their bit windows at different symbols, so nearly EVERY step of the wave
// waited for somebody's global load — 1.8 us per symbol step with one wave per SIMD, measured.)
template <int THREADS>
struct WgShared {
  JpegGpuTables tab;
  uint32_t bits[THREADS * (kSubBits / 32) + kMarginWords];
  DevImg img;
  uint32_t slot_tabs[12];               // per MCU slot: (component << 16) | (AC table << 8) | DC table
  uint8_t zig[64];
};

// (THREADS = the subsequences whose bits are staged; the loops stride by the real workgroup size)
template <int THREADS>
__device__ __forceinline__ void load_record(WgShared<THREADS>* sh, const DevImg* img) {
  static_assert(sizeof(DevImg) % 4 == 0 && sizeof(JpegGpuTables) % 4 == 0, "word copies");
  const uint32_t* gi = reinterpret_cast<const uint32_t*>(img);
  uint32_t* li = reinterpret_cast<uint32_t*>(&sh->img);
  for (uint32_t k = threadIdx.x; k < sizeof(DevImg) / 4; k += blockDim.x) li[k] = gi[k];
  if (threadIdx.x < 64) sh->zig[threadIdx.x] = kZig[threadIdx.x];
  if (threadIdx.x < 10) {
    const uint32_t comp = img->slot_comp[threadIdx.x] < 3 ? img->slot_comp[threadIdx.x] : 0u;
    sh->slot_tabs[threadIdx.x] = (img->dc_tab[comp] & 1u) | (static_cast<uint32_t>(img->ac_tab[comp] & 1u) << 8) | (comp << 16);
  }
  __syncthreads();
}
template <int THREADS>
__device__ __forceinline__ void load_stream(WgShared<THREADS>* sh, const DevImg* img, uint32_t first_bit) {
  const uint32_t* gt = reinterpret_cast<const uint32_t*>(img->tables);
  uint32_t* lt = reinterpret_cast<uint32_t*>(&sh->tab);
  for (uint32_t k = threadIdx.x; k < sizeof(JpegGpuTables) / 4; k += blockDim.x) lt[k] = gt[k];
  // the stream is 256-byte aligned on the device and carries 16 bytes of zero padding behind its last bit; first_bit
  // is a multiple of kSubBits, so these are aligned word loads.  Words past the padding read as zero.
  const uint32_t words_in_stream = static_cast<uint32_t>((img->bits >> 3) + 16) >> 2;
  const GlobalWords gw = (GlobalWords)(img->stream) + (first_bit >> 5);
  const uint32_t w0 = first_bit >> 5;
  for (uint32_t k = threadIdx.x; k < THREADS * (kSubBits / 32) + kMarginWords; k += blockDim.x)
    sh->bits[k] = (w0 + k < words_in_stream) ? __builtin_bswap32(gw[k]) : 0u;
  __syncthreads();
}

// at least the next 33 bits of the stream at bit position p, left-aligned in 64: enough for one code (<= 16 bits) and
// its magnitude bits (<= 15), from ONE two-word LDS read.  Everything a thread can reach lies inside the staged words: a block is at most 27 + 63 * 26 = 1665 bits
// long and the margin behind the workgroup's own bits is 2048; the index is clamped all the same.
template <int THREADS>
__device__ __forceinline__ uint64_t window(const WgShared<THREADS>* sh, uint32_t first_bit, uint32_t p) {
  constexpr uint32_t kWords = THREADS * (kSubBits / 32) + kMarginWords;
  const uint32_t q = p - first_bit;
  const uint32_t i = min(q >> 5, kWords - 2u);
  return ((static_cast<uint64_t>(sh->bits[i]) << 32) | sh->bits[i + 1]) << (q & 31u);
}

// symbol + its length; a bit pattern that is no code (only ever met on a speculative, out-of-phase path, or in a corrupt
// file) consumes one bit.  tab = which of the scan's two tables of its class.  Codes behind the look-ahead (9 bits DC,
// 11 bits AC: ist_jpeg.h) take three dependent LDS reads instead of a loop over the lengths.
__device__ __forceinline__ int huff(const JpegGpuTables& T, bool isdc, uint32_t tab, uint32_t v16, uint32_t* len) {
  const uint16_t* look = reinterpret_cast<const uint16_t*>(&T);      // (look_dc and look_ac are adjacent: one base, two index rules)
  static_assert(offsetof(JpegGpuTables, look_dc) == 0 && offsetof(JpegGpuTables, look_ac) == sizeof(uint16_t) * 2 * (1 << kJpegDcLookBits), "flat look-ahead index");
  const uint32_t at = isdc ? (tab << kJpegDcLookBits) + (v16 >> (16 - kJpegDcLookBits))
                           : (2u << kJpegDcLookBits) + (tab << kJpegAcLookBits) + (v16 >> (16 - kJpegAcLookBits));
  const uint32_t e = look[at];
  if (e) { *len = e >> 8; return static_cast<int>(e & 0xFF); }
  const JpegHuffTail& h = T.tail[isdc ? tab : 2u + tab];
  const uint4 a = *reinterpret_cast<const uint4*>(&h.lim[0]), b = *reinterpret_cast<const uint4*>(&h.lim[4]);
  if (v16 >= b.w) { *len = 1; return -1; }
  const uint32_t k = (v16 >= a.y) + (v16 >= a.z) + (v16 >= a.w) + (v16 >= b.x) + (v16 >= b.y) + (v16 >= b.z);
  *len = 10 + k;
  return h.vals[(h.vptr[k] + ((v16 - h.lim[k]) >> (6 - k))) & 255];
}

__device__ __forceinline__ int extend_bits(uint32_t v, int s) { return (v < (1u << (s - 1))) ? static_cast<int>(v) - (1 << s) + 1 : static_cast<int>(v); }

// One symbol of the scan from state S (DC and AC share the code: only the table differs, so the two kinds of lanes of
// a wave do not serialise).  *at = the zig-zag position a coded value belongs to (0 = the DC difference), *val = the
// value (AC values only when AC_VALUES); *stored = a value was coded (otherwise EOB / ZRL / a bit pattern that is no
// code, which consumes one bit and leaves the state in the block); *bad = the host decoder would call this corrupt.
template <bool AC_VALUES, int THREADS>
__device__ __forceinline__ void symbol(const WgShared<THREADS>* sh, uint32_t first_bit, State& S, uint32_t tabs, uint32_t* at, int* val, bool* stored, bool* bad) {
  const bool isdc = S.z == 0;
  uint32_t len;
  const uint64_t w = window(sh, first_bit, S.p);
  const int rs = huff(sh->tab, isdc, (isdc ? tabs : (tabs >> 8)) & 1u, static_cast<uint32_t>(w >> 48), &len);
  S.p += len;
  *stored = false; *bad = false;
  if (rs < 0 && !isdc) { *bad = true; return; }
  uint32_t r, sz;
  if (isdc) { *bad = rs < 0 || rs > 11; r = 0; sz = (rs > 0 && rs <= 15) ? static_cast<uint32_t>(rs) : 0u; }
  else { r = static_cast<uint32_t>(rs) >> 4; sz = static_cast<uint32_t>(rs) & 15u; }
  if (!isdc && sz == 0) {
    if (r == 15) S.z += 16; else S.z = 64;                  // ZRL / EOB
    return;
  }
  S.z += r;
  *at = S.z;
  *stored = true;
  *val = (sz && (AC_VALUES || isdc)) ? extend_bits(static_cast<uint32_t>((w << len) >> (64 - sz)), static_cast<int>(sz)) : 0;
  S.p += sz;
  S.z += 1;
}

// what a subsequence contributes to the ones behind it: blocks finished, and the sum of the DC differences coded in it,
// per component (a block's DC difference counts where the block STARTS — the same rule as the block's ownership in the
// writing pass)
struct Tally { uint32_t blocks; int32_t dc[3]; };

// Synchronisation passes: decodes from state S until the bit position reaches `limit`, in two legs: up to bit `mid` of
// the subsequence's range, then the rest.  The state between the legs is a CHECKPOINT.  A re-decode from a new start
// state almost always falls into step with the previous decode of the same subsequence within a few symbols; when it
// arrives at the checkpoint in the recorded state, the second leg would repeat the old path, so it is skipped: the exit
// state is the old one and the tally is the new first leg + the old second leg.  A workgroup's pass costs as much as
// its slowest lane: in the later passes, where only a few lanes re-decode, this cuts a pass from a whole subsequence to
// its first quarter.  (As a test inside the symbol loop — two checkpoints, checked at every step — the same idea cost
// every step of every pass and LOST 25 %; as a loop boundary it is free.)
constexpr uint32_t kCheckBits = 256;
struct Check { uint32_t p, cz; Tally t; };

// THE COUNTING STEP'S OWN TABLE (round 4).  The synchronisation passes need of a symbol only what it does to the state: how many
// bits it takes (code + magnitude bits) and where it leaves the zig-zag index.  For every look-ahead pattern that holds a whole code
// (9 bits: ~98 % of a photo's symbols) that is one 16-bit entry, made from the scan's look-ahead tables when a workgroup stages them:
//   bits 0-4   code length + magnitude bits         bits 5-9   what an AC symbol adds to the zig-zag index (run + 1; 16 for ZRL)
//   bit 10     end of block                          bits 11-14 the code's own length (a DC symbol's value starts behind it)
// 0 = the pattern is longer than the look-ahead, or no code: the general step (symbol<>) takes it.  The state a symbol leaves is
// the same either way - it must be: the writing pass walks the stream with the general step from the states these passes agree on.
// A lone lane's chain per symbol falls from ~70 dependent instructions to ~40 (it is latency, not issue, that paces these passes).
constexpr uint32_t kFastEob = 0x400u;
// Two rewrites of the counting step, measured on one box against the step that ships (0 / 0) and NOT taken (LAB_NOTES 1.2, "what paces a pass"):
// IST_HUFF_STEP=1 - branch-free (selects instead of a dozen exec-mask regions; the slot's tables from a 64-bit scalar instead of LDS): launch
// 850-870 us against 778-805; IST_HUFF_PAIRS=1 (needs STEP=1) - two AC symbols per look-up where an 11-bit pattern holds both: loop trips of the
// busiest lane 214 -> 144, first pass 114 -> 104 us, launch 821-837 us.  Kept as compile-time switches with the instrumented build (IST_SYNC_DEBUG).
#ifndef IST_HUFF_STEP
#define IST_HUFF_STEP 0
#endif
#ifndef IST_HUFF_PAIRS
#define IST_HUFF_PAIRS 0
#endif
#if IST_HUFF_PAIRS && !IST_HUFF_STEP
#error "IST_HUFF_PAIRS needs IST_HUFF_STEP=1"
#endif
__device__ __forceinline__ uint16_t fast_entry(uint32_t e, bool isdc) {
  if (!e) return 0;
  const uint32_t L = e >> 8, rs = e & 255u;
  if (isdc) {
    const uint32_t sz = (rs > 0 && rs <= 15) ? rs : 0u;      // (symbol<>: the same rule)
    return static_cast<uint16_t>((L + sz) | (1u << 5) | (L << 11));
  }
  const uint32_t r = rs >> 4, sz = rs & 15u;
  if (sz == 0) return static_cast<uint16_t>(L | (r == 15 ? (16u << 5) : kFastEob) | (L << 11));
  return static_cast<uint16_t>((L + sz) | ((r + 1) << 5) | (L << 11));
}
static_assert(kJpegDcLookBits == 9 && kJpegAcLookBits == 9, "fast table: 512 entries per table, code + magnitude bits <= 24 < 32");

// TWO AC SYMBOLS PER LOOK-UP (round 4).  A counting pass is a chain of dependent steps - ~1250 cycles per loop trip, 214 trips for the busiest lane
// of a workgroup's first pass (instrumented build, LAB_NOTES 1.2) - so what shortens it is fewer trips.  A photo's symbols average 5.4 bits: an
// 11-bit look-ahead pattern often holds two whole AC symbols (codes and magnitude bits).  pair[] has one 16-bit entry per AC table and pattern:
//   bit 15 = 1   bits 0-3 = the bits both symbols take   bits 4-9 = what both add to the zig-zag index (63: the second is the end of block)
//   bits 10-14 = what the FIRST adds (the pair is used only while that keeps the index inside the block)          0 = no pair in this pattern
// The first symbol is never an end of block, and a pair is taken only when it ENDS no later than the leg's limit, so the state a leg leaves
// is bit for bit the one the single steps leave (tools/sim_slot_sync.cpp: 0.69 loop trips per symbol on the bench's photos).
#if IST_HUFF_PAIRS
constexpr int kPairBits = 11;
__device__ __forceinline__ uint16_t pair_entry(const uint16_t* look_ac_tab, uint32_t pat) {
  const uint32_t e1 = look_ac_tab[pat >> (kPairBits - kJpegAcLookBits)];
  if (!e1) return 0;
  const uint32_t L1 = e1 >> 8, r1 = (e1 >> 4) & 15u, s1 = e1 & 15u;
  if (s1 == 0 && r1 != 15) return 0;                                  // end of block first: nothing to pair
  const uint32_t tot1 = L1 + s1, adv1 = s1 == 0 ? 16u : r1 + 1u;
  if (tot1 + 2 > static_cast<uint32_t>(kPairBits)) return 0;          // (a code has at least two bits)
  const uint32_t R = kPairBits - tot1;
  const uint32_t pat2 = (pat << tot1) & ((1u << kPairBits) - 1u);      // the rest of the pattern, left-aligned, zero-filled
  const uint32_t e2 = look_ac_tab[pat2 >> (kPairBits - kJpegAcLookBits)];
  if (!e2) return 0;
  const uint32_t L2 = e2 >> 8, r2 = (e2 >> 4) & 15u, s2 = e2 & 15u;
  if (L2 + s2 > R) return 0;                                          // the second symbol's code or magnitude bits reach past the pattern
  const uint32_t advt = s2 == 0 ? (r2 == 15 ? adv1 + 16u : 63u) : adv1 + r2 + 1u;
  return static_cast<uint16_t>(0x8000u | (tot1 + L2 + s2) | (advt << 4) | (adv1 << 10));
}
#endif

template <int THREADS>
__device__ __forceinline__ Tally run_count(const WgShared<THREADS>* sh, const uint16_t* fast, const uint16_t* pair, uint32_t first_bit, State& S, uint32_t limit, uint32_t mid,
                                           bool have_ref, uint32_t old_p, uint32_t old_cz, const Tally& old_t, Check& ck
#ifdef IST_SYNC_DEBUG
                                           , uint32_t& g_dbg_iters
#endif
                                           ) {
  Tally T; T.blocks = 0; T.dc[0] = T.dc[1] = T.dc[2] = 0;
  const uint32_t slots = static_cast<uint32_t>(sh->img.slots);
  // (measured and dropped, round 4: keeping the bits at S.p in a register between symbols - one two-word LDS read per ~4 symbols
  // instead of one per symbol - changed nothing: sync launch 817-894 us against 808-881 us; nor did a pad word per 32 words of the
  // staged stream, which takes the lanes of a wave - 32 words apart - off each other's LDS bank: 801-893 us, writing pass 234-238
  // against 229-237 us)
  // The step is BRANCH-FREE for every symbol the table holds (IST_HUFF_STEP=1, the default): DC and AC symbols, block ends and slot changes go
  // through the same ~50 instructions with selects; the one branch left is the miss (a code longer than 9 bits), which takes the general step.
  // The branchy form it replaces (IST_HUFF_STEP=0) spent a third of its ~75 instructions and a dozen exec-mask regions per symbol on
  // "is it a DC symbol", "did it store", "did the block end" - each a VALU -> SGPR -> exec round trip - and read the next slot's tables from LDS;
  // here the per-slot nibbles (DC table, AC table, component) of all ten slots sit in one 64-bit scalar.
#if IST_HUFF_STEP
  uint64_t packed = 0;
  for (uint32_t k = 0; k < 10; ++k) {
    const uint32_t t = sh->slot_tabs[k];
    packed |= static_cast<uint64_t>((t & 1u) | (((t >> 8) & 1u) << 1) | (((t >> 16) & 3u) << 2)) << (4 * k);
  }
  uint32_t nib = static_cast<uint32_t>(packed >> (4 * S.c)) & 15u;
#endif
  auto leg = [&](uint32_t until) {
#if IST_HUFF_STEP
    while (S.p < until) {
#ifdef IST_SYNC_DEBUG
      ++g_dbg_iters;
#endif
      const bool isdc = S.z == 0;
      const uint64_t w = window(sh, first_bit, S.p);
      const uint32_t f = fast[((isdc ? (nib & 1u) : 2u + ((nib >> 1) & 1u)) << 9) + static_cast<uint32_t>(w >> 55)];
#if IST_HUFF_PAIRS
      // (both look-ups leave together: the pair's is not on the chain.  A DC lane reads an entry it ignores.)
      const uint32_t e2 = pair[(((nib >> 1) & 1u) << kPairBits) + static_cast<uint32_t>(w >> (64 - kPairBits))];
      const bool two = !isdc && e2 != 0u && S.z + ((e2 >> 10) & 31u) < 64u && S.p + (e2 & 15u) <= until;
#else
      const bool two = false; const uint32_t e2 = 0;
#endif
      if (__builtin_expect(f == 0u && !two, 0)) {                 // longer than the look-ahead, or no code: the general step
#ifdef IST_SYNC_DEBUG
        g_dbg_iters += 0x10000u;                                   // (high half: this lane's misses)
#endif
        uint32_t at = 1; int val = 0; bool stored = false, bad;
        symbol<false>(sh, first_bit, S, sh->slot_tabs[S.c], &at, &val, &stored, &bad);
        if (stored && at == 0) { const uint32_t comp = nib >> 2; T.dc[0] += comp == 0 ? val : 0; T.dc[1] += comp == 1 ? val : 0; T.dc[2] += comp == 2 ? val : 0; }
        if (S.z >= 64) { S.z = 0; S.c = (S.c + 1 == slots) ? 0u : S.c + 1; ++T.blocks; nib = static_cast<uint32_t>(packed >> (4 * S.c)) & 15u; }
        continue;
      }
      const uint32_t tot1 = f & 31u, L = (f >> 11) & 15u, sz = tot1 - L;
      const uint32_t tot = two ? (e2 & 15u) : tot1;
      // the DC difference (sz magnitude bits behind the code; 0 bits -> 0), computed for every symbol and kept for DC symbols only
      const uint32_t raw = static_cast<uint32_t>(((w << L) >> 1) >> (63u - sz));
      const int ext = (raw < ((1u << sz) >> 1)) ? static_cast<int>(raw) - (1 << sz) + 1 : static_cast<int>(raw);
      const int val = isdc ? ext : 0;
      const uint32_t comp = nib >> 2;
      T.dc[0] += comp == 0 ? val : 0; T.dc[1] += comp == 1 ? val : 0; T.dc[2] += comp == 2 ? val : 0;
      S.p += tot;
      const uint32_t advt = (e2 >> 4) & 63u;
      const uint32_t z = two ? (advt == 63u ? 64u : S.z + advt) : ((f & kFastEob) ? 64u : S.z + ((f >> 5) & 31u));      // (a DC entry advances by 1)
      const bool fin = z >= 64u;
      const uint32_t c1 = (S.c + 1 == slots) ? 0u : S.c + 1;
      S.z = fin ? 0u : z;
      S.c = fin ? c1 : S.c;
      T.blocks += fin ? 1u : 0u;
      nib = static_cast<uint32_t>(packed >> (4 * S.c)) & 15u;
    }
#else
    uint32_t tabs = sh->slot_tabs[S.c];
    while (S.p < until) {
#ifdef IST_SYNC_DEBUG
      ++g_dbg_iters;
#endif
      const bool isdc = S.z == 0;
      const uint64_t w = window(sh, first_bit, S.p);
      const uint32_t f = fast[(((isdc ? 0u : 2u) + ((isdc ? tabs : (tabs >> 8)) & 1u)) << 9) + static_cast<uint32_t>(w >> 55)];
      uint32_t at = 1; int val = 0; bool stored = false, bad;
      if (f) {
        const uint32_t tot = f & 31u;
        if (isdc) {
          const uint32_t L = (f >> 11) & 15u, sz = tot - L;
          val = sz ? extend_bits(static_cast<uint32_t>((w << L) >> (64 - sz)), static_cast<int>(sz)) : 0;
          at = 0; stored = true;
          S.z = 1;
        } else S.z = (f & kFastEob) ? 64u : S.z + ((f >> 5) & 31u);
        S.p += tot;
      } else {
#ifdef IST_SYNC_DEBUG
        g_dbg_iters += 0x10000u;
#endif
        symbol<false>(sh, first_bit, S, tabs, &at, &val, &stored, &bad);
      }
      if (stored && at == 0) {
        const uint32_t comp = tabs >> 16;
        T.dc[0] += comp == 0 ? val : 0; T.dc[1] += comp == 1 ? val : 0; T.dc[2] += comp == 2 ? val : 0;
      }
      if (S.z >= 64) {                                          // block finished
        S.z = 0;
        S.c = (S.c + 1 == slots) ? 0u : S.c + 1;
        ++T.blocks;
        tabs = sh->slot_tabs[S.c];
      }
    }
#endif
  };
  leg(min(mid, limit));
  const uint32_t cz = (S.c << 8) | S.z;
  if (have_ref && ck.p == S.p && ck.cz == cz) {                 // in step with the previous decode from here on
    const Tally first = T;
    T.blocks += old_t.blocks - ck.t.blocks;
    for (int c = 0; c < 3; ++c) T.dc[c] += old_t.dc[c] - ck.t.dc[c];
    ck.t = first;
    S.p = old_p; S.c = old_cz >> 8; S.z = old_cz & 255u;
    return T;
  }
  ck.p = S.p; ck.cz = cz; ck.t = T;
  leg(limit);
  return T;
}

// Writing pass.  A block belongs to the subsequence its FIRST symbol starts in: the thread skips the tail of a block
// that began further left (decoding it only to stay in step), then decodes every block that starts before `limit` to
// its end — past `limit` if need be — into a 128-byte LDS slot and stores the slot as one full 128-byte block, DC
// coefficient included (pred = the component's DC predictor where the subsequence starts, from the scan of the
// tallies).  Every block of the image is therefore written exactly once, whole: the planes need no clearing pass,
// there is no separate DC pass, and HBM sees full lines instead of one 2-byte store per non-zero coefficient.
// `blk` = index of the block the start state is in.
template <int THREADS>
__device__ __forceinline__ void run_write(const WgShared<THREADS>* sh, uint32_t first_bit, uint32_t* slot, State S, uint32_t limit, uint32_t blk, int32_t pred0, int32_t pred1, int32_t pred2) {
  const DevImg& I = sh->img;
  GlobalU32 err = (GlobalU32)I.err;
  const uint32_t total = static_cast<uint32_t>(I.total_blocks), slots = static_cast<uint32_t>(I.slots);
  const uint32_t bits = static_cast<uint32_t>(I.bits);
  uint32_t tabs = sh->slot_tabs[S.c];
  auto next_block = [&]() {
    S.z = 0;
    S.c = (S.c + 1 == slots) ? 0u : S.c + 1;
    ++blk;
    tabs = sh->slot_tabs[S.c];
  };
  if (S.z != 0) {                                             // somebody else's block: only stay in step
    while (S.z < 64 && S.p < bits) {
      uint32_t at; int val; bool stored, bad;
      symbol<false>(sh, first_bit, S, tabs, &at, &val, &stored, &bad);
      if (bad) return;                                        // corrupt data on the true path: the block's owner reports it
    }
    if (S.z < 64) return;                                     // the scan ended inside it
    next_block();
  }
  while (S.p < limit) {
    const bool wr = blk < total;                              // (beyond: the padding bits behind the last block)
    const uint32_t comp = tabs >> 16, si = I.slot_idx[S.c];
    bool corrupt = false;
    do {
      uint32_t at = 0; int val = 0; bool stored, bad;
      symbol<true>(sh, first_bit, S, tabs, &at, &val, &stored, &bad);
      corrupt |= bad;
      if (bad && !stored) break;                              // no code at all: stop here (every step of this pass is on the
                                                              // true path, so the image is corrupt; a block is <= 64 steps otherwise)
      if (stored) {
        if (at == 0) {
          pred0 += comp == 0 ? val : 0; pred1 += comp == 1 ? val : 0; pred2 += comp == 2 ? val : 0;
          reinterpret_cast<int16_t*>(slot)[0] = static_cast<int16_t>(comp == 0 ? pred0 : comp == 1 ? pred1 : pred2);
        } else if (at >= 64) corrupt = true;                  // run past the block
        else reinterpret_cast<int16_t*>(slot)[sh->zig[at]] = static_cast<int16_t>(val);
      }
    } while (S.z < 64 && S.p < bits);
    if (S.z < 64) corrupt = true;                             // the scan ended inside the block
    if (wr) {
      if (corrupt) *err = 1u;
      const uint32_t h = static_cast<uint32_t>(I.h[comp]), v = static_cast<uint32_t>(I.v[comp]);
      const uint32_t mcu = static_cast<uint32_t>(I.mcu0) + blk / slots;
      const uint32_t mx = mcu % static_cast<uint32_t>(I.mcus_x), my = mcu / static_cast<uint32_t>(I.mcus_x);
      const uint32_t bx = mx * h + si % h, by = my * v + si / h;
      typedef uint32_t V4 __attribute__((ext_vector_type(4)));
      __attribute__((address_space(1))) V4* dst = (__attribute__((address_space(1))) V4*)((GlobalI16)I.coef[comp] + (static_cast<size_t>(by) * I.blocks_x[comp] + bx) * 64);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        V4 w;
        w.x = slot[4 * q]; w.y = slot[4 * q + 1]; w.z = slot[4 * q + 2]; w.w = slot[4 * q + 3];
        slot[4 * q] = 0u; slot[4 * q + 1] = 0u; slot[4 * q + 2] = 0u; slot[4 * q + 3] = 0u;
        dst[q] = w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 32; ++q) slot[q] = 0u;
    }
    if (S.z < 64) return;
    next_block();
  }
}

struct SyncArgs {
  const DevImg* imgs; const uint16_t* sub_img;
  const uint32_t* in_p; const uint32_t* in_cz;       // exit states of the previous pass
  uint32_t* out_p; uint32_t* out_cz;
  uint32_t* tally;                                     // per subsequence: blocks, dc[3]
  uint32_t* start_p; uint32_t* start_cz;               // the start state each subsequence was last decoded from
  uint32_t* checks;                                    // per subsequence: its checkpoint (p, cz, tally: 6 words)
  uint32_t* half_total;                                // per 128 subsequences: the sum of their tallies
  uint32_t* changed; int32_t pass;
};

// One LAUNCH = as many synchronisation passes as its workgroup needs: the 256 subsequences of a workgroup exchange exit
// states through LDS and iterate (re-decode whoever's start state changed, barrier) until none of them changes, so a
// chain of out-of-phase subsequences is chased to its end inside one launch instead of one subsequence per
// host-synchronised pass.  Across workgroups the first thread starts from the exit state its left neighbour reached in
// the PREVIOUS launch; a launch in which no workgroup's last exit state changed is the global fixed point.  (The
// "overflow" idea of the published scheme, restated for a barrier-synchronised workgroup.)
constexpr int kInnerPasses = 48;
// GHOST LANES.  In the first launch the first thread of a workgroup has no left neighbour to take its start state from
// (that neighbour belongs to the previous workgroup), so the whole workgroup used to settle on a guess, and the second
// launch — whose only news is the previous workgroup's true exit state — re-decoded thread 0 of nearly EVERY workgroup:
// a full subsequence time for the launch, since a pass costs as much as its slowest lane.  Now a fifth wave carries
// ghost lanes that decode the subsequences in front of the workgroup (from a guess, then from each other), and thread 0
// starts from the last ghost's exit.  How many: a decoder falls into step with the code boundaries within a few symbols,
// but with the MCU SLOT (which table set applies) only by chance: of the subsequences entered in a wrong state 38 % hand a
// wrong state on (tools/sim_slot_sync.cpp replays the rule on the CPU: 62 % exit true, 12 % with only the slot wrong, 26 % elsewhere).  With 2 ghosts some of a call's 531 workgroups therefore still started wrong and the second
// launch cost 0.24 ms; with 6 or more it finds nothing to redo and returns in 7 us (nine 12 MP photos, kernel trace:
// sync launches 0.89 + 0.24 ms -> 0.94-1.09 + 0.007 ms; both launches timed on the host 1.19-1.25 -> 0.99-1.00 ms).
#ifndef IST_GHOSTS
#define IST_GHOSTS 8
#endif
constexpr int kGhosts = IST_GHOSTS;
constexpr int kSyncBlock = kSyncThreads + 64;

#ifdef IST_SYNC_DEBUG          // measurement build only (tools/exp_huff.py with IST_LIB_PATH): per workgroup of the FIRST launch - passes, 10 ns ticks in all,
__device__ uint32_t g_sync_dbg[8 * 8192];   // in staging, in pass 0, and the longest single later pass
#endif
__global__ __launch_bounds__(kSyncBlock) void ist_jpeg_sync_kernel(const SyncArgs A) {
#ifdef IST_SYNC_DEBUG
  const uint64_t dbg_t0 = wall_clock64();
  uint64_t dbg_staged = dbg_t0, dbg_pass0 = dbg_t0, dbg_prev = dbg_t0, dbg_longest = 0;
  uint32_t dbg_passes = 0, dbg_redo_lanes = 0, dbg_iters = 0, dbg_iters0 = 0;
  uint64_t dbg_c0 = 0, dbg_c1 = 0;
  __shared__ uint32_t dbg_max_iters, dbg_miss;
  if (threadIdx.x == 0) { dbg_max_iters = 0; dbg_miss = 0; }
#endif
  __shared__ uint32_t ex_p[kSyncThreads + kGhosts], ex_cz[kSyncThreads + kGhosts];
  constexpr int kHalves = kSyncThreads / 128;                    // groups of kWriteThreads subsequences per workgroup
  static_assert(kSyncThreads % 128 == 0 && kHalves >= 1 && kHalves <= 2, "half totals");
  __shared__ uint32_t tot[kHalves][4];
  __shared__ WgShared<kSyncThreads + kGhosts> sh;
  __shared__ uint16_t fast[4 << 9];                              // the counting step's table: DC 0, DC 1, AC 0, AC 1
#if IST_HUFF_PAIRS
  __shared__ uint16_t pair[2 << kPairBits];                      // ... and its pairs: AC 0, AC 1
#else
  const uint16_t* pair = nullptr;
#endif
  const int tid = threadIdx.x;
  const bool owned = tid < kSyncThreads;
  const int g0 = blockIdx.x * kSyncThreads;                      // (the grid is exactly the padded subsequence count / kSyncThreads)
  const DevImg* gimg = &A.imgs[A.sub_img[g0 / kWriteThreads]];   // one image per workgroup
  const int i0 = g0 - gimg->first_sub;                           // the workgroup's first subsequence inside its image
  const int n_ghost = A.pass == 0 ? min(kGhosts, i0) : 0;
  const uint32_t first_bit = static_cast<uint32_t>(i0 - n_ghost) * static_cast<uint32_t>(kSubBits);
  if (tid < 4 * kHalves) tot[tid >> 2][tid & 3] = 0u;
  load_record(&sh, gimg);
  // e = position in the exchange arrays: ghosts first, then the owned subsequences; everybody's left neighbour is e - 1
  const int ghost_k = tid - kSyncThreads;                        // 0 .. for the ghost wave's first lanes
  const int e = owned ? tid + kGhosts : kGhosts - n_ghost + ghost_k;
  const int g = owned ? g0 + tid : g0 - n_ghost + ghost_k;
  const uint32_t i = static_cast<uint32_t>(g - sh.img.first_sub);
  const bool ghost = !owned && ghost_k < n_ghost;
  const bool live = owned ? i < static_cast<uint32_t>(sh.img.n_sub) : ghost;      // the padding subsequences of an image idle
  uint32_t limit = 0;
  if (live) {
    const uint64_t end = static_cast<uint64_t>(i + 1) * static_cast<uint64_t>(kSubBits);
    limit = static_cast<uint32_t>(end < static_cast<uint64_t>(sh.img.bits) ? end : static_cast<uint64_t>(sh.img.bits));
  }
  // what this subsequence last decoded from, and to (carried across launches in start_* / in_* / tally)
  bool have = live && A.pass > 0;
  uint32_t st_p = have ? A.start_p[g] : 0u, st_cz = have ? A.start_cz[g] : 0u;
  uint32_t my_p = have ? A.in_p[g] : 0u, my_cz = have ? A.in_cz[g] : 0u;
  Tally T; T.blocks = 0; T.dc[0] = T.dc[1] = T.dc[2] = 0;
  Check ck; ck.p = 0xFFFFFFFFu; ck.cz = 0; ck.t = T;
  if (have) {
    T.blocks = A.tally[4 * g]; T.dc[0] = static_cast<int32_t>(A.tally[4 * g + 1]); T.dc[1] = static_cast<int32_t>(A.tally[4 * g + 2]); T.dc[2] = static_cast<int32_t>(A.tally[4 * g + 3]);
    const uint32_t* q = A.checks + 6 * static_cast<size_t>(g);
    ck.p = q[0]; ck.cz = q[1]; ck.t.blocks = q[2]; ck.t.dc[0] = static_cast<int32_t>(q[3]); ck.t.dc[1] = static_cast<int32_t>(q[4]); ck.t.dc[2] = static_cast<int32_t>(q[5]);
  }
  // a later launch in which nobody's start state moved has nothing to decode: skip the staging too
  bool any = true;
  if (A.pass > 0) {
    const bool moved = live && i != 0 && !(st_p == A.in_p[g - 1] && st_cz == A.in_cz[g - 1]);
    any = __syncthreads_or(moved ? 1 : 0) != 0;
  }
  bool settled = !any;
  if (any) {
    load_stream(&sh, gimg, first_bit);
    for (int k = tid; k < (4 << 9); k += kSyncBlock) fast[k] = fast_entry(k < (2 << 9) ? sh.tab.look_dc[k >> 9][k & 511] : sh.tab.look_ac[(k >> 9) - 2][k & 511], k < (2 << 9));
#if IST_HUFF_PAIRS
    for (int k = tid; k < (2 << kPairBits); k += kSyncBlock) pair[k] = pair_entry(sh.tab.look_ac[k >> kPairBits], static_cast<uint32_t>(k) & ((1u << kPairBits) - 1u));
#endif
    __syncthreads();
#ifdef IST_SYNC_DEBUG
    dbg_staged = dbg_prev = wall_clock64();
    dbg_c0 = clock64();
#endif
    for (int it = 0; it < kInnerPasses; ++it) {
      uint32_t sp = 0, scz = 0;
      if (live && i != 0) {                            // (the first subsequence of an image starts from the true state 0, 0, 0)
        const bool has_left = e > kGhosts - n_ghost;   // a ghost or an owned lane to the left, inside this workgroup's exchange
        if (A.pass == 0) {
          if (it == 0 || !has_left) { sp = i * static_cast<uint32_t>(kSubBits); scz = 0; }           // a guess: a block starts here
          else { sp = ex_p[e - 1]; scz = ex_cz[e - 1]; }
        } else {
          if (it == 0 || tid == 0) { sp = A.in_p[g - 1]; scz = A.in_cz[g - 1]; }                       // the previous launch's exit states
          else { sp = ex_p[e - 1]; scz = ex_cz[e - 1]; }                                               // the left neighbour, this launch
        }
      }
      const bool redo = live && !(have && st_p == sp && st_cz == scz);
      if (redo) {
        State S; S.p = sp; S.c = scz >> 8; S.z = scz & 255u;
        T = run_count(&sh, fast, pair, first_bit, S, limit, i * static_cast<uint32_t>(kSubBits) + kCheckBits, have, my_p, my_cz, T, ck
#ifdef IST_SYNC_DEBUG
                      , dbg_iters
#endif
                      );
        my_p = S.p; my_cz = (S.c << 8) | S.z;
        st_p = sp; st_cz = scz; have = true;
      }
      __syncthreads();                                 // every thread has read its neighbour's previous exit state
      if (live) { ex_p[e] = my_p; ex_cz[e] = my_cz; }
#ifdef IST_SYNC_DEBUG
      {
        const uint64_t now = wall_clock64();
        if (it == 0) { dbg_pass0 = now; dbg_c1 = clock64(); atomicMax(&dbg_max_iters, dbg_iters & 0xFFFFu); atomicAdd(&dbg_miss, dbg_iters >> 16); } else dbg_longest = max(dbg_longest, now - dbg_prev);
        dbg_prev = now; ++dbg_passes;
        dbg_redo_lanes += __syncthreads_count(redo ? 1 : 0);
      }
#endif
      if (!__syncthreads_or(redo ? 1 : 0)) { settled = true; break; }
    }
#ifdef IST_SYNC_DEBUG
    if (A.pass == 0 && tid == 0 && blockIdx.x < 8192) {
      uint32_t* q = g_sync_dbg + 8 * blockIdx.x;
      q[0] = dbg_passes; q[1] = static_cast<uint32_t>(wall_clock64() - dbg_t0); q[2] = static_cast<uint32_t>(dbg_staged - dbg_t0);
      q[3] = static_cast<uint32_t>(dbg_pass0 - dbg_staged); q[4] = static_cast<uint32_t>(dbg_longest); q[5] = dbg_redo_lanes;
      q[6] = static_cast<uint32_t>(dbg_c1 - dbg_c0); q[7] = dbg_max_iters | 0x80000000u | (min(dbg_miss, 0x7FFFu) << 16);
    }
#endif
  }
  // the tallies of each half of the workgroup, for the writing pass's bases (integer adds: order does not matter)
  if (live && owned) {
    atomicAdd(&tot[tid >> 7][0], T.blocks); atomicAdd(&tot[tid >> 7][1], static_cast<uint32_t>(T.dc[0]));
    atomicAdd(&tot[tid >> 7][2], static_cast<uint32_t>(T.dc[1])); atomicAdd(&tot[tid >> 7][3], static_cast<uint32_t>(T.dc[2]));
  }
  __syncthreads();
  if (tid < 4 * kHalves) A.half_total[(kHalves * blockIdx.x + (tid >> 2)) * 4 + (tid & 3)] = tot[tid >> 2][tid & 3];
  if (!live || !owned) return;
  // the launch changed something the NEXT workgroup depends on (or ran out of inner passes): not the fixed point yet
  const bool last = tid == kSyncThreads - 1 || i + 1 == static_cast<uint32_t>(sh.img.n_sub);
  if (A.pass == 0 || !settled || (last && (A.in_p[g] != my_p || A.in_cz[g] != my_cz))) {
    if (A.pass == 0 ? (tid == 0) : true) *A.changed = 1u;
  }
  A.out_p[g] = my_p; A.out_cz[g] = my_cz;
  A.tally[4 * g] = T.blocks; A.tally[4 * g + 1] = static_cast<uint32_t>(T.dc[0]); A.tally[4 * g + 2] = static_cast<uint32_t>(T.dc[1]); A.tally[4 * g + 3] = static_cast<uint32_t>(T.dc[2]);
  A.start_p[g] = st_p; A.start_cz[g] = st_cz;
  uint32_t* q = A.checks + 6 * static_cast<size_t>(g);
  q[0] = ck.p; q[1] = ck.cz; q[2] = ck.t.blocks; q[3] = static_cast<uint32_t>(ck.t.dc[0]); q[4] = static_cast<uint32_t>(ck.t.dc[1]); q[5] = static_cast<uint32_t>(ck.t.dc[2]);
}

struct WriteArgs { const DevImg* imgs; const uint16_t* sub_img; const uint32_t* p; const uint32_t* cz; const uint32_t* tally; const uint32_t* half_total; };

// One workgroup = 128 subsequences.  Its base (blocks finished and DC sums in front of it, inside its image) is the sum
// of the half_total entries of the image in front of it; an exclusive scan of its own tallies gives every thread its own.
__global__ __launch_bounds__(kWriteThreads) void ist_jpeg_write_kernel(const WriteArgs A) {
  __shared__ WgShared<kWriteThreads> sh;
  __shared__ uint32_t slots[kWriteThreads * 33];                 // one 8x8 block per thread, 33-word pitch (bank-conflict free)
  // (the scan of the tallies borrows the first 2 KB of the slots, which are cleared again once it is done: 39 KB of LDS per
  // workgroup = FOUR per CU.  The 1017 workgroups of nine 12 MP photos then run side by side; with three per CU (768
  // places) a quarter of them ran as a second round behind the others, and the launch took two decode times instead of one.)
  uint32_t (*sc)[kWriteThreads] = reinterpret_cast<uint32_t (*)[kWriteThreads]>(slots);
  static_assert(4 * kWriteThreads <= kWriteThreads * 33, "the scan arrays fit in the slots");
  __shared__ uint32_t base[4];
  const int tid = threadIdx.x;
  const int g0 = blockIdx.x * kWriteThreads, g = g0 + tid;
  const DevImg* gimg = &A.imgs[A.sub_img[blockIdx.x]];
  const uint32_t first_bit = static_cast<uint32_t>(g0 - gimg->first_sub) * static_cast<uint32_t>(kSubBits);
  uint32_t* slot = slots + tid * 33;
#pragma unroll
  for (int q = 0; q < 32; ++q) slot[q] = 0u;
  if (tid < 4) base[tid] = 0u;
  load_record(&sh, gimg);
  load_stream(&sh, gimg, first_bit);
  const DevImg& I = sh.img;
  const uint32_t i = static_cast<uint32_t>(g - I.first_sub);
  const bool live = i < static_cast<uint32_t>(I.n_sub);
  // base of the workgroup: the halves of this image in front of it
  {
    uint32_t acc[4] = {0u, 0u, 0u, 0u};
    for (int u = I.first_sub / kWriteThreads + tid; u < static_cast<int>(blockIdx.x); u += kWriteThreads)
      for (int c = 0; c < 4; ++c) acc[c] += A.half_total[4 * u + c];
    for (int c = 0; c < 4; ++c) if (acc[c]) atomicAdd(&base[c], acc[c]);
  }
  // exclusive scan of the workgroup's own tallies
  uint32_t mine[4];
  for (int c = 0; c < 4; ++c) { mine[c] = live ? A.tally[4 * g + c] : 0u; sc[c][tid] = mine[c]; }
  __syncthreads();
  for (int off = 1; off < kWriteThreads; off <<= 1) {
    uint32_t t[4];
    for (int c = 0; c < 4; ++c) t[c] = tid >= off ? sc[c][tid - off] : 0u;
    __syncthreads();
    for (int c = 0; c < 4; ++c) sc[c][tid] += t[c];
    __syncthreads();
  }
  uint32_t ex[4];
  for (int c = 0; c < 4; ++c) ex[c] = base[c] + sc[c][tid] - mine[c];
  __syncthreads();                                               // every thread has its sums: the borrowed words are slots again
  for (int k = tid; k < 4 * kWriteThreads; k += kWriteThreads) slots[k] = 0u;
  __syncthreads();
  if (!live) return;
  State S;
  if (i == 0) { S.p = 0; S.c = 0; S.z = 0; } else { S.p = A.p[g - 1]; S.c = A.cz[g - 1] >> 8; S.z = A.cz[g - 1] & 255u; }
  const uint64_t end = static_cast<uint64_t>(i + 1) * static_cast<uint64_t>(kSubBits);
  const uint32_t limit = static_cast<uint32_t>(end < static_cast<uint64_t>(I.bits) ? end : static_cast<uint64_t>(I.bits));
  run_write(&sh, first_bit, slot, S, limit, ex[0], static_cast<int32_t>(ex[1]), static_cast<int32_t>(ex[2]), static_cast<int32_t>(ex[3]));
}

}  // namespace

int jpeg_gpu_entropy_decode(const std::vector<JpegGpuItem>& items, std::vector<uint8_t>* ok, void* stream_, void** scratch, size_t* scratch_bytes) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const size_t n_img = items.size();
  static const bool timing = std::getenv("IST_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    (void)hipStreamSynchronize(stream);
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[ist timing]   huffman/%-20s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  ok->assign(n_img, 0);
  if (n_img == 0) return IST_OK;
#define JG_HIP(e) do { const hipError_t e_ = (e); if (e_ != hipSuccess) return fail(IST_E_HIP, std::string(#e) + ": " + hipGetErrorString(e_)); } while (0)
  // ---- one device arena: streams, tables, image records, per-subsequence state
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~static_cast<size_t>(255); return at; };
  // A unit = what the kernels call an image: a whole scan, or ONE RESTART INTERVAL of a scan with DRI (the coder starts
  // afresh behind every RSTn, so an interval is an independent stream whose blocks start at MCU mcu0 of the same planes).
  struct Unit { size_t item; uint32_t byte_off; };
  std::vector<Unit> units;
  std::vector<size_t> o_stream(n_img), o_tab(n_img), first_unit(n_img + 1);
  std::vector<DevImg> H;
  int64_t n_sub_total = 0;
  // (the group -> unit map is 16 bits wide: when a batch has more units than that, the files with the most restart intervals
  // are left to the host decoder - their ok[] stays 0 - until the rest fits)
  std::vector<char> skip(n_img, 0);
  {
    size_t total = 0;
    for (size_t k = 0; k < n_img; ++k) total += items[k].S->iv.empty() ? 1 : items[k].S->iv.size();
    while (total > 65535) {
      size_t worst = n_img, most = 1;
      for (size_t k = 0; k < n_img; ++k) if (!skip[k] && items[k].S->iv.size() > most) { most = items[k].S->iv.size(); worst = k; }
      if (worst == n_img) break;                       // (only single-unit files left: more than 65535 files cannot happen, kMaxImages)
      skip[worst] = 1; total -= most;
    }
  }
  for (size_t k = 0; k < n_img; ++k) {
    const JpegImage& J = *items[k].J; const JpegGpuScan& S = *items[k].S;
    first_unit[k] = units.size();
    if (skip[k]) { o_stream[k] = 0; continue; }
    if (S.bits >= (1ll << 32) - 65536) return fail(IST_E_UNSUPPORTED, "JPEG scan too large for the GPU entropy decoder");
    // the kernels index slot_comp / slot_idx (10 entries, T.81 B.2.3) with the MCU slot: never launch outside that
    if (S.slots < 1 || S.slots > 10) return fail(IST_E_DECODE, "JPEG scan with more than 10 blocks per MCU");
    if (S.iv.empty() && S.stream.size() != static_cast<size_t>(S.bits / 8) + 16) return fail(IST_E_INVALID, "JPEG scan buffer without its padding");
    // the intervals must tile the frame's MCUs exactly, in order: the per-unit block count below cannot see a missing
    // interval, and the planes of a MCU nobody writes would keep what an earlier call left in the caller's arena
    if (!S.iv.empty()) {
      int64_t at = 0;
      bool tiled = true;
      for (const JpegGpuInterval& V : S.iv) { tiled = tiled && V.n_mcus > 0 && static_cast<int64_t>(V.mcu0) == at; at += V.n_mcus; }
      if (!tiled || at != static_cast<int64_t>(J.mcus_x) * J.mcus_y) { skip[k] = 1; o_stream[k] = 0; continue; }   // ok[k] stays 0: the host decodes (and reports)
    }
    o_stream[k] = items[k].d_stream ? 0 : take(S.stream.size());
    const size_t n_units = S.iv.empty() ? 1 : S.iv.size();
    for (size_t u = 0; u < n_units; ++u) {
      DevImg I;
      std::memset(&I, 0, sizeof I);
      uint32_t byte_off = 0;
      if (S.iv.empty()) {
        I.bits = S.bits;
        I.total_blocks = J.mcus_x * J.mcus_y * S.slots;
      } else {
        const JpegGpuInterval& V = S.iv[u];
        if (V.byte_off % 256 != 0 || V.bits < 0 || V.bits % 8 != 0 || static_cast<size_t>(V.byte_off) + static_cast<size_t>(V.bits / 8) + 16 > S.stream.size() ||
            static_cast<int64_t>(V.mcu0) + V.n_mcus > static_cast<int64_t>(J.mcus_x) * J.mcus_y)
          return fail(IST_E_INVALID, "JPEG restart interval outside its scan");
        byte_off = V.byte_off;
        I.bits = V.bits;
        I.total_blocks = static_cast<int32_t>(V.n_mcus) * S.slots;
        I.mcu0 = static_cast<int32_t>(V.mcu0);
      }
      I.first_sub = static_cast<int32_t>(n_sub_total & 0x7fffffff);
      I.n_sub = static_cast<int32_t>((I.bits + kSubBits - 1) / kSubBits);
      if (I.n_sub < 1) I.n_sub = 1;
      n_sub_total += (static_cast<int64_t>(I.n_sub) + (kSyncThreads - 1)) & ~static_cast<int64_t>(kSyncThreads - 1);        // a workgroup never spans two units
      I.slots = S.slots; I.mcus_x = J.mcus_x;
      std::memcpy(I.slot_comp, S.slot_comp, 10); std::memcpy(I.slot_idx, S.slot_idx, 10);
      std::memcpy(I.dc_tab, S.dc_tab, 3); std::memcpy(I.ac_tab, S.ac_tab, 3);
      for (int c = 0; c < J.ncomp; ++c) {
        I.coef[c] = items[k].d_coef[c];
        I.h[c] = J.comp[c].h; I.v[c] = J.comp[c].v; I.blocks_x[c] = J.comp[c].blocks_x;
      }
      for (int c = J.ncomp; c < 3; ++c) I.h[c] = I.v[c] = 1;
      H.push_back(I);
      units.push_back(Unit{k, byte_off});
    }
  }
  first_unit[n_img] = units.size();
  const size_t n_unit = units.size();
  if (n_unit > 65535 || n_unit == 0) return IST_OK;          // (every file skipped: the host decodes)
  if (n_sub_total >= (1ll << 31)) return fail(IST_E_UNSUPPORTED, "too much JPEG data for one GPU entropy-decode batch");
  const int ns = static_cast<int>(n_sub_total);
  const int n_half = ns / kWriteThreads;            // groups of 128 subsequences (ns is a multiple of kSyncThreads)
  // the small inputs (Huffman tables, image records, the group -> image map) are ONE contiguous region, uploaded by ONE copy
  // from a pinned block: as nine pageable copies of 10 KB each they were blocking staged copies, ~0.2 ms of the call
  const size_t o_small = off;
  for (size_t k = 0; k < n_img; ++k) o_tab[k] = take(sizeof(items[k].S->tables));
  const size_t o_img = take(sizeof(DevImg) * n_unit), o_sub = take(2 * static_cast<size_t>(n_half));
  const size_t small_bytes = off - o_small;
  const size_t o_p0 = take(4 * static_cast<size_t>(ns)), o_p1 = take(4 * static_cast<size_t>(ns)), o_cz0 = take(4 * static_cast<size_t>(ns)), o_cz1 = take(4 * static_cast<size_t>(ns));
  const size_t o_sp = take(4 * static_cast<size_t>(ns)), o_scz = take(4 * static_cast<size_t>(ns));
  const size_t o_checks = take(24 * static_cast<size_t>(ns));
  const size_t o_tally = take(16 * static_cast<size_t>(ns)), o_half = take(16 * static_cast<size_t>(n_half)), o_flag = take(4), o_err = take(4 * n_unit);
  // the caller's grow-only scratch (a context keeps it across calls: no allocation, and no implicit device synchronisation
  // of a free, per call), or a one-off allocation
  uint8_t* d = nullptr;
  struct Free { void* p; ~Free() { dev_free(p); } } fr{nullptr};
  if (scratch && scratch_bytes) {
    if (*scratch_bytes < off) {
      if (*scratch) { dev_free(*scratch); *scratch = nullptr; *scratch_bytes = 0; }
      if (dev_malloc(scratch, off + off / 4) != 0) { (void)hipGetLastError(); return fail(IST_E_NOMEM, "out of device memory for the Huffman decoder"); }
      *scratch_bytes = off + off / 4;
    }
    d = static_cast<uint8_t*>(*scratch);
  } else {
    if (dev_malloc(reinterpret_cast<void**>(&d), off) != 0) { (void)hipGetLastError(); return fail(IST_E_NOMEM, "out of device memory for the Huffman decoder"); }
    fr.p = d;
  }
  // pinned block: [small inputs | results: flag, half totals, error words]
  const size_t res_bytes = 256 + 16 * static_cast<size_t>(n_half) + 4 * n_unit;
  struct Pin { uint8_t* p; ~Pin() { if (p) pool_give(p); } } pin{static_cast<uint8_t*>(pool_take(small_bytes + res_bytes))};
  if (!pin.p) return fail(IST_E_NOMEM, "out of pinned host memory for the Huffman decoder");
  // (every return below leaves the stream idle before `pin` goes back to the pool: the copies into it are waited for)
  struct Idle { hipStream_t s; ~Idle() { (void)hipStreamSynchronize(s); } } idle{stream};
  uint8_t* hs = pin.p;                                 // host image of the small-input region
  volatile uint32_t* h_flag = reinterpret_cast<volatile uint32_t*>(pin.p + small_bytes);
  uint32_t* h_half = reinterpret_cast<uint32_t*>(pin.p + small_bytes + 256);
  uint32_t* h_err = h_half + 4 * static_cast<size_t>(n_half);
  std::memset(hs, 0, small_bytes);
  uint16_t* half_img = reinterpret_cast<uint16_t*>(hs + (o_sub - o_small));   // unit of every group of 128 subsequences
  for (size_t k = 0; k < n_img; ++k) {
    const JpegGpuScan& S = *items[k].S;
    if (skip[k]) continue;
    if (!items[k].d_stream) JG_HIP(hipMemcpyAsync(d + o_stream[k], S.stream.data(), S.stream.size(), hipMemcpyHostToDevice, stream));
    std::memcpy(hs + (o_tab[k] - o_small), &S.tables, sizeof(S.tables));
  }
  for (size_t u = 0; u < n_unit; ++u) {
    const size_t k = units[u].item;
    H[u].stream = (items[k].d_stream ? items[k].d_stream : d + o_stream[k]) + units[u].byte_off;
    H[u].tables = reinterpret_cast<const JpegGpuTables*>(d + o_tab[k]);
    H[u].err = reinterpret_cast<uint32_t*>(d + o_err) + u;
    for (int i = 0; i < ((H[u].n_sub + (kSyncThreads - 1)) & ~(kSyncThreads - 1)); i += kWriteThreads) half_img[static_cast<size_t>((H[u].first_sub + i) / kWriteThreads)] = static_cast<uint16_t>(u);
    // (the writing pass stores every block of the planes whole, DC included: no clearing pass, no DC pass)
  }
  JG_HIP(hipMemsetAsync(d + o_err, 0, 4 * n_unit, stream));
  std::memcpy(hs + (o_img - o_small), H.data(), sizeof(DevImg) * n_unit);
  JG_HIP(hipMemcpyAsync(d + o_small, hs, small_bytes, hipMemcpyHostToDevice, stream));
  lap("alloc + uploads");
  const DevImg* d_img = reinterpret_cast<const DevImg*>(d + o_img);
  const uint16_t* d_sub = reinterpret_cast<const uint16_t*>(d + o_sub);
  uint32_t* P[2] = {reinterpret_cast<uint32_t*>(d + o_p0), reinterpret_cast<uint32_t*>(d + o_p1)};
  uint32_t* CZ[2] = {reinterpret_cast<uint32_t*>(d + o_cz0), reinterpret_cast<uint32_t*>(d + o_cz1)};
  uint32_t* d_tally = reinterpret_cast<uint32_t*>(d + o_tally);
  uint32_t* d_half = reinterpret_cast<uint32_t*>(d + o_half);
  uint32_t* d_flag = reinterpret_cast<uint32_t*>(d + o_flag);
  // ---- synchronisation passes until a pass changes nothing
  int cur = 0, passes_run = 0;
  bool converged = false;
  for (int pass = 0; pass < kMaxPasses; ++pass) {
    passes_run = pass + 1;
    JG_HIP(hipMemsetAsync(d_flag, 0, 4, stream));
    SyncArgs A{d_img, d_sub, P[cur], CZ[cur], P[cur ^ 1], CZ[cur ^ 1], d_tally, reinterpret_cast<uint32_t*>(d + o_sp), reinterpret_cast<uint32_t*>(d + o_scz), reinterpret_cast<uint32_t*>(d + o_checks), d_half, d_flag, pass};
    hipLaunchKernelGGL(ist_jpeg_sync_kernel, dim3(static_cast<unsigned>(ns / kSyncThreads)), dim3(kSyncBlock), 0, stream, A);
    JG_HIP(hipGetLastError());
    cur ^= 1;
#ifdef IST_SYNC_DEBUG
    if (pass == 0) {
      JG_HIP(hipStreamSynchronize(stream));
      const unsigned nwg = std::min<unsigned>(8192u, static_cast<unsigned>(ns / kSyncThreads));
      std::vector<uint32_t> dbg(8 * static_cast<size_t>(nwg));
      JG_HIP(hipMemcpyFromSymbol(dbg.data(), HIP_SYMBOL(g_sync_dbg), dbg.size() * 4));
      std::vector<uint32_t> tot, p0, st, passes, longest, cyc, iters; uint64_t redo = 0; uint32_t tmin = 0, tmax = 0;
      for (unsigned w = 0; w < nwg; ++w) { const uint32_t* q = &dbg[8 * w]; if (!q[7]) continue; passes.push_back(q[0]); tot.push_back(q[1]); st.push_back(q[2]); p0.push_back(q[3]); longest.push_back(q[4]); redo += q[5];
        cyc.push_back(q[6]); iters.push_back(q[7] & 0xFFFFu); tmax += (q[7] >> 16) & 0x7FFFu; }
      auto pct = [](std::vector<uint32_t> v, double f) { std::sort(v.begin(), v.end()); return v.empty() ? 0u : v[static_cast<size_t>(f * (v.size() - 1))]; };
      std::fprintf(stderr, "[sync debug] %zu workgroups; ticks of 10 ns.  total p50 %u p90 %u max %u | staging p50 %u max %u | pass 0 p50 %u p90 %u max %u | passes p50 %u p90 %u max %u | longest later pass p50 %u max %u | lane re-decodes %llu | pass 0: clock64 ticks p50 %u, loop trips of the busiest lane p50 %u; misses (general step, all lanes of all workgroups, pass 0) %u\n",
                   tot.size(), pct(tot, .5), pct(tot, .9), pct(tot, 1.0), pct(st, .5), pct(st, 1.0), pct(p0, .5), pct(p0, .9), pct(p0, 1.0), pct(passes, .5), pct(passes, .9), pct(passes, 1.0),
                   pct(longest, .5), pct(longest, 1.0), static_cast<unsigned long long>(redo), pct(cyc, .5), pct(iters, .5), tmax - tmin);
    }
#endif
    if (pass == 0) continue;                       // (the first launch starts from guesses: a second one always runs)
    *h_flag = 1u;
    JG_HIP(hipMemcpyAsync(const_cast<uint32_t*>(h_flag), d_flag, 4, hipMemcpyDeviceToHost, stream));
    JG_HIP(hipStreamSynchronize(stream));
    tl_mark("huffman: synchronisation launch done, pass", static_cast<long>(pass));
    if (!*h_flag) { converged = true; break; }
  }
  if (timing) std::fprintf(stderr, "[ist timing] GPU Huffman: %d subsequences of %d bits, %s after %d launches\n", ns, kSubBits, converged ? "fixed point" : "NO fixed point", passes_run);
  if (!converged) { JG_HIP(hipStreamSynchronize(stream)); return IST_OK; }      // every ok[] stays 0: the host decodes
  lap("sync passes");
  // ---- the writing pass
  WriteArgs W{d_img, d_sub, P[cur], CZ[cur], d_tally, d_half};
  hipLaunchKernelGGL(ist_jpeg_write_kernel, dim3(static_cast<unsigned>(n_half)), dim3(kWriteThreads), 0, stream, W);
  JG_HIP(hipGetLastError());
  lap("write");
  // ---- validation: exactly the blocks the frame header promises, and nothing the host decoder would reject
  const uint32_t* half = h_half; const uint32_t* err = h_err;
  JG_HIP(hipMemcpyAsync(h_half, d_half, 16 * static_cast<size_t>(n_half), hipMemcpyDeviceToHost, stream));
  JG_HIP(hipMemcpyAsync(h_err, d + o_err, 4 * n_unit, hipMemcpyDeviceToHost, stream));
  JG_HIP(hipStreamSynchronize(stream));
  tl_mark("huffman: writing pass done, validation read back");
  for (size_t k = 0; k < n_img; ++k) {
    bool good = !skip[k];
    for (size_t w = first_unit[k]; w < first_unit[k + 1]; ++w) {              // every restart interval holds exactly its MCUs
      const DevImg& I = H[w];
      uint32_t blocks = 0;
      for (int u = I.first_sub / kWriteThreads; u < (I.first_sub + ((I.n_sub + (kSyncThreads - 1)) & ~(kSyncThreads - 1))) / kWriteThreads; ++u) blocks += half[4 * static_cast<size_t>(u)];
      good = good && blocks == static_cast<uint32_t>(I.total_blocks) && err[w] == 0;
    }
    (*ok)[k] = good ? 1 : 0;
  }
  lap("validation");
#undef JG_HIP
  return IST_OK;
}

}  // namespace ist
