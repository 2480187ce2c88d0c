// ist_jpeg.cpp — baseline / extended-sequential JPEG: container parsing + Huffman entropy decoding on the host.
//
// Reference anchor: loadImageFrom (utils/canvas.js:27-121) hands the file to the platform decoder; 'jpg'/'jpeg' head
// SUPPORTED_IMAGE_TYPES (pages/index/index.js:4) and phone photos are JPEGs, so this is the decode step in front of the
// stitch path.  The orientation the planner needs (index.js:734) is read from the EXIF
// APP1 segment here.
//
// Split for the hardware: entropy decoding is a serial bit stream (host, this file); everything after it — dequantise,
// 8x8 inverse DCT, chroma upsampling, YCbCr->RGB — is independent per block / per pixel and runs on the GPU
// (ist_jpeg_kernels.hip) on the coefficient planes this file produces.
// Supported: SOF0 / SOF1, 8-bit, 1 or 3 components, any sampling factors h,v in {1,2} with luma >= chroma, interleaved
// and non-interleaved scans, restart intervals; progressive (SOF2: DC/AC first and refinement scans, EOB runs) into dense
// coefficient planes.  Arithmetic coding, lossless, 12-bit, CMYK: IST_E_UNSUPPORTED.
// Exactly ONE frame header per file (a second SOF is JERR_SOF_DUPLICATE in libjpeg): every buffer below is sized from it.
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ist_internal.h"
#include "ist_jpeg.h"

#include <emmintrin.h>

namespace ist {

namespace {

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                             15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool present = false;
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  // canonical decoding
  int32_t maxcode[18];
  int32_t valptr[17];
  int32_t mincode[17];
  // 9-bit lookahead: (length << 8) | symbol, 0 = not resolved
  uint16_t look[512];
  // AC tables only: when code + magnitude bits fit in the same 9 bits, the whole coefficient comes from one look-up:
  // (value << 8) | (run << 4) | (code length + magnitude bits), 0 = take the general path
  int16_t fast_ac[512];
  bool build() {                        // false: the code lengths over-subscribe the code space (a corrupt table)
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      k += bits[l];
      code += bits[l];
      if (code > (1 << l)) return false;
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    std::memset(look, 0, sizeof look);
    code = 0; k = 0;
    for (int l = 1; l <= 9; ++l) {
      for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
        const int lo = code << (9 - l);
        for (int f = 0; f < (1 << (9 - l)); ++f) look[lo + f] = static_cast<uint16_t>((l << 8) | vals[k]);
      }
      code <<= 1;
    }
    for (int i = 0; i < 512; ++i) {
      fast_ac[i] = 0;
      const uint16_t e = look[i];
      if (!e) continue;
      const int len = e >> 8, rs = e & 0xFF, run = rs >> 4, mag = rs & 15;
      if (mag == 0 || len + mag > 9) continue;
      int k = ((i << len) & 511) >> (9 - mag);               // the magnitude bits that follow the code
      if (k < (1 << (mag - 1))) k += -(1 << mag) + 1;        // EXTEND (T.81 F.2.2.1)
      if (k >= -128 && k <= 127) fast_ac[i] = static_cast<int16_t>(k * 256 + run * 16 + len + mag);
    }
    return true;
  }
};

struct BitReader {
  const uint8_t* p; const uint8_t* end;
  uint64_t acc = 0; int n = 0;
  bool hit_marker = false;
  void fill() {
    // fast path: the next six bytes hold no 0xFF (no stuffing, no marker): take them at once
    if (n <= 16 && !hit_marker && end - p >= 8) {
      uint64_t w = 0;
      for (int i = 0; i < 8; ++i) w = (w << 8) | p[i];        // big-endian load (the compiler makes it a bswap)
      const uint64_t v = ~w | 0xFFFFull;                      // only the upper six bytes are consumed
      if (!((v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull)) {
        acc |= (w >> 16) << (16 - n);
        n += 48; p += 6;
        return;
      }
    }
    while (n <= 56) {
      uint32_t b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) p += 2;            // stuffed zero
          else { hit_marker = true; b = 0; }                  // a marker: feed zeros, do not consume
        } else ++p;
      }
      acc |= static_cast<uint64_t>(b) << (56 - n);
      n += 8;
    }
  }
  inline uint32_t peek(int k) { if (n < k) fill(); return static_cast<uint32_t>(acc >> (64 - k)); }
  inline void skip(int k) { acc <<= k; n -= k; }
  inline uint32_t get(int k) { if (k == 0) return 0; const uint32_t v = peek(k); skip(k); return v; }
  void reset() { acc = 0; n = 0; hit_marker = false; }
};

inline int decode_symbol(BitReader& br, const Huff& h) {
  const uint32_t v = br.peek(16);
  const uint16_t e = h.look[v >> 7];
  if (e) { br.skip(e >> 8); return e & 0xFF; }
  int l = 10;
  int32_t code = static_cast<int32_t>(v >> 6);
  while (code > h.maxcode[l]) { if (++l > 16) return -1; code = static_cast<int32_t>(v >> (16 - l)); }
  br.skip(l);
  return h.vals[(h.valptr[l] + code - h.mincode[l]) & 255];
}

inline int extend(uint32_t v, int s) { if (s <= 0) return 0; return (v < (1u << (s - 1))) ? static_cast<int>(v) - (1 << s) + 1 : static_cast<int>(v); }

inline uint32_t be16(const uint8_t* p) { return (uint32_t(p[0]) << 8) | p[1]; }

// EXIF orientation (tag 0x0112) from an APP1 "Exif\0\0" payload
int exif_orientation(const uint8_t* d, size_t n) {
  if (n < 14 || std::memcmp(d, "Exif\0\0", 6) != 0) return 0;
  const uint8_t* t = d + 6; const size_t tn = n - 6;
  const bool le = t[0] == 'I' && t[1] == 'I';
  if (!le && !(t[0] == 'M' && t[1] == 'M')) return 0;
  auto r16 = [&](size_t o) -> uint32_t { return le ? (t[o] | (t[o + 1] << 8)) : ((t[o] << 8) | t[o + 1]); };
  auto r32 = [&](size_t o) -> uint32_t { return le ? (t[o] | (t[o + 1] << 8) | (t[o + 2] << 16) | (uint32_t(t[o + 3]) << 24))
                                                  : ((uint32_t(t[o]) << 24) | (t[o + 1] << 16) | (t[o + 2] << 8) | t[o + 3]); };
  if (tn < 8 || r16(2) != 42) return 0;
  size_t ifd = r32(4);
  if (ifd + 2 > tn) return 0;
  const uint32_t cnt = r16(ifd);
  for (uint32_t i = 0; i < cnt; ++i) {
    const size_t e = ifd + 2 + 12 * i;
    if (e + 12 > tn) break;
    if (r16(e) == 0x0112) { const uint32_t v = r16(e + 8); return (v >= 1 && v <= 8) ? static_cast<int>(v) : 0; }
  }
  return 0;
}

}  // namespace

constexpr uint32_t kMaxGpuIntervals = 2048;

static int jpeg_parse_inner(const uint8_t* f, int64_t n, JpegImage* J, bool header_only, JpegGpuScan* gs) {
  if (!f || n < 4 || f[0] != 0xFF || f[1] != 0xD8) return fail(IST_E_DECODE, "not a JPEG file");
  Huff dc[4], ac[4];
  uint16_t qt[4][64]; bool have_q[4] = {false, false, false, false};
  int restart_interval = 0;
  bool have_sof = false, progressive = false;
  int64_t pos = 2;
  *J = JpegImage();
  while (pos + 4 <= n) {
    if (f[pos] != 0xFF) { ++pos; continue; }
    const int m = f[pos + 1];
    if (m == 0xFF) { ++pos; continue; }
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { pos += 2; continue; }
    if (m == 0xD9) break;
    const int64_t len = be16(f + pos + 2);
    if (len < 2 || pos + 2 + len > n) return fail(IST_E_DECODE, "truncated JPEG segment");
    const uint8_t* d = f + pos + 4; const int64_t dl = len - 2;
    if (m == 0xE1 && J->orientation == 0) J->orientation = exif_orientation(d, static_cast<size_t>(dl));
    else if (m == 0xDB) {                                           // DQT
      int64_t o = 0;
      while (o < dl) {
        const int pq = d[o] >> 4, tq = d[o] & 15; ++o;
        if (tq > 3 || o + (pq ? 128 : 64) > dl) return fail(IST_E_DECODE, "bad JPEG quantisation table");
        for (int i = 0; i < 64; ++i) { qt[tq][kZigzag[i]] = static_cast<uint16_t>(pq ? be16(d + o + 2 * i) : d[o + i]); }
        o += pq ? 128 : 64;
        have_q[tq] = true;
      }
    } else if (m == 0xC4) {                                         // DHT
      int64_t o = 0;
      while (o + 17 <= dl) {
        const int tc = d[o] >> 4, th = d[o] & 15;
        if (tc > 1 || th > 3) return fail(IST_E_DECODE, "bad JPEG Huffman table");
        Huff& h = tc ? ac[th] : dc[th];
        int cnt = 0;
        for (int i = 1; i <= 16; ++i) { h.bits[i] = d[o + i]; cnt += h.bits[i]; }
        if (cnt > 256 || o + 17 + cnt > dl) return fail(IST_E_DECODE, "bad JPEG Huffman table");
        std::memcpy(h.vals, d + o + 17, static_cast<size_t>(cnt));
        h.present = true;
        if (!h.build()) return fail(IST_E_DECODE, "bad JPEG Huffman table");
        o += 17 + cnt;
      }
    } else if (m == 0xDD) { if (dl >= 2) restart_interval = static_cast<int>(be16(d)); }
    else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {                 // SOF0 / SOF1 (sequential), SOF2 (progressive)
      if (have_sof) return fail(IST_E_DECODE, "JPEG with more than one frame header");     // the planes are sized from the first one
      progressive = (m == 0xC2);
      if (dl < 6) return fail(IST_E_DECODE, "bad JPEG frame header");
      if (d[0] != 8) return fail(IST_E_UNSUPPORTED, "only 8-bit JPEG is supported");
      J->height = static_cast<int>(be16(d + 1)); J->width = static_cast<int>(be16(d + 3)); J->ncomp = d[5];
      if (J->width < 1 || J->height < 1) return fail(IST_E_DECODE, "bad JPEG size");
      if (J->ncomp != 1 && J->ncomp != 3) return fail(IST_E_UNSUPPORTED, "only greyscale and YCbCr JPEG are supported");
      if (dl < 6 + 3 * J->ncomp) return fail(IST_E_DECODE, "bad JPEG frame header");
      int hmax = 1, vmax = 1;
      for (int c = 0; c < J->ncomp; ++c) {
        JpegComp& C = J->comp[c];
        C.id = d[6 + 3 * c]; C.h = d[7 + 3 * c] >> 4; C.v = d[7 + 3 * c] & 15; C.tq = d[8 + 3 * c];
        if (C.h < 1 || C.h > 2 || C.v < 1 || C.v > 2 || C.tq > 3) return fail(IST_E_UNSUPPORTED, "JPEG sampling factors beyond 2 are not supported");
        hmax = std::max(hmax, C.h); vmax = std::max(vmax, C.v);
        for (int p = 0; p < c; ++p) if (J->comp[p].id == C.id) return fail(IST_E_DECODE, "JPEG frame names a component twice");
      }
      // colour: luma carries the sampling (4:4:4, 4:2:2, 4:2:0, 4:4:0), both chroma planes are 1x1
      if (J->ncomp == 3 && (J->comp[1].h != 1 || J->comp[1].v != 1 || J->comp[2].h != 1 || J->comp[2].v != 1))
        return fail(IST_E_UNSUPPORTED, "unusual JPEG sampling layout (chroma must be 1x1)");
      if (J->ncomp == 1) { J->comp[0].h = J->comp[0].v = 1; hmax = vmax = 1; }
      J->hmax = hmax; J->vmax = vmax;
      J->mcus_x = (J->width + 8 * hmax - 1) / (8 * hmax); J->mcus_y = (J->height + 8 * vmax - 1) / (8 * vmax);
      for (int c = 0; c < J->ncomp; ++c) {
        JpegComp& C = J->comp[c];
        C.blocks_x = J->mcus_x * C.h; C.blocks_y = J->mcus_y * C.v;
      }
      have_sof = true;
      if (header_only) return IST_OK;
    } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) return fail(IST_E_UNSUPPORTED, "this JPEG process (lossless / arithmetic) is not supported");
    else if (m == 0xDA) {                                           // SOS
      if (!have_sof) return fail(IST_E_DECODE, "JPEG scan before frame header");
      if (dl < 1) return fail(IST_E_DECODE, "bad JPEG scan header");
      const int ns = d[0];
      if (ns < 1 || ns > J->ncomp || dl < 1 + 2 * ns + 3) return fail(IST_E_DECODE, "bad JPEG scan header");
      int ci[3], td[3], ta[3];
      for (int s = 0; s < ns; ++s) {
        const int id = d[1 + 2 * s]; ci[s] = -1;
        for (int c = 0; c < J->ncomp; ++c) if (J->comp[c].id == id) ci[s] = c;
        if (ci[s] < 0) return fail(IST_E_DECODE, "JPEG scan names an unknown component");
        for (int p = 0; p < s; ++p) if (ci[p] == ci[s]) return fail(IST_E_DECODE, "JPEG scan names a component twice");   // T.81 B.2.3
        td[s] = d[2 + 2 * s] >> 4; ta[s] = d[2 + 2 * s] & 15;
        if (td[s] > 3 || ta[s] > 3) return fail(IST_E_DECODE, "JPEG scan uses an undefined Huffman table");
      }
      // spectral selection Ss..Se and successive approximation Ah/Al (T.81 Annex G); a sequential scan is 0..63, 0/0
      const int Ss = d[1 + 2 * ns], Se = d[2 + 2 * ns], Ah = d[3 + 2 * ns] >> 4, Al = d[3 + 2 * ns] & 15;
      if (progressive) {
        if (Ss > Se || Se > 63 || Al > 13 || Ah > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || (Ah != 0 && Ah != Al + 1))
          return fail(IST_E_DECODE, "bad progressive JPEG scan parameters");
      } else if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) return fail(IST_E_DECODE, "bad JPEG scan parameters");
      const bool dc_scan = Ss == 0, need_ac = !progressive || !dc_scan, need_dc = dc_scan && Ah == 0;
      for (int s = 0; s < ns; ++s) {
        if ((need_dc && !dc[td[s]].present) || (need_ac && !ac[ta[s]].present)) return fail(IST_E_DECODE, "JPEG scan uses an undefined Huffman table");
      }
      // the GPU entropy decoder takes baseline files whose single scan interleaves all components (restart intervals or not)
      int mcu_blocks = 0;
      for (int s2 = 0; s2 < ns; ++s2) mcu_blocks += J->comp[ci[s2]].h * J->comp[ci[s2]].v;
      // ... and names at most two DC and two AC tables (the GPU decoder keeps that many in LDS)
      int dc_ids[2] = {-1, -1}, ac_ids[2] = {-1, -1}, dc_local[4] = {0, 0, 0, 0}, ac_local[4] = {0, 0, 0, 0};
      bool two_tables = true;
      for (int s2 = 0; s2 < ns; ++s2) {
        auto local = [&](int (&ids)[2], int id) { for (int q = 0; q < 2; ++q) { if (ids[q] == id) return q; if (ids[q] < 0) { ids[q] = id; return q; } } return -1; };
        dc_local[s2] = local(dc_ids, td[s2]); ac_local[s2] = local(ac_ids, ta[s2]);
        if (dc_local[s2] < 0 || ac_local[s2] < 0) two_tables = false;
      }
      if (gs && !progressive && J->scans == 0 && ns == J->ncomp && two_tables && mcu_blocks <= 10 &&      // (T.81 B.2.3: at most 10 blocks per MCU)
          (f + n) - (d + dl) < (1ll << 28)) {      // (32-bit bit positions on the GPU)
        for (int c = 0; c < J->ncomp; ++c) {
          if (!have_q[J->comp[c].tq]) return fail(IST_E_DECODE, "JPEG component uses an undefined quantisation table");
          std::memcpy(J->comp[c].q, qt[J->comp[c].tq], sizeof J->comp[c].q);
        }
        gs->slots = 0;
        for (int s2 = 0; s2 < ns; ++s2) {
          const JpegComp& C = J->comp[ci[s2]];
          for (int k = 0; k < C.h * C.v; ++k) { gs->slot_comp[gs->slots] = static_cast<uint8_t>(ci[s2]); gs->slot_idx[gs->slots] = static_cast<uint8_t>(k); ++gs->slots; }
          gs->dc_tab[ci[s2]] = static_cast<uint8_t>(dc_local[s2]); gs->ac_tab[ci[s2]] = static_cast<uint8_t>(ac_local[s2]);
        }
        std::memset(&gs->tables, 0, sizeof gs->tables);      // (a table the scan does not name stays all "no code")
        for (int t = 0; t < 4; ++t) {
          const bool is_ac = t >= 2;
          const int id = is_ac ? ac_ids[t - 2] : dc_ids[t];
          if (id < 0) continue;
          const Huff& h = is_ac ? ac[id] : dc[id];
          const int look_bits = is_ac ? kJpegAcLookBits : kJpegDcLookBits;
          uint16_t* look = is_ac ? gs->tables.look_ac[t - 2] : gs->tables.look_dc[t];
          int code = 0, k = 0;
          for (int l = 1; l <= look_bits; ++l) {              // canonical codes in order of length (T.81 C.2)
            for (int i = 0; i < h.bits[l]; ++i, ++k, ++code) {
              const uint16_t e = static_cast<uint16_t>((l << 8) | h.vals[k]);
              uint16_t* at = look + (static_cast<size_t>(code) << (look_bits - l));
              std::fill(at, at + (size_t{1} << (look_bits - l)), e);
            }
            code <<= 1;
          }
          JpegHuffTail& o = gs->tables.tail[t];
          for (int q = 0; q < 8; ++q) {                       // one past the last code of length 9+q, left-aligned
            const int l = 9 + q;
            o.lim[q] = static_cast<uint32_t>(h.mincode[l] + h.bits[l]) << (16 - l);
            if (q < 7) o.vptr[q] = static_cast<uint8_t>(h.valptr[l + 1]);
          }
          std::memcpy(o.vals, h.vals, sizeof o.vals);
        }
        // de-stuff: FF 00 -> FF; the scan ends at the first real marker.  With a restart interval (DRI) every RSTn closes an
        // interval: the entropy coder starts afresh behind it (byte aligned, DC predictors zero), so the intervals are
        // independent streams for the GPU decoder - each is padded to a 256-byte boundary with at least 16 zero bytes.
        // A restart marker out of sequence, or one more than the frame has intervals for, leaves this scan to the host
        // decoder below (which has the resynchronisation rules); so does a scan that ends with fewer intervals than the frame
        // needs (checked behind the loop: the GPU decoder's block count is per interval and would not see it).
        const uint8_t* q = d + dl; const uint8_t* qe = f + n;
        gs->iv.clear();
        const uint32_t total_mcus = static_cast<uint32_t>(J->mcus_x) * static_cast<uint32_t>(J->mcus_y);
        const uint32_t ri = static_cast<uint32_t>(restart_interval);
        const uint32_t n_iv = ri ? (total_mcus + ri - 1) / ri : 0;
        // (every interval occupies whole workgroups of the GPU decoder - 16 KB of bitstream positions - so a file cut into
        // thousands of tiny intervals is cheaper on the host)
        bool in_sequence = n_iv <= kMaxGpuIntervals;
        // the output never overtakes the input; every interval adds at most 271 bytes of padding; 32 bytes of slack for the
        // 16-byte stores of the copy loop and the 16 zero bytes behind the last bit
        gs->stream.clear();
        if (in_sequence && !gs->stream.reserve(static_cast<size_t>(qe - q) + static_cast<size_t>(n_iv) * 272 + 32)) return fail(IST_E_NOMEM, "out of memory for the JPEG scan");
        uint8_t* const base = gs->stream.data();
        uint8_t* out = base;
        size_t iv_start = 0; uint32_t next_rst = 0;
        auto close_interval = [&]() {
          const uint32_t k = static_cast<uint32_t>(gs->iv.size());
          const size_t at = static_cast<size_t>(out - base);
          JpegGpuInterval I;
          I.byte_off = static_cast<uint32_t>(iv_start); I.mcu0 = k * ri; I.n_mcus = std::min(ri, total_mcus - k * ri);
          I.bits = static_cast<int64_t>(at - iv_start) * 8;
          gs->iv.push_back(I);
          size_t pad = (256 - at % 256) % 256;
          if (pad < 16) pad += 256;
          std::memset(out, 0, pad);
          out += pad;
          iv_start = at + pad;
        };
        const __m128i all_ff = _mm_set1_epi8(static_cast<char>(0xFF));
        while (in_sequence && q < qe) {
          // 16 bytes at a time while none of them is FF (a photo's scan has one FF in ~256 bytes): the copy runs ahead of
          // the test, the bytes behind an FF are simply overwritten by the next round
          while (q + 16 <= qe) {
            const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(q));
            _mm_storeu_si128(reinterpret_cast<__m128i*>(out), v);
            const int m = _mm_movemask_epi8(_mm_cmpeq_epi8(v, all_ff));
            if (m) { const int k = __builtin_ctz(static_cast<unsigned>(m)); q += k; out += k; break; }
            q += 16; out += 16;
          }
          if (q >= qe) break;
          if (*q != 0xFF) { *out++ = *q++; continue; }                            // (the last 15 bytes of the file, one by one)
          if (q + 1 < qe && q[1] == 0x00) { *out++ = 0xFF; q += 2; continue; }
          if (q + 1 < qe && q[1] == 0xFF) { q += 1; continue; }                   // fill byte
          if (q + 1 < qe && q[1] >= 0xD0 && q[1] <= 0xD7) {
            if (!ri || q[1] != 0xD0 + next_rst || gs->iv.size() + 1 >= n_iv) { in_sequence = false; break; }
            close_interval();                                                     // RSTn in sequence, and another interval is due
            next_rst = (next_rst + 1) & 7u;
            q += 2;
            continue;
          }
          break;                                                                  // a marker (or a lone FF at the end)
        }
        if (in_sequence && ri) {
          close_interval();                                                       // the last interval (ends at the marker that ends the scan)
          // A scan that ends after k < n_iv intervals (EOI or the end of the file right behind an RSTn) would pass the GPU
          // decoder's PER-INTERVAL block count while the MCUs behind k*ri are never written: the planes there would still
          // hold what an earlier call left in the arena.  Such a file is the host decoder's (which reports it).
          if (gs->iv.size() != n_iv) in_sequence = false;
        }
        if (in_sequence) {
          if (ri) {
            gs->bits = static_cast<int64_t>(out - base) * 8;
          } else {
            gs->bits = static_cast<int64_t>(out - base) * 8;
            std::memset(out, 0, 16);
            out += 16;
          }
          gs->stream.set_size(static_cast<size_t>(out - base));
          gs->eligible = true;
          pos = q - f;
          J->scans++;
          continue;
        }
        gs->stream.clear(); gs->iv.clear();
      }
      // allocate coefficient planes on first use; copy the quantisation tables in use
      for (int c = 0; c < J->ncomp; ++c) {
        JpegComp& C = J->comp[c];
        if (!have_q[C.tq]) return fail(IST_E_DECODE, "JPEG component uses an undefined quantisation table");
        std::memcpy(C.q, qt[C.tq], sizeof C.q);
        const size_t nblk = static_cast<size_t>(C.blocks_x) * C.blocks_y;
        if (progressive) {
          if (C.coef.empty() && !C.coef.alloc_zero(nblk * 64)) return fail(IST_E_NOMEM, "out of memory for the JPEG coefficients");
        } else if (!C.sparse) {
          C.sparse = true;
          C.start.assign(nblk, 0); C.cnt.assign(nblk, 0);
          C.ent.reserve(2 * static_cast<size_t>(n) + 1024);      // virtual only: a photo has ~1.3 entries per file byte, untouched pages cost nothing
        }
      }
      BitReader br; br.p = d + dl; br.end = f + n;
      int pred[3] = {0, 0, 0}, eobrun = 0;
      const bool interleaved = ns > 1;
      int mx, my;
      if (interleaved) { mx = J->mcus_x; my = J->mcus_y; }
      else {   // a non-interleaved scan covers only the blocks that hold image samples
        const JpegComp& C = J->comp[ci[0]];
        mx = (((J->width * C.h + J->hmax - 1) / J->hmax) + 7) / 8;
        my = (((J->height * C.v + J->vmax - 1) / J->vmax) + 7) / 8;
      }
      int until_restart = restart_interval, next_rst = 0;
      for (int y = 0; y < my; ++y) {
        for (int x = 0; x < mx; ++x) {
          if (restart_interval && until_restart == 0) {
            // byte-align and expect RSTn
            br.reset();
            while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
            if (br.p + 1 >= br.end) return fail(IST_E_DECODE, "JPEG restart marker missing");
            if ((br.p[1] & 7) != next_rst) return fail(IST_E_DECODE, "JPEG restart markers out of order");
            br.p += 2; next_rst = (next_rst + 1) & 7;
            pred[0] = pred[1] = pred[2] = 0; eobrun = 0;
            until_restart = restart_interval;
          }
          for (int s = 0; s < ns; ++s) {
            JpegComp& C = J->comp[ci[s]];
            const int bh = interleaved ? C.h : 1, bv = interleaved ? C.v : 1;
            for (int by = 0; by < bv; ++by) for (int bx = 0; bx < bh; ++bx) {
              const int gx = x * bh + bx, gy = y * bv + by;
              const size_t bi = static_cast<size_t>(gy) * C.blocks_x + gx;
              int16_t* blk = progressive ? C.coef.data() + bi * 64 : nullptr;
              if (!progressive) {                            // sequential: the block's non-zero coefficients as sparse entries
                const size_t first = C.ent.size();
                C.ent.resize(first + 64);                   // room for a whole block, trimmed below (no per-entry capacity check)
                uint32_t* e = C.ent.data() + first;
                int t = decode_symbol(br, dc[td[s]]);
                if (t < 0 || t > 11) return fail(IST_E_DECODE, "corrupt JPEG entropy data (DC)");
                const int diff = t ? extend(br.get(t), t) : 0;
                pred[s] += diff;
                if (pred[s]) *e++ = static_cast<uint32_t>(static_cast<uint16_t>(static_cast<int16_t>(pred[s])));
                const Huff& hac = ac[ta[s]];
                for (int k = 1; k < 64;) {
                  const int fa = hac.fast_ac[br.peek(16) >> 7];
                  if (fa) {                                  // run, value and both bit counts from one look-up
                    k += (fa >> 4) & 15;
                    if (k > 63) return fail(IST_E_DECODE, "corrupt JPEG entropy data (run)");
                    br.skip(fa & 15);
                    *e++ = (static_cast<uint32_t>(kZigzag[k]) << 16) | static_cast<uint16_t>(static_cast<int16_t>(fa >> 8));
                    ++k;
                    continue;
                  }
                  const int rs = decode_symbol(br, hac);
                  if (rs < 0) return fail(IST_E_DECODE, "corrupt JPEG entropy data (AC)");
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                  k += r;
                  if (k > 63) return fail(IST_E_DECODE, "corrupt JPEG entropy data (run)");
                  const int v = extend(br.get(sz), sz);
                  if (v) *e++ = (static_cast<uint32_t>(kZigzag[k]) << 16) | static_cast<uint16_t>(static_cast<int16_t>(v));
                  ++k;
                }
                const size_t used = static_cast<size_t>(e - (C.ent.data() + first));      // <= 64
                C.ent.resize(first + used);
                C.start[bi] = static_cast<uint32_t>(first);
                C.cnt[bi] = static_cast<uint8_t>(used);
              } else if (dc_scan && Ah == 0) {               // DC first pass: the difference, scaled by 2^Al
                const int t = decode_symbol(br, dc[td[s]]);
                if (t < 0 || t > 11) return fail(IST_E_DECODE, "corrupt JPEG entropy data (DC)");
                pred[s] += t ? extend(br.get(t), t) : 0;
                blk[0] = static_cast<int16_t>(pred[s] * (1 << Al));
              } else if (dc_scan) {                          // DC refinement: one more bit of every DC coefficient
                if (br.get(1)) blk[0] = static_cast<int16_t>(blk[0] | (1 << Al));
              } else if (Ah == 0) {                          // AC first pass over the band Ss..Se
                if (eobrun > 0) { --eobrun; continue; }
                for (int k = Ss; k <= Se;) {
                  const int rs = decode_symbol(br, ac[ta[s]]);
                  if (rs < 0) return fail(IST_E_DECODE, "corrupt JPEG entropy data (AC)");
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz == 0) {
                    if (r == 15) { k += 16; continue; }
                    eobrun = (1 << r) - 1;
                    if (r) eobrun += static_cast<int>(br.get(r));
                    break;
                  }
                  k += r;
                  if (k > Se) return fail(IST_E_DECODE, "corrupt JPEG entropy data (run)");
                  blk[kZigzag[k]] = static_cast<int16_t>(extend(br.get(sz), sz) * (1 << Al));
                  ++k;
                }
              } else {                                       // AC refinement (T.81 G.1.2.3)
                const int p1 = 1 << Al, m1 = -(1 << Al);
                auto refine = [&](int16_t* c) {              // a correction bit for a coefficient that is already non-zero
                  if (br.get(1) && (*c & p1) == 0) *c = static_cast<int16_t>(*c + (*c >= 0 ? p1 : m1));
                };
                int k = Ss;
                if (eobrun == 0) {
                  for (; k <= Se; ++k) {
                    const int rs = decode_symbol(br, ac[ta[s]]);
                    if (rs < 0) return fail(IST_E_DECODE, "corrupt JPEG entropy data (AC)");
                    int r = rs >> 4, val = 0;
                    const int sz = rs & 15;
                    if (sz) {
                      if (sz != 1) return fail(IST_E_DECODE, "corrupt JPEG entropy data (refinement)");
                      val = br.get(1) ? p1 : m1;
                    } else if (r != 15) {
                      eobrun = 1 << r;
                      if (r) eobrun += static_cast<int>(br.get(r));
                      break;                                 // the rest of this block is handled as part of the EOB run
                    }
                    for (; k <= Se; ++k) {                   // pass already-non-zero coefficients, skip r zero ones
                      int16_t* c = blk + kZigzag[k];
                      if (*c != 0) refine(c);
                      else if (--r < 0) break;
                    }
                    if (val) {
                      if (k > Se) return fail(IST_E_DECODE, "corrupt JPEG entropy data (run)");
                      blk[kZigzag[k]] = static_cast<int16_t>(val);
                    }
                  }
                }
                if (eobrun > 0) {
                  for (; k <= Se; ++k) { int16_t* c = blk + kZigzag[k]; if (*c != 0) refine(c); }
                  --eobrun;
                }
              }
            }
          }
          if (restart_interval) --until_restart;
        }
      }
      // continue after the entropy-coded segment: the reader stopped at (or before) the next marker
      const uint8_t* q = br.p;
      while (q + 1 < f + n && !(q[0] == 0xFF && q[1] != 0x00 && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
      pos = q - f;
      J->scans++;
      continue;
    }
    pos += 2 + len;
  }
  if (!have_sof) return fail(IST_E_DECODE, "JPEG without a frame header");
  if (!header_only && J->scans == 0) return fail(IST_E_DECODE, "JPEG without image data");
  return IST_OK;
}

// the C boundary never lets an exception through: a header may announce planes the host cannot allocate
int jpeg_parse_and_entropy_decode(const uint8_t* f, int64_t n, JpegImage* J, bool header_only, JpegGpuScan* gs) {
  try { return jpeg_parse_inner(f, n, J, header_only, gs); }
  catch (const std::bad_alloc&) { return fail(IST_E_NOMEM, "out of memory while decoding the JPEG"); }
}

std::vector<int16_t> jpeg_dense_coefficients(const JpegComp& c) {
  const size_t nblk = static_cast<size_t>(c.blocks_x) * c.blocks_y;
  std::vector<int16_t> d(nblk * 64, 0);
  if (!c.sparse) { if (c.coef.size() == d.size()) std::memcpy(d.data(), c.coef.data(), d.size() * 2); return d; }
  for (size_t b = 0; b < nblk; ++b)
    for (uint32_t k = 0; k < c.cnt[b]; ++k) { const uint32_t e = c.ent[c.start[b] + k]; d[b * 64 + (e >> 16)] = static_cast<int16_t>(e & 0xFFFFu); }
  return d;
}

}  // namespace ist
