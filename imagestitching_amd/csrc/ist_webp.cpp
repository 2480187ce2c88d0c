// ist_webp.cpp — WebP files -> RGBA8 (straight alpha): the RIFF container and the lossless (VP8L) bitstream; the lossy
// (VP8) bitstream is decoded by ist_webp_vp8.cpp.  'webp' is one of SUPPORTED_IMAGE_TYPES (pages/index/index.js:4) and
// one of the extensions the chooser offers (index.js:1030); the decode step itself is the platform's Image.src
// (utils/canvas.js:27-121), so this is SURVEY.md section 8f rank 3 for the last raster format of that list.
//
// Source of the algorithm: the published specifications — "WebP Container Specification" and "WebP Lossless Bitstream
// Specification" (RFC 9649).  Lossless means the result is pinned: every conforming decoder yields the same ARGB values
// (tests compare with PIL / libwebp bit for bit).
//
// Host code: the stream is one serial entropy-coded sequence with LZ77 back-references and a colour cache, followed by
// inverse transforms whose predictor depends on already reconstructed neighbours — like PNG, nothing to parallelise before
// the pixels exist.  Animated files (ANIM / ANMF) decode to their FIRST frame on its transparent canvas: a still Image shows one frame.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ist_internal.h"
#include "ist_webp.h"

namespace ist {

namespace {

inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (uint32_t(p[3]) << 24); }
inline uint32_t le24(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16); }

// ---- LSB-first bit reader (RFC 9649 section 3.1) --------------------------------------------------------------------
struct Bits {
  const uint8_t* p; const uint8_t* end;
  uint64_t acc = 0; int n = 0;
  bool eos = false;                      // more bits were asked for than the chunk holds
  Bits(const uint8_t* b, size_t len) : p(b), end(b + len) {}
  inline uint32_t read(int k) {          // k <= 32
    if (k == 0) return 0;
    if (n < k) refill();
    const uint32_t v = static_cast<uint32_t>(acc & ((k == 32) ? 0xFFFFFFFFull : ((1ull << k) - 1)));
    acc >>= k; n -= k;
    return v;
  }
  inline uint32_t peek(int k) { if (n < k) refill(); return static_cast<uint32_t>(acc & ((1ull << k) - 1)); }
  inline void skip(int k) { acc >>= k; n -= k; }
  void refill() {
    while (n <= 56) {
      uint64_t b = 0;
      if (p < end) b = *p++;
      else { ++past; if (past > 8) eos = true; }
      acc |= b << n;
      n += 8;
    }
  }
  int past = 0;
};

// ---- prefix codes (canonical Huffman, RFC 9649 section 6.2) ---------------------------------------------------------
constexpr int kLookBits = 8;
struct Code {
  // direct table over the next kLookBits bits: (length << 16) | symbol, 0 when the code is longer; longer codes walk the
  // canonical first-code table
  std::vector<uint32_t> look;
  int32_t first_code[17], first_sym[17], count[17];
  std::vector<uint16_t> sorted;
  int only = -1;                         // a code with one symbol takes no bits
  bool build(const std::vector<uint8_t>& len) {
    std::memset(count, 0, sizeof count);
    int used = 0, last = -1;
    for (size_t s = 0; s < len.size(); ++s) if (len[s]) { if (len[s] > 15) return false; ++count[len[s]]; ++used; last = static_cast<int>(s); }
    only = -1;
    if (used == 0) return false;
    if (used == 1) { only = last; return true; }
    // complete code required (Kraft sum == 1)
    int64_t space = 1 << 15, need = 0;
    for (int l = 1; l <= 15; ++l) need += static_cast<int64_t>(count[l]) << (15 - l);
    if (need != space) return false;
    int code = 0, sym = 0;
    for (int l = 1; l <= 15; ++l) { first_code[l] = code; first_sym[l] = sym; code = (code + count[l]) << 1; sym += count[l]; }
    sorted.assign(static_cast<size_t>(used), 0);
    {
      int next[17];
      for (int l = 1; l <= 15; ++l) next[l] = first_sym[l];
      for (size_t s = 0; s < len.size(); ++s) if (len[s]) sorted[static_cast<size_t>(next[len[s]]++)] = static_cast<uint16_t>(s);
    }
    look.assign(1u << kLookBits, 0);
    for (int l = 1; l <= kLookBits; ++l)
      for (int k = 0; k < count[l]; ++k) {
        const int c = first_code[l] + k;                 // MSB-first code of length l
        uint32_t rev = 0;                                // the stream delivers its bits LSB first
        for (int b = 0; b < l; ++b) rev |= ((c >> (l - 1 - b)) & 1u) << b;
        const uint32_t e = (static_cast<uint32_t>(l) << 16) | sorted[static_cast<size_t>(first_sym[l] + k)];
        for (uint32_t hi = 0; hi < (1u << (kLookBits - l)); ++hi) look[rev | (hi << l)] = e;
      }
    return true;
  }
  inline int decode(Bits& br) const {
    if (only >= 0) return only;
    const uint32_t e = look[br.peek(kLookBits)];
    if (e) { br.skip(static_cast<int>(e >> 16)); return static_cast<int>(e & 0xFFFF); }
    int code = 0;
    for (int l = 1; l <= 15; ++l) {
      code = (code << 1) | static_cast<int>(br.read(1));
      const int idx = code - first_code[l];
      if (idx >= 0 && idx < count[l]) return sorted[static_cast<size_t>(first_sym[l] + idx)];
    }
    return -1;
  }
};

constexpr int kCodeLengthCodes = 19;
const uint8_t kCodeLengthOrder[kCodeLengthCodes] = {17, 18, 0, 1, 2, 3, 4, 5, 16, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};

int read_code(Bits& br, int alphabet, Code* out) {
  std::vector<uint8_t> len(static_cast<size_t>(alphabet), 0);
  if (br.read(1)) {                                      // simple code: one or two symbols
    const int n = static_cast<int>(br.read(1)) + 1;
    const int first_8 = static_cast<int>(br.read(1));
    const int s0 = static_cast<int>(br.read(first_8 ? 8 : 1));
    if (s0 >= alphabet) return fail(IST_E_DECODE, "WebP: prefix code symbol out of range");
    len[static_cast<size_t>(s0)] = 1;
    if (n == 2) {
      const int s1 = static_cast<int>(br.read(8));
      if (s1 >= alphabet) return fail(IST_E_DECODE, "WebP: prefix code symbol out of range");
      len[static_cast<size_t>(s1)] = 1;
    }
  } else {
    std::vector<uint8_t> cl(kCodeLengthCodes, 0);
    const int n = 4 + static_cast<int>(br.read(4));
    if (n > kCodeLengthCodes) return fail(IST_E_DECODE, "WebP: bad code length code count");
    for (int i = 0; i < n; ++i) cl[kCodeLengthOrder[i]] = static_cast<uint8_t>(br.read(3));
    Code clc;
    if (!clc.build(cl)) return fail(IST_E_DECODE, "WebP: bad code length code");
    int max_symbol = alphabet;
    if (br.read(1)) {
      const int length_nbits = 2 + 2 * static_cast<int>(br.read(3));
      max_symbol = 2 + static_cast<int>(br.read(length_nbits));
      if (max_symbol > alphabet) return fail(IST_E_DECODE, "WebP: bad max_symbol");
    }
    int sym = 0, prev = 8;
    while (sym < alphabet) {
      if (max_symbol-- == 0) break;
      const int c = clc.decode(br);
      if (c < 0 || br.eos) return fail(IST_E_DECODE, "WebP: corrupt code lengths");
      if (c < 16) { len[static_cast<size_t>(sym++)] = static_cast<uint8_t>(c); if (c) prev = c; continue; }
      int rep, val = 0;
      if (c == 16) { rep = 3 + static_cast<int>(br.read(2)); val = prev; }
      else if (c == 17) rep = 3 + static_cast<int>(br.read(3));
      else rep = 11 + static_cast<int>(br.read(7));
      if (sym + rep > alphabet) return fail(IST_E_DECODE, "WebP: code length run past the alphabet");
      for (int k = 0; k < rep; ++k) len[static_cast<size_t>(sym++)] = static_cast<uint8_t>(val);
    }
  }
  if (br.eos) return fail(IST_E_DECODE, "WebP: truncated prefix code");
  if (!out->build(len)) return fail(IST_E_DECODE, "WebP: incomplete prefix code");
  return IST_OK;
}

struct Group { Code c[5]; };             // green + length + cache, red, blue, alpha, distance

// ---- LZ77 (RFC 9649 section 5.2.2) ---------------------------------------------------------------------------------
const int8_t kDistMap[120][2] = {
    {0, 1},  {1, 0},  {1, 1},  {-1, 1}, {0, 2},  {2, 0},  {1, 2},  {-1, 2}, {2, 1},  {-2, 1}, {2, 2},  {-2, 2}, {0, 3},  {3, 0},  {1, 3},
    {-1, 3}, {3, 1},  {-3, 1}, {2, 3},  {-2, 3}, {3, 2},  {-3, 2}, {0, 4},  {4, 0},  {1, 4},  {-1, 4}, {4, 1},  {-4, 1}, {3, 3},  {-3, 3},
    {2, 4},  {-2, 4}, {4, 2},  {-4, 2}, {0, 5},  {3, 4},  {-3, 4}, {4, 3},  {-4, 3}, {5, 0},  {1, 5},  {-1, 5}, {5, 1},  {-5, 1}, {2, 5},
    {-2, 5}, {5, 2},  {-5, 2}, {4, 4},  {-4, 4}, {3, 5},  {-3, 5}, {5, 3},  {-5, 3}, {0, 6},  {6, 0},  {1, 6},  {-1, 6}, {6, 1},  {-6, 1},
    {2, 6},  {-2, 6}, {6, 2},  {-6, 2}, {4, 5},  {-4, 5}, {5, 4},  {-5, 4}, {3, 6},  {-3, 6}, {6, 3},  {-6, 3}, {0, 7},  {7, 0},  {1, 7},
    {-1, 7}, {5, 5},  {-5, 5}, {7, 1},  {-7, 1}, {4, 6},  {-4, 6}, {6, 4},  {-6, 4}, {2, 7},  {-2, 7}, {7, 2},  {-7, 2}, {3, 7},  {-3, 7},
    {7, 3},  {-7, 3}, {5, 6},  {-5, 6}, {6, 5},  {-6, 5}, {8, 0},  {4, 7},  {-4, 7}, {7, 4},  {-7, 4}, {8, 1},  {8, 2},  {6, 6},  {-6, 6},
    {8, 3},  {5, 7},  {-5, 7}, {7, 5},  {-7, 5}, {8, 4},  {6, 7},  {-6, 7}, {7, 6},  {-7, 6}, {8, 5},  {7, 7},  {-7, 7}, {8, 6},  {8, 7}};

inline int prefix_value(Bits& br, int prefix) {
  if (prefix < 4) return prefix + 1;
  const int extra = (prefix - 2) >> 1;
  const int offset = (2 + (prefix & 1)) << extra;
  return offset + static_cast<int>(br.read(extra)) + 1;
}

// ---- one entropy-coded image (RFC 9649 section 5): the ARGB image itself (is_main: may carry meta prefix codes) or a
// sub-image of a transform / the entropy image / the colour table ----------------------------------------------------
int read_image(Bits& br, int w, int h, bool is_main, std::vector<uint32_t>* out, int depth = 0) {
  if (depth > 2) return fail(IST_E_DECODE, "WebP: nested sub-images");
  int cache_bits = 0;
  if (br.read(1)) {
    cache_bits = static_cast<int>(br.read(4));
    if (cache_bits < 1 || cache_bits > 11) return fail(IST_E_DECODE, "WebP: bad colour cache size");
  }
  int meta_bits = 0, meta_w = 0;
  std::vector<uint32_t> meta;
  int n_groups = 1;
  if (is_main && br.read(1)) {
    meta_bits = static_cast<int>(br.read(3)) + 2;
    meta_w = (w + (1 << meta_bits) - 1) >> meta_bits;
    const int meta_h = (h + (1 << meta_bits) - 1) >> meta_bits;
    const int rc = read_image(br, meta_w, meta_h, false, &meta, depth + 1);
    if (rc) return rc;
    for (uint32_t& m : meta) { m = (m >> 8) & 0xFFFF; n_groups = std::max(n_groups, static_cast<int>(m) + 1); }
    if (n_groups > 65536) return fail(IST_E_DECODE, "WebP: too many prefix code groups");
  }
  const int cache_size = cache_bits ? 1 << cache_bits : 0;
  std::vector<Group> groups(static_cast<size_t>(n_groups));
  const int alphabet[5] = {256 + 24 + cache_size, 256, 256, 256, 40};
  for (Group& g : groups)
    for (int k = 0; k < 5; ++k) { const int rc = read_code(br, alphabet[k], &g.c[k]); if (rc) return rc; }
  std::vector<uint32_t> cache(static_cast<size_t>(cache_size), 0);
  const size_t total = static_cast<size_t>(w) * h;
  out->assign(total, 0);
  uint32_t* px = out->data();
  size_t pos = 0, cached = 0;
  int x = 0, y = 0;
  const Group* g = &groups[0];
  auto insert_upto = [&](size_t upto) {
    if (!cache_size) { cached = upto; return; }
    for (; cached < upto; ++cached) cache[(0x1e35a7bdu * px[cached]) >> (32 - cache_bits)] = px[cached];
  };
  while (pos < total) {
    if (meta_bits && (x & ((1 << meta_bits) - 1)) == 0) g = &groups[meta[static_cast<size_t>(y >> meta_bits) * meta_w + (x >> meta_bits)]];
    const int s = g->c[0].decode(br);
    if (s < 0 || br.eos) return fail(IST_E_DECODE, "WebP: corrupt pixel data");
    if (s < 256) {
      const int r = g->c[1].decode(br), b = g->c[2].decode(br), a = g->c[3].decode(br);
      if ((r | b | a) < 0) return fail(IST_E_DECODE, "WebP: corrupt pixel data");
      px[pos++] = (static_cast<uint32_t>(a) << 24) | (static_cast<uint32_t>(r) << 16) | (static_cast<uint32_t>(s) << 8) | static_cast<uint32_t>(b);
      if (++x >= w) { x = 0; ++y; }
    } else if (s < 256 + 24) {
      const int length = prefix_value(br, s - 256);
      const int ds = g->c[4].decode(br);
      if (ds < 0) return fail(IST_E_DECODE, "WebP: corrupt pixel data");
      const int dcode = prefix_value(br, ds);
      int64_t dist;
      if (dcode > 120) dist = dcode - 120;
      else { dist = kDistMap[dcode - 1][0] + static_cast<int64_t>(kDistMap[dcode - 1][1]) * w; if (dist < 1) dist = 1; }
      if (static_cast<size_t>(dist) > pos || pos + static_cast<size_t>(length) > total || br.eos) return fail(IST_E_DECODE, "WebP: back-reference out of range");
      for (int k = 0; k < length; ++k) { px[pos] = px[pos - static_cast<size_t>(dist)]; ++pos; }
      x += length;
      while (x >= w) { x -= w; ++y; }
      if (meta_bits && pos < total) g = &groups[meta[static_cast<size_t>(y >> meta_bits) * meta_w + (x >> meta_bits)]];
    } else {
      const int idx = s - (256 + 24);
      if (idx >= cache_size) return fail(IST_E_DECODE, "WebP: colour cache index out of range");
      insert_upto(pos);
      px[pos++] = cache[static_cast<size_t>(idx)];
      if (++x >= w) { x = 0; ++y; }
    }
    if (cache_size && (s < 256 || s >= 256 + 24)) insert_upto(pos);     // (copies are inserted lazily, before the next cache use)
  }
  if (br.eos) return fail(IST_E_DECODE, "WebP: truncated image data");
  return IST_OK;
}

// ---- inverse transforms (RFC 9649 section 4) -----------------------------------------------------------------------
inline uint32_t add_px(uint32_t a, uint32_t b) {          // per-channel add mod 256
  const uint32_t ag = (a & 0xFF00FF00u) + (b & 0xFF00FF00u), rb = (a & 0x00FF00FFu) + (b & 0x00FF00FFu);
  return (ag & 0xFF00FF00u) | (rb & 0x00FF00FFu);
}
inline uint32_t avg2(uint32_t a, uint32_t b) { return (((a ^ b) & 0xFEFEFEFEu) >> 1) + (a & b); }
inline int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
inline uint32_t select_px(uint32_t L, uint32_t T, uint32_t TL) {
  int pL = 0, pT = 0;
  for (int s = 0; s < 32; s += 8) {
    const int l = (L >> s) & 255, t = (T >> s) & 255, tl = (TL >> s) & 255;
    const int p = l + t - tl;
    pL += std::abs(p - l); pT += std::abs(p - t);
  }
  return pL < pT ? L : T;
}
inline uint32_t clamp_add_sub_full(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r = 0;
  for (int s = 0; s < 32; s += 8) r |= static_cast<uint32_t>(clamp255(int((a >> s) & 255) + int((b >> s) & 255) - int((c >> s) & 255))) << s;
  return r;
}
inline uint32_t clamp_add_sub_half(uint32_t a, uint32_t b) {
  uint32_t r = 0;
  for (int s = 0; s < 32; s += 8) { const int x = (a >> s) & 255, y = (b >> s) & 255; r |= static_cast<uint32_t>(clamp255(x + (x - y) / 2)) << s; }
  return r;
}

struct Transform { int type = 0, bits = 0, w = 0; std::vector<uint32_t> data; };

void inverse_predictor(const Transform& t, int w, int h, uint32_t* px) {
  // first row: L; first column: T; the top-left pixel: opaque black
  px[0] = add_px(px[0], 0xFF000000u);
  for (int x = 1; x < w; ++x) px[x] = add_px(px[x], px[x - 1]);
  const int bw = (w + (1 << t.bits) - 1) >> t.bits;
  for (int y = 1; y < h; ++y) {
    uint32_t* row = px + static_cast<size_t>(y) * w;
    const uint32_t* up = row - w;
    row[0] = add_px(row[0], up[0]);
    const uint32_t* modes = t.data.data() + static_cast<size_t>(y >> t.bits) * bw;
    for (int x = 1; x < w; ++x) {
      const int mode = (modes[x >> t.bits] >> 8) & 15;
      const uint32_t L = row[x - 1], T = up[x], TL = up[x - 1];
      const uint32_t TR = (x + 1 < w) ? up[x + 1] : row[0];       // past the right edge: the leftmost pixel of the current row
      uint32_t p;
      switch (mode) {
        case 0: p = 0xFF000000u; break;
        case 1: p = L; break;
        case 2: p = T; break;
        case 3: p = TR; break;
        case 4: p = TL; break;
        case 5: p = avg2(avg2(L, TR), T); break;
        case 6: p = avg2(L, TL); break;
        case 7: p = avg2(L, T); break;
        case 8: p = avg2(TL, T); break;
        case 9: p = avg2(T, TR); break;
        case 10: p = avg2(avg2(L, TL), avg2(T, TR)); break;
        case 11: p = select_px(L, T, TL); break;
        case 12: p = clamp_add_sub_full(L, T, TL); break;
        case 13: p = clamp_add_sub_half(avg2(L, T), TL); break;
        default: p = 0xFF000000u; break;                           // 14, 15: as mode 0 (libwebp maps them to black)
      }
      row[x] = add_px(row[x], p);
    }
  }
}

inline int ctd(int8_t t, int8_t c) { return (static_cast<int>(t) * static_cast<int>(c)) >> 5; }
void inverse_cross_color(const Transform& t, int w, int h, uint32_t* px) {
  const int bw = (w + (1 << t.bits) - 1) >> t.bits;
  for (int y = 0; y < h; ++y) {
    uint32_t* row = px + static_cast<size_t>(y) * w;
    const uint32_t* el = t.data.data() + static_cast<size_t>(y >> t.bits) * bw;
    for (int x = 0; x < w; ++x) {
      const uint32_t e = el[x >> t.bits];
      const int8_t g2r = static_cast<int8_t>(e & 255), g2b = static_cast<int8_t>((e >> 8) & 255), r2b = static_cast<int8_t>((e >> 16) & 255);
      const uint32_t v = row[x];
      const int8_t green = static_cast<int8_t>((v >> 8) & 255);
      int red = (v >> 16) & 255, blue = v & 255;
      red = (red + ctd(g2r, green)) & 255;
      blue = (blue + ctd(g2b, green)) & 255;
      blue = (blue + ctd(r2b, static_cast<int8_t>(red))) & 255;
      row[x] = (v & 0xFF00FF00u) | (static_cast<uint32_t>(red) << 16) | static_cast<uint32_t>(blue);
    }
  }
}

void inverse_subtract_green(size_t n, uint32_t* px) {
  for (size_t i = 0; i < n; ++i) {
    const uint32_t v = px[i], g = (v >> 8) & 255;
    px[i] = (v & 0xFF00FF00u) | ((((v >> 16) & 255) + g) & 255) << 16 | (((v & 255) + g) & 255);
  }
}

// decodes the VP8L payload (after the chunk header) into ARGB words, row-major, w*h
int vp8l_decode(const uint8_t* d, size_t n, int* out_w, int* out_h, std::vector<uint32_t>* out, bool header_only) {
  if (n < 5 || d[0] != 0x2F) return fail(IST_E_DECODE, "WebP: bad lossless signature");
  Bits br(d + 1, n - 1);
  const int w = static_cast<int>(br.read(14)) + 1, h = static_cast<int>(br.read(14)) + 1;
  br.read(1);                                                // alpha_is_used: a hint only
  if (br.read(3) != 0) return fail(IST_E_DECODE, "WebP: unknown lossless version");
  *out_w = w; *out_h = h;
  if (header_only) return IST_OK;
  std::vector<Transform> tr;
  int cur_w = w;
  bool seen[4] = {false, false, false, false};
  while (br.read(1)) {
    Transform t;
    t.type = static_cast<int>(br.read(2));
    if (seen[t.type]) return fail(IST_E_DECODE, "WebP: a transform is used twice");
    seen[t.type] = true;
    t.w = cur_w;
    if (t.type == 0 || t.type == 1) {
      t.bits = static_cast<int>(br.read(3)) + 2;
      const int bw = (cur_w + (1 << t.bits) - 1) >> t.bits, bh = (h + (1 << t.bits) - 1) >> t.bits;
      const int rc = read_image(br, bw, bh, false, &t.data);
      if (rc) return rc;
    } else if (t.type == 3) {
      const int size = static_cast<int>(br.read(8)) + 1;
      const int rc = read_image(br, size, 1, false, &t.data);
      if (rc) return rc;
      for (int i = 1; i < size; ++i) t.data[static_cast<size_t>(i)] = add_px(t.data[static_cast<size_t>(i)], t.data[static_cast<size_t>(i - 1)]);
      t.bits = size <= 2 ? 3 : size <= 4 ? 2 : size <= 16 ? 1 : 0;
      t.data.resize(256, 0);                                 // indices past the table read transparent black
      cur_w = (cur_w + (1 << t.bits) - 1) >> t.bits;
    }
    tr.push_back(std::move(t));
    if (br.eos) return fail(IST_E_DECODE, "WebP: truncated transform data");
  }
  std::vector<uint32_t> px;
  int rc = read_image(br, cur_w, h, true, &px);
  if (rc) return rc;
  for (size_t k = tr.size(); k-- > 0;) {
    const Transform& t = tr[k];
    if (t.type == 0) inverse_predictor(t, t.w, h, px.data());
    else if (t.type == 1) inverse_cross_color(t, t.w, h, px.data());
    else if (t.type == 2) inverse_subtract_green(px.size(), px.data());
    else {                                                   // colour indexing: unpack 8 >> bits indices per green byte
      const int full_w = t.w, packed_w = (full_w + (1 << t.bits) - 1) >> t.bits;
      std::vector<uint32_t> wide(static_cast<size_t>(full_w) * h);
      const int per = 1 << t.bits, bits_per = 8 >> t.bits, mask = (1 << bits_per) - 1;
      for (int y = 0; y < h; ++y) {
        const uint32_t* s = px.data() + static_cast<size_t>(y) * packed_w;
        uint32_t* o = wide.data() + static_cast<size_t>(y) * full_w;
        for (int x = 0; x < full_w; ++x) {
          const uint32_t g = (s[x / per] >> 8) & 255;
          o[x] = t.data[(g >> ((x % per) * bits_per)) & static_cast<uint32_t>(mask)];
        }
      }
      px.swap(wide);
    }
  }
  if (px.size() != static_cast<size_t>(w) * h) return fail(IST_E_DECODE, "WebP: transform sizes do not add up");
  out->swap(px);
  return IST_OK;
}

// EXIF orientation (tag 0x0112) from a raw TIFF block (the payload of the container's EXIF chunk)
int tiff_orientation(const uint8_t* t, size_t tn) {
  if (tn >= 6 && !std::memcmp(t, "Exif\0\0", 6)) { t += 6; tn -= 6; }
  if (tn < 8) return 0;
  const bool le = t[0] == 'I' && t[1] == 'I';
  if (!le && !(t[0] == 'M' && t[1] == 'M')) return 0;
  auto r16 = [&](size_t o) -> uint32_t { return le ? (t[o] | (t[o + 1] << 8)) : ((t[o] << 8) | t[o + 1]); };
  auto r32 = [&](size_t o) -> uint32_t { return le ? (t[o] | (t[o + 1] << 8) | (t[o + 2] << 16) | (uint32_t(t[o + 3]) << 24))
                                                  : ((uint32_t(t[o]) << 24) | (t[o + 1] << 16) | (t[o + 2] << 8) | t[o + 3]); };
  if (r16(2) != 42) return 0;
  const size_t ifd = r32(4);
  if (ifd + 2 > tn) return 0;
  const uint32_t cnt = r16(ifd);
  for (uint32_t i = 0; i < cnt; ++i) {
    const size_t e = ifd + 2 + 12 * static_cast<size_t>(i);
    if (e + 12 > tn) break;
    if (r16(e) == 0x0112) { const uint32_t v = r16(e + 8); return (v >= 1 && v <= 8) ? static_cast<int>(v) : 0; }
  }
  return 0;
}

struct Riff {
  const uint8_t* vp8l = nullptr; size_t vp8l_n = 0;
  const uint8_t* vp8 = nullptr; size_t vp8_n = 0;
  const uint8_t* alph = nullptr; size_t alph_n = 0;
  int canvas_w = 0, canvas_h = 0, orientation = 0;
  bool animated = false;
  // an animated file shows its FIRST frame as a still image (what Image.src gives the page, utils/canvas.js:27-121): the
  // frame's own bitstream chunks are picked up above; it sits at (fx, fy) of the canvas, the rest of which is transparent
  bool frame = false; int fx = 0, fy = 0, fw = 0, fh = 0;
};

int parse_riff(const uint8_t* f, int64_t n, Riff* R) {
  if (!f || n < 20 || std::memcmp(f, "RIFF", 4) != 0 || std::memcmp(f + 8, "WEBP", 4) != 0) return fail(IST_E_DECODE, "not a WebP file");
  int64_t end = 8 + static_cast<int64_t>(le32(f + 4));
  if (end > n) end = n;                                      // a short file is judged by the chunks it does hold
  int64_t pos = 12;
  while (pos + 8 <= end) {
    const uint8_t* tag = f + pos;
    const int64_t len = le32(f + pos + 4);
    const uint8_t* d = f + pos + 8;
    if (pos + 8 + len > end) return fail(IST_E_DECODE, "truncated WebP chunk");
    if (!std::memcmp(tag, "VP8L", 4)) { if (!R->vp8l && !R->vp8) { R->vp8l = d; R->vp8l_n = static_cast<size_t>(len); } }
    else if (!std::memcmp(tag, "VP8 ", 4)) { if (!R->vp8l && !R->vp8) { R->vp8 = d; R->vp8_n = static_cast<size_t>(len); } }
    else if (!std::memcmp(tag, "ALPH", 4)) { if (!R->alph) { R->alph = d; R->alph_n = static_cast<size_t>(len); } }
    else if (!std::memcmp(tag, "VP8X", 4)) {
      if (len < 10) return fail(IST_E_DECODE, "bad WebP VP8X chunk");
      R->animated = (d[0] & 0x02) != 0;
      R->canvas_w = static_cast<int>(le24(d + 4)) + 1; R->canvas_h = static_cast<int>(le24(d + 7)) + 1;
    } else if (!std::memcmp(tag, "ANIM", 4)) R->animated = true;
    else if (!std::memcmp(tag, "ANMF", 4)) {
      R->animated = true;
      if (!R->frame) {                                        // the first frame: 16 header bytes, then its ALPH / VP8 / VP8L chunks
        if (len < 16) return fail(IST_E_DECODE, "bad WebP ANMF chunk");
        R->frame = true;
        R->fx = 2 * static_cast<int>(le24(d)); R->fy = 2 * static_cast<int>(le24(d + 3));
        R->fw = static_cast<int>(le24(d + 6)) + 1; R->fh = static_cast<int>(le24(d + 9)) + 1;
        R->vp8 = R->vp8l = R->alph = nullptr;
        int64_t q = 16;
        while (q + 8 <= len) {
          const uint8_t* t2 = d + q;
          const int64_t l2 = le32(d + q + 4);
          if (q + 8 + l2 > len) return fail(IST_E_DECODE, "truncated WebP frame chunk");
          if (!std::memcmp(t2, "VP8L", 4)) { if (!R->vp8l && !R->vp8) { R->vp8l = t2 + 8; R->vp8l_n = static_cast<size_t>(l2); } }
          else if (!std::memcmp(t2, "VP8 ", 4)) { if (!R->vp8l && !R->vp8) { R->vp8 = t2 + 8; R->vp8_n = static_cast<size_t>(l2); } }
          else if (!std::memcmp(t2, "ALPH", 4)) { if (!R->alph) { R->alph = t2 + 8; R->alph_n = static_cast<size_t>(l2); } }
          q += 8 + l2 + (l2 & 1);
        }
      }
    }
    else if (!std::memcmp(tag, "EXIF", 4)) R->orientation = tiff_orientation(d, static_cast<size_t>(len));
    pos += 8 + len + (len & 1);
  }
  if (R->animated && !R->frame) return fail(IST_E_DECODE, "animated WebP without a frame");
  if (R->frame && (R->canvas_w < 1 || R->canvas_h < 1 || R->fx + R->fw > R->canvas_w || R->fy + R->fh > R->canvas_h))
    return fail(IST_E_DECODE, "WebP frame outside its canvas");
  if (!R->vp8l && !R->vp8) return fail(IST_E_DECODE, "WebP without image data");
  return IST_OK;
}

}  // namespace

bool is_webp(const uint8_t* f, int64_t n) { return f && n >= 12 && !std::memcmp(f, "RIFF", 4) && !std::memcmp(f + 8, "WEBP", 4); }

int webp_info(const uint8_t* f, int64_t n, int32_t* w, int32_t* h, int32_t* orientation) {
  Riff R;
  int rc = parse_riff(f, n, &R);
  if (rc) return rc;
  int iw = 0, ih = 0;
  if (R.vp8l) {
    std::vector<uint32_t> none;
    rc = vp8l_decode(R.vp8l, R.vp8l_n, &iw, &ih, &none, true);
  } else rc = vp8_info(R.vp8, R.vp8_n, &iw, &ih);
  if (rc) return rc;
  if (R.frame) {
    if (iw != R.fw || ih != R.fh) return fail(IST_E_DECODE, "WebP frame header and bitstream sizes differ");
    iw = R.canvas_w; ih = R.canvas_h;
  } else if (R.canvas_w && (R.canvas_w != iw || R.canvas_h != ih)) return fail(IST_E_DECODE, "WebP canvas and frame sizes differ");
  if (w) *w = iw;
  if (h) *h = ih;
  if (orientation) *orientation = R.orientation;
  return IST_OK;
}

static int webp_decode_inner(const uint8_t* f, int64_t n, uint8_t* out, size_t pitch, int64_t out_rows) {
  Riff R;
  int rc = parse_riff(f, n, &R);
  if (rc) return rc;
  if (R.frame) {                                            // first frame of an animation, on its transparent canvas
    if (!out || pitch < static_cast<size_t>(R.canvas_w) * 4 || out_rows < R.canvas_h) return fail(IST_E_INVALID, "output buffer too small");
    for (int y = 0; y < R.canvas_h; ++y) std::memset(out + static_cast<size_t>(y) * pitch, 0, static_cast<size_t>(R.canvas_w) * 4);
    out += static_cast<size_t>(R.fy) * pitch + static_cast<size_t>(R.fx) * 4;
    out_rows = R.fh;
    R.canvas_w = R.fw; R.canvas_h = R.fh;                   // below: the frame is the image
  }
  int w = 0, h = 0;
  if (R.vp8l) {
    std::vector<uint32_t> px;
    rc = vp8l_decode(R.vp8l, R.vp8l_n, &w, &h, &px, true);
    if (rc) return rc;
    if (!out || pitch < static_cast<size_t>(w) * 4 || out_rows < h) return fail(IST_E_INVALID, "output buffer too small");
    if (R.canvas_w && (R.canvas_w != w || R.canvas_h != h)) return fail(IST_E_DECODE, "WebP canvas and frame sizes differ");
    rc = vp8l_decode(R.vp8l, R.vp8l_n, &w, &h, &px, false);
    if (rc) return rc;
    for (int y = 0; y < h; ++y) {
      const uint32_t* s = px.data() + static_cast<size_t>(y) * w;
      uint8_t* o = out + static_cast<size_t>(y) * pitch;
      for (int x = 0; x < w; ++x, o += 4) { const uint32_t v = s[x]; o[0] = (v >> 16) & 255; o[1] = (v >> 8) & 255; o[2] = v & 255; o[3] = v >> 24; }
    }
    return IST_OK;
  }
  rc = vp8_info(R.vp8, R.vp8_n, &w, &h);
  if (rc) return rc;
  if (!out || pitch < static_cast<size_t>(w) * 4 || out_rows < h) return fail(IST_E_INVALID, "output buffer too small");
  if (R.canvas_w && (R.canvas_w != w || R.canvas_h != h)) return fail(IST_E_DECODE, "WebP canvas and frame sizes differ");
  rc = vp8_decode_rgba8(R.vp8, R.vp8_n, out, pitch);
  if (rc) return rc;
  if (R.alph) {                                              // lossy + alpha: the ALPH chunk carries the alpha plane
    std::vector<uint8_t> alpha;
    rc = webp_alpha_plane(R.alph, R.alph_n, w, h, &alpha);
    if (rc) return rc;
    for (int y = 0; y < h; ++y) { uint8_t* o = out + static_cast<size_t>(y) * pitch; for (int x = 0; x < w; ++x) o[4 * x + 3] = alpha[static_cast<size_t>(y) * w + x]; }
  }
  return IST_OK;
}

int webp_decode_rgba8(const uint8_t* f, int64_t n, uint8_t* out, size_t pitch, int64_t out_rows) {
  try { return webp_decode_inner(f, n, out, pitch, out_rows); }
  catch (const std::bad_alloc&) { return fail(IST_E_NOMEM, "out of memory while decoding the WebP"); }
}

// ALPH chunk (container spec): header byte = rsrv:2 | preprocessing:2 | filtering:2 | compression:2; the plane is raw or a
// VP8L image whose green channel carries alpha; then horizontal / vertical / gradient un-filtering.
int webp_alpha_plane(const uint8_t* d, size_t n, int w, int h, std::vector<uint8_t>* out) {
  if (n < 1) return fail(IST_E_DECODE, "WebP: empty alpha chunk");
  const int compression = d[0] & 3, filtering = (d[0] >> 2) & 3;
  const size_t total = static_cast<size_t>(w) * h;
  out->assign(total, 255);
  if (compression == 0) {
    if (n - 1 < total) return fail(IST_E_DECODE, "WebP: truncated alpha plane");
    std::memcpy(out->data(), d + 1, total);
  } else if (compression == 1) {
    // a headerless VP8L stream: the size comes from the frame
    std::vector<uint8_t> fake(5 + (n - 1));
    const uint32_t hdr = static_cast<uint32_t>(w - 1) | (static_cast<uint32_t>(h - 1) << 14);      // alpha_is_used 0, version 0
    fake[0] = 0x2F; fake[1] = hdr & 255; fake[2] = (hdr >> 8) & 255; fake[3] = (hdr >> 16) & 255; fake[4] = (hdr >> 24) & 255;
    std::memcpy(fake.data() + 5, d + 1, n - 1);
    std::vector<uint32_t> px; int aw = 0, ah = 0;
    const int rc = vp8l_decode(fake.data(), fake.size(), &aw, &ah, &px, false);
    if (rc) return rc;
    for (size_t i = 0; i < total; ++i) (*out)[i] = static_cast<uint8_t>((px[i] >> 8) & 255);
  } else return fail(IST_E_DECODE, "WebP: unknown alpha compression");
  uint8_t* a = out->data();
  if (filtering == 1) {                                      // horizontal
    for (int y = 0; y < h; ++y) { uint8_t* r = a + static_cast<size_t>(y) * w; if (y) r[0] = static_cast<uint8_t>(r[0] + r[-w]); for (int x = 1; x < w; ++x) r[x] = static_cast<uint8_t>(r[x] + r[x - 1]); }
  } else if (filtering == 2) {                               // vertical
    for (int x = 1; x < w; ++x) a[x] = static_cast<uint8_t>(a[x] + a[x - 1]);
    for (int y = 1; y < h; ++y) { uint8_t* r = a + static_cast<size_t>(y) * w; for (int x = 0; x < w; ++x) r[x] = static_cast<uint8_t>(r[x] + r[x - w]); }
  } else if (filtering == 3) {                               // gradient: clip(L + T - TL)
    for (int x = 1; x < w; ++x) a[x] = static_cast<uint8_t>(a[x] + a[x - 1]);
    for (int y = 1; y < h; ++y) {
      uint8_t* r = a + static_cast<size_t>(y) * w;
      r[0] = static_cast<uint8_t>(r[0] + r[-w]);
      for (int x = 1; x < w; ++x) r[x] = static_cast<uint8_t>(r[x] + clamp255(int(r[x - 1]) + int(r[x - w]) - int(r[x - w - 1])));
    }
  }
  return IST_OK;
}

}  // namespace ist
