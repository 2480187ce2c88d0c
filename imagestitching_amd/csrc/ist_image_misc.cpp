// ist_image_misc.cpp — the two remaining easy members of SUPPORTED_IMAGE_TYPES (pages/index/index.js:4): BMP and GIF
// (first frame), decoded on the host into RGBA8.  Reference anchor: loadImageFrom (utils/canvas.js:27-121) — the platform
// decoder behind Image.src.  Both are lossless formats, so the result is pinned by any conforming decoder (tests: PIL).
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ist_internal.h"

using namespace ist;

namespace {

inline uint32_t le16(const uint8_t* p) { return p[0] | (p[1] << 8); }
inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (uint32_t(p[3]) << 24); }

// ------------------------------------------------------------------------------------------------ BMP
struct Bmp { int w = 0, h = 0, bpp = 0; bool top_down = false; uint32_t comp = 0, off = 0, ncol = 0, dib = 0; uint32_t mask[4] = {0, 0, 0, 0}; int shift[4] = {0, 0, 0, 0}, bits[4] = {0, 0, 0, 0}; };

int bmp_header(const uint8_t* f, int64_t n, Bmp* B) {
  if (n < 26 || f[0] != 'B' || f[1] != 'M') return fail(IST_E_DECODE, "not a BMP file");
  B->off = le32(f + 10); B->dib = le32(f + 14);
  if (B->dib < 40 || 14 + int64_t(B->dib) > n) return fail(IST_E_UNSUPPORTED, "unsupported BMP header");
  const int32_t w = int32_t(le32(f + 18)), h = int32_t(le32(f + 22));
  if (h == INT32_MIN) return fail(IST_E_DECODE, "bad BMP size");
  B->w = w; B->h = h < 0 ? -h : h; B->top_down = h < 0;
  B->bpp = int(le16(f + 28)); B->comp = le32(f + 30); B->ncol = le32(f + 46);
  if (B->w < 1 || B->h < 1 || B->w > (1 << 29)) return fail(IST_E_DECODE, "bad BMP size");
  if (!(B->bpp == 1 || B->bpp == 4 || B->bpp == 8 || B->bpp == 16 || B->bpp == 24 || B->bpp == 32)) return fail(IST_E_UNSUPPORTED, "unsupported BMP bit depth");
  // BI_RGB, BI_BITFIELDS, and the two run-length forms of palette images (BI_RLE8 on 8 bit, BI_RLE4 on 4 bit; always bottom-up)
  const bool rle = (B->comp == 1 && B->bpp == 8) || (B->comp == 2 && B->bpp == 4);
  if (B->comp != 0 && B->comp != 3 && !rle) return fail(IST_E_UNSUPPORTED, "this BMP compression (JPEG / PNG payload, or RLE on the wrong bit depth) is not supported");
  if (rle && B->top_down) return fail(IST_E_DECODE, "run-length BMP cannot be top-down");
  if (B->comp == 3) {
    const uint8_t* m = f + 14 + 40;            // masks follow a 40-byte header, or live inside V4/V5 headers at the same offset
    if (14 + 40 + 12 > n) return fail(IST_E_DECODE, "truncated BMP masks");
    B->mask[0] = le32(m); B->mask[1] = le32(m + 4); B->mask[2] = le32(m + 8);
    B->mask[3] = (B->dib >= 56 && 14 + 40 + 16 <= n) ? le32(m + 12) : 0;
  } else if (B->bpp == 16) { B->mask[0] = 0x7C00; B->mask[1] = 0x03E0; B->mask[2] = 0x001F; }
  // position and width of every bit field, once per file: a mask is one contiguous run of ones (BI_BITFIELDS), or empty
  for (int c = 0; c < 4; ++c) {
    const uint32_t m = B->mask[c];
    if (!m) continue;
    B->shift[c] = __builtin_ctz(m);
    B->bits[c] = __builtin_popcount(m);
    const uint64_t run = ((uint64_t(1) << B->bits[c]) - 1) << B->shift[c];
    if (run != m) return fail(IST_E_DECODE, "BMP bit-field mask is not contiguous");
  }
  return IST_OK;
}

inline uint8_t field(uint32_t v, uint32_t mask, int shift, int bits) {
  if (!mask) return 0;
  const uint32_t x = (v & mask) >> shift;
  return bits >= 8 ? uint8_t(x >> (bits - 8)) : uint8_t((x * 255 + ((1u << bits) - 1) / 2) / ((1u << bits) - 1));
}

// BI_RLE8 / BI_RLE4: (count, value) pairs; count 0 escapes: 0 end of line, 1 end of bitmap, 2 delta (dx, dy), n >= 3 a literal
// run of n pixels padded to 16 bits.  Pixels the stream never sets keep palette entry 0; runs are clipped at the row end.
int bmp_decode_rle(const uint8_t* f, int64_t n, const Bmp& B, uint8_t* out, size_t pitch) {
  if (B.off > n) return fail(IST_E_DECODE, "truncated BMP pixel data");
  const uint8_t* pal = f + 14 + B.dib;
  const uint32_t ncol = B.ncol ? B.ncol : (1u << B.bpp);
  if (ncol > 256 || pal + 4 * size_t(ncol) > f + n) return fail(IST_E_DECODE, "truncated BMP palette");
  if (int64_t(B.w) * B.h > (int64_t(1) << 31)) return fail(IST_E_UNSUPPORTED, "run-length BMP too large");
  std::vector<uint8_t> idx;
  try { idx.assign(size_t(B.w) * B.h, 0); } catch (const std::bad_alloc&) { return fail(IST_E_NOMEM, "out of memory for a BMP"); }
  const uint8_t* p = f + B.off; const uint8_t* e = f + n;
  int64_t x = 0, y = 0;                              // y counts rows from the BOTTOM
  auto put = [&](uint32_t v) { if (x < B.w && y < B.h) idx[size_t(B.h - 1 - y) * B.w + size_t(x)] = uint8_t(v); ++x; };
  bool done = false;
  while (!done && p + 2 <= e) {
    const int c = p[0], v = p[1]; p += 2;
    if (c > 0) {
      if (B.bpp == 8) for (int k = 0; k < c; ++k) put(uint32_t(v));
      else for (int k = 0; k < c; ++k) put(uint32_t((k & 1) ? (v & 15) : (v >> 4)));
    } else if (v == 0) { x = 0; ++y; }
    else if (v == 1) done = true;
    else if (v == 2) { if (p + 2 > e) return fail(IST_E_DECODE, "truncated run-length BMP"); x += p[0]; y += p[1]; p += 2; }
    else {
      const int64_t bytes = B.bpp == 8 ? v : (v + 1) / 2;
      if (p + ((bytes + 1) & ~int64_t(1)) > e) return fail(IST_E_DECODE, "truncated run-length BMP");
      for (int k = 0; k < v; ++k) put(B.bpp == 8 ? uint32_t(p[k]) : uint32_t((k & 1) ? (p[k / 2] & 15) : (p[k / 2] >> 4)));
      p += (bytes + 1) & ~int64_t(1);
    }
    if (y >= B.h && !(c == 0 && v == 1)) { /* rows above the top are dropped; keep parsing until end of bitmap or data */ }
  }
  for (int yy = 0; yy < B.h; ++yy) {
    uint8_t* o = out + size_t(yy) * pitch;
    for (int xx = 0; xx < B.w; ++xx, o += 4) {
      const uint32_t i = idx[size_t(yy) * B.w + size_t(xx)];
      if (i >= ncol) return fail(IST_E_DECODE, "BMP palette index out of range");
      o[0] = pal[4 * i + 2]; o[1] = pal[4 * i + 1]; o[2] = pal[4 * i]; o[3] = 255;
    }
  }
  return IST_OK;
}

int bmp_decode(const uint8_t* f, int64_t n, uint8_t* out, size_t pitch) {
  Bmp B;
  int rc = bmp_header(f, n, &B);
  if (rc) return rc;
  if (B.comp == 1 || B.comp == 2) return bmp_decode_rle(f, n, B, out, pitch);
  const size_t stride = ((size_t(B.w) * B.bpp + 31) / 32) * 4;
  if (int64_t(B.off) + int64_t(stride) * B.h > n) return fail(IST_E_DECODE, "truncated BMP pixel data");
  const uint8_t* pal = f + 14 + B.dib + (B.comp == 3 && B.dib == 40 ? 12 : 0);
  const uint32_t ncol = B.bpp <= 8 ? (B.ncol ? B.ncol : (1u << B.bpp)) : 0;
  if (B.bpp <= 8 && pal + 4 * size_t(ncol) > f + n) return fail(IST_E_DECODE, "truncated BMP palette");
  auto F = [&](uint32_t v, int c) { return field(v, B.mask[c], B.shift[c], B.bits[c]); };
  for (int y = 0; y < B.h; ++y) {
    const uint8_t* s = f + B.off + stride * size_t(B.top_down ? y : B.h - 1 - y);
    uint8_t* o = out + size_t(y) * pitch;
    for (int x = 0; x < B.w; ++x, o += 4) {
      if (B.bpp == 24) { o[0] = s[3 * x + 2]; o[1] = s[3 * x + 1]; o[2] = s[3 * x]; o[3] = 255; }
      else if (B.bpp == 32) {
        const uint32_t v = le32(s + 4 * x);
        if (B.comp == 3) { o[0] = F(v, 0); o[1] = F(v, 1); o[2] = F(v, 2); o[3] = B.mask[3] ? F(v, 3) : 255; }
        else { o[0] = s[4 * x + 2]; o[1] = s[4 * x + 1]; o[2] = s[4 * x]; o[3] = 255; }      // BI_RGB: the 4th byte is padding
      } else if (B.bpp == 16) {
        const uint32_t v = le16(s + 2 * x);
        o[0] = F(v, 0); o[1] = F(v, 1); o[2] = F(v, 2); o[3] = B.mask[3] ? F(v, 3) : 255;
      } else {
        const int per = 8 / B.bpp;
        const uint32_t idx = (s[x / per] >> ((per - 1 - x % per) * B.bpp)) & ((1u << B.bpp) - 1);
        if (idx >= ncol) return fail(IST_E_DECODE, "BMP palette index out of range");
        o[0] = pal[4 * idx + 2]; o[1] = pal[4 * idx + 1]; o[2] = pal[4 * idx]; o[3] = 255;
      }
    }
  }
  return IST_OK;
}

// ------------------------------------------------------------------------------------------------ GIF (first frame)
struct Gif { int w = 0, h = 0; };

int gif_header(const uint8_t* f, int64_t n, Gif* G) {
  if (n < 13 || std::memcmp(f, "GIF8", 4) != 0 || (f[4] != '7' && f[4] != '9') || f[5] != 'a') return fail(IST_E_DECODE, "not a GIF file");
  G->w = int(le16(f + 6)); G->h = int(le16(f + 8));
  if (G->w < 1 || G->h < 1) return fail(IST_E_DECODE, "bad GIF size");
  return IST_OK;
}

int gif_decode(const uint8_t* f, int64_t n, uint8_t* out, size_t pitch) {
  Gif G;
  int rc = gif_header(f, n, &G);
  if (rc) return rc;
  int64_t pos = 13;
  const uint8_t* gct = nullptr; int gct_n = 0;
  if (f[10] & 0x80) { gct_n = 1 << ((f[10] & 7) + 1); gct = f + pos; pos += 3 * gct_n; if (pos > n) return fail(IST_E_DECODE, "truncated GIF"); }
  for (int y = 0; y < G.h; ++y) std::memset(out + size_t(y) * pitch, 0, size_t(G.w) * 4);      // transparent canvas
  int transparent = -1;
  while (pos < n) {
    const uint8_t b = f[pos++];
    if (b == 0x3B) break;
    if (b == 0x21) {                                   // extension
      if (pos >= n) break;
      const uint8_t label = f[pos++];
      if (label == 0xF9 && pos + 6 <= n && f[pos] == 4) { if (f[pos + 1] & 1) transparent = f[pos + 4]; }
      while (pos < n && f[pos]) pos += 1 + f[pos];
      ++pos;
      continue;
    }
    if (b != 0x2C) return fail(IST_E_DECODE, "corrupt GIF block");
    if (pos + 9 > n) return fail(IST_E_DECODE, "truncated GIF image descriptor");
    const int ix = int(le16(f + pos)), iy = int(le16(f + pos + 2)), iw = int(le16(f + pos + 4)), ih = int(le16(f + pos + 6));
    const uint8_t fl = f[pos + 8];
    pos += 9;
    const uint8_t* ct = gct; int ct_n = gct_n;
    if (fl & 0x80) { ct_n = 1 << ((fl & 7) + 1); ct = f + pos; pos += 3 * ct_n; }
    if (!ct || pos >= n) return fail(IST_E_DECODE, "GIF without a colour table");
    const bool interlaced = (fl & 0x40) != 0;
    const int min_code = f[pos++];
    if (min_code < 2 || min_code > 8) return fail(IST_E_DECODE, "bad GIF LZW code size");
    std::vector<uint8_t> data;
    while (pos < n && f[pos]) { const int l = f[pos]; if (pos + 1 + l > n) return fail(IST_E_DECODE, "truncated GIF data"); data.insert(data.end(), f + pos + 1, f + pos + 1 + l); pos += 1 + l; }
    // LZW
    const int clear = 1 << min_code, eoi = clear + 1;
    std::vector<uint16_t> prefix(4096); std::vector<uint8_t> suffix(4096), stack(4097);
    // (the descriptor is untrusted: never reserve more than the logical screen can show)
    std::vector<uint8_t> idx; idx.reserve(std::min(size_t(iw) * ih, size_t(G.w) * G.h));
    int code_size = min_code + 1, next = eoi + 1, prev = -1;
    uint32_t acc = 0; int nbits = 0; size_t dp = 0;
    const size_t want = size_t(iw) * ih;
    while (idx.size() < want) {
      while (nbits < code_size && dp < data.size()) { acc |= uint32_t(data[dp++]) << nbits; nbits += 8; }
      if (nbits < code_size) break;
      int code = int(acc & ((1u << code_size) - 1)); acc >>= code_size; nbits -= code_size;
      if (code == clear) { code_size = min_code + 1; next = eoi + 1; prev = -1; continue; }
      if (code == eoi) break;
      if (prev < 0) { if (code >= clear) return fail(IST_E_DECODE, "corrupt GIF LZW stream"); idx.push_back(uint8_t(code)); prev = code; continue; }
      if (code > next) return fail(IST_E_DECODE, "corrupt GIF LZW stream");
      int sp = 0, cur = (code == next) ? prev : code;          // code == next: the string is string(prev) + its own first symbol
      while (cur >= clear) {
        if (cur >= 4096 || cur == clear || cur == eoi || sp >= 4096) return fail(IST_E_DECODE, "corrupt GIF LZW stream");
        stack[sp++] = suffix[cur]; cur = prefix[cur];
      }
      const uint8_t first = uint8_t(cur);
      stack[sp++] = first;
      for (int i = sp - 1; i >= 0; --i) idx.push_back(stack[i]);
      if (code == next) idx.push_back(first);
      if (next < 4096) { prefix[next] = uint16_t(prev); suffix[next] = first; ++next; if (next == (1 << code_size) && code_size < 12) ++code_size; }
      prev = code;
    }
    if (idx.size() > want) idx.resize(want);
    // place the frame (clipped to the logical screen)
    static const int start[4] = {0, 4, 2, 1}, step[4] = {8, 8, 4, 2};
    size_t k = 0;
    auto put_row = [&](int ry) {
      const int Y = iy + ry;
      for (int x = 0; x < iw; ++x, ++k) {
        if (k >= idx.size()) return;
        const int X = ix + x, v = idx[k];
        if (Y < 0 || Y >= G.h || X < 0 || X >= G.w || v == transparent || v >= ct_n) continue;
        uint8_t* o = out + size_t(Y) * pitch + size_t(X) * 4;
        o[0] = ct[3 * v]; o[1] = ct[3 * v + 1]; o[2] = ct[3 * v + 2]; o[3] = 255;
      }
    };
    if (!interlaced) for (int y = 0; y < ih; ++y) put_row(y);
    else for (int p = 0; p < 4; ++p) for (int y = start[p]; y < ih; y += step[p]) put_row(y);
    return IST_OK;                                      // first frame only (what a still Image shows)
  }
  return fail(IST_E_DECODE, "GIF without an image");
}

}  // namespace

extern "C" {

// returns IST_OK and fills w,h when `file` is a BMP or a GIF; IST_E_DECODE "unknown" otherwise
int ist_misc_info(const uint8_t* file, int64_t len, int32_t* w, int32_t* h) {
  if (file && len >= 2 && file[0] == 'B' && file[1] == 'M') { Bmp B; const int rc = bmp_header(file, len, &B); if (rc) return rc; if (w) *w = B.w; if (h) *h = B.h; return IST_OK; }
  if (file && len >= 4 && !std::memcmp(file, "GIF8", 4)) { Gif G; const int rc = gif_header(file, len, &G); if (rc) return rc; if (w) *w = G.w; if (h) *h = G.h; return IST_OK; }
  return fail(IST_E_DECODE, "unknown image format");
}

int ist_misc_decode_rgba8(const uint8_t* file, int64_t len, uint8_t* out, size_t pitch, int64_t out_rows) {
  int32_t w = 0, h = 0;
  const int rc = ist_misc_info(file, len, &w, &h);
  if (rc) return rc;
  if (!out || pitch < size_t(w) * 4 || out_rows < h) return fail(IST_E_INVALID, "output buffer too small");
  try { return file[0] == 'B' ? bmp_decode(file, len, out, pitch) : gif_decode(file, len, out, pitch); }
  catch (const std::bad_alloc&) { return fail(IST_E_NOMEM, "out of memory while decoding the image"); }
}

}  // extern "C"
