// ist_png_decode.cpp — PNG file -> RGBA8 (straight alpha), the decode step in front of the stitch path for PNG inputs
// (SURVEY.md section 8f rank 3).
//
// Reference anchor: loadImageFrom (utils/canvas.js:27-121) sets Image.src and the platform decodes the file into a
// bitmap; 'png' is one of SUPPORTED_IMAGE_TYPES (pages/index/index.js:4).  PNG is lossless, so the result is pinned:
// every conforming decoder yields the same RGBA bytes (tests compare with PIL, bit for bit).
//
// Host code: inflate is zlib's (libz is part of the image), un-filtering is the five PNG predictors.  Un-filtering is
// row- and pixel-serial by construction (Sub/Average/Paeth depend on the reconstructed left neighbour), so it stays on
// the host and the decoded rows are what gets uploaded.  JPEG / WebP / HEIC inputs need their own entropy decoders and
// are not built (there are no codec headers in the image): IST_E_UNSUPPORTED names the format.
// Supported: colour types 0, 2, 3, 4, 6; bit depths 1-16; tRNS; non-interlaced.  Adam7 -> IST_E_UNSUPPORTED.
#include <zlib.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "ist_internal.h"

using namespace ist;

namespace {

inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

struct Header { uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0; };

// walks the chunks; collects IDAT, PLTE, tRNS.  Returns 0 or an error code.
int parse(const uint8_t* f, int64_t n, Header* H, std::vector<uint8_t>* idat, std::vector<uint8_t>* plte, std::vector<uint8_t>* trns) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (!f || n < 8 + 25 + 12) return fail(IST_E_DECODE, "not a PNG file (too short)");
  if (std::memcmp(f, sig, 8) != 0) {
    if (f[0] == 0xFF && f[1] == 0xD8) return fail(IST_E_UNSUPPORTED, "this is a JPEG file: use ist_image_decode_rgba8");
    if (n >= 12 && !std::memcmp(f, "RIFF", 4) && !std::memcmp(f + 8, "WEBP", 4)) return fail(IST_E_UNSUPPORTED, "WebP decode is not built (PNG, JPEG, BMP and GIF inputs are)");
    return fail(IST_E_DECODE, "not a PNG file");
  }
  int64_t pos = 8;
  bool have_ihdr = false, end = false;
  while (pos + 12 <= n && !end) {
    const uint32_t len = be32(f + pos);
    const uint8_t* type = f + pos + 4;
    if (pos + 12 + int64_t(len) > n) return fail(IST_E_DECODE, "truncated PNG chunk");
    const uint8_t* data = f + pos + 8;
    const uint32_t crc = be32(data + len);
    if (uint32_t(crc32(crc32(0L, type, 4), data, len)) != crc) return fail(IST_E_DECODE, "PNG chunk CRC mismatch");
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) return fail(IST_E_DECODE, "bad IHDR");
      H->w = be32(data); H->h = be32(data + 4); H->depth = data[8]; H->ctype = data[9]; H->interlace = data[12];
      if (data[10] != 0 || data[11] != 0) return fail(IST_E_DECODE, "unknown PNG compression / filter method");
      have_ihdr = true;
    } else if (!std::memcmp(type, "IDAT", 4)) idat->insert(idat->end(), data, data + len);
    else if (!std::memcmp(type, "PLTE", 4)) plte->assign(data, data + len);
    else if (!std::memcmp(type, "tRNS", 4)) trns->assign(data, data + len);
    else if (!std::memcmp(type, "IEND", 4)) end = true;
    pos += 12 + int64_t(len);
  }
  if (!have_ihdr || !end) return fail(IST_E_DECODE, "PNG without IHDR / IEND");
  if (H->w == 0 || H->h == 0 || H->w > (1u << 29) || H->h > 0x7FFFFFFFu) return fail(IST_E_DECODE, "bad PNG size");
  const int d = H->depth, c = H->ctype;
  const bool ok = (c == 0 && (d == 1 || d == 2 || d == 4 || d == 8 || d == 16)) || (c == 3 && (d == 1 || d == 2 || d == 4 || d == 8)) ||
                  ((c == 2 || c == 4 || c == 6) && (d == 8 || d == 16));
  if (!ok) return fail(IST_E_DECODE, "bad PNG colour type / bit depth");
  if (c == 3 && plte->size() < 3) return fail(IST_E_DECODE, "palette PNG without PLTE");
  if (H->interlace > 1) return fail(IST_E_DECODE, "unknown PNG interlace method");
  return IST_OK;
}

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

extern "C" {

int ist_png_info(const uint8_t* file, int64_t len, int32_t* width, int32_t* height) {
  Header H; std::vector<uint8_t> idat, plte, trns;
  const int rc = parse(file, len, &H, &idat, &plte, &trns);
  if (rc) return rc;
  if (width) *width = static_cast<int32_t>(H.w);
  if (height) *height = static_cast<int32_t>(H.h);
  return IST_OK;
}

int ist_png_decode_rgba8(const uint8_t* file, int64_t len, uint8_t* out, size_t out_pitch) {
  Header H; std::vector<uint8_t> idat, plte, trns;
  int rc = parse(file, len, &H, &idat, &plte, &trns);
  if (rc) return rc;
  if (!out || out_pitch < size_t(H.w) * 4) return fail(IST_E_INVALID, "ist_png_decode_rgba8: output buffer too small");
  const int channels = H.ctype == 0 ? 1 : H.ctype == 2 ? 3 : H.ctype == 3 ? 1 : H.ctype == 4 ? 2 : 4;
  const int bpp_bits = channels * H.depth;
  const size_t bpp = size_t(bpp_bits + 7) / 8;                          // filter unit in bytes (>= 1)
  // passes: the whole image, or the seven Adam7 sub-images (x0, y0, dx, dy), each filtered on its own
  struct Pass { uint32_t x0, y0, dx, dy; };
  static const Pass whole[1] = {{0, 0, 1, 1}};
  static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  const Pass* passes = H.interlace ? adam7 : whole;
  const int n_passes = H.interlace ? 7 : 1;
  auto pass_w = [&](const Pass& P) { return H.w > P.x0 ? (H.w - P.x0 + P.dx - 1) / P.dx : 0u; };
  auto pass_h = [&](const Pass& P) { return H.h > P.y0 ? (H.h - P.y0 + P.dy - 1) / P.dy : 0u; };
  size_t raw_len = 0;
  for (int k = 0; k < n_passes; ++k) {
    const uint32_t pw = pass_w(passes[k]), ph = pass_h(passes[k]);
    if (pw && ph) raw_len += ((size_t(pw) * bpp_bits + 7) / 8 + 1) * size_t(ph);
  }
  std::vector<uint8_t> raw(raw_len);
  uLongf got = static_cast<uLongf>(raw_len);
  const int zr = uncompress(raw.data(), &got, idat.data(), static_cast<uLong>(idat.size()));
  if (zr != Z_OK || got != raw_len) return fail(IST_E_DECODE, "PNG image data does not inflate to the declared size");

  // tRNS for grey / RGB: one colour is fully transparent
  int t_grey = -1, t_r = -1, t_g = -1, t_b = -1;
  if (H.ctype == 0 && trns.size() >= 2) t_grey = (trns[0] << 8) | trns[1];
  if (H.ctype == 2 && trns.size() >= 6) { t_r = (trns[0] << 8) | trns[1]; t_g = (trns[2] << 8) | trns[3]; t_b = (trns[4] << 8) | trns[5]; }
  const int scale = H.depth < 8 ? 255 / ((1 << H.depth) - 1) : 1;        // 1,2,4-bit greys expand by replication
  const uint8_t* in = raw.data();
  for (int k = 0; k < n_passes; ++k) {
    const Pass& P = passes[k];
    const uint32_t pw = pass_w(P), ph = pass_h(P);
    if (!pw || !ph) continue;
    const size_t stride = (size_t(pw) * bpp_bits + 7) / 8;
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    for (uint32_t py = 0; py < ph; ++py) {
      const int ft = in[0];
      ++in;
      switch (ft) {
        case 0: std::memcpy(cur.data(), in, stride); break;
        case 1: for (size_t i = 0; i < stride; ++i) cur[i] = uint8_t(in[i] + (i >= bpp ? cur[i - bpp] : 0)); break;
        case 2: for (size_t i = 0; i < stride; ++i) cur[i] = uint8_t(in[i] + prev[i]); break;
        case 3: for (size_t i = 0; i < stride; ++i) cur[i] = uint8_t(in[i] + (((i >= bpp ? cur[i - bpp] : 0) + prev[i]) >> 1)); break;
        case 4: for (size_t i = 0; i < stride; ++i) cur[i] = uint8_t(in[i] + paeth(i >= bpp ? cur[i - bpp] : 0, prev[i], i >= bpp ? prev[i - bpp] : 0)); break;
        default: return fail(IST_E_DECODE, "unknown PNG filter type");
      }
      in += stride;
      uint8_t* o = out + size_t(P.y0 + py * P.dy) * out_pitch + size_t(P.x0) * 4;
      const size_t ostep = size_t(P.dx) * 4;
      const uint8_t* s = cur.data();
      for (uint32_t x = 0; x < pw; ++x, o += ostep) {
        if (H.depth == 8) {
          switch (H.ctype) {
            case 6: o[0] = s[4 * x]; o[1] = s[4 * x + 1]; o[2] = s[4 * x + 2]; o[3] = s[4 * x + 3]; break;
            case 2: o[0] = s[3 * x]; o[1] = s[3 * x + 1]; o[2] = s[3 * x + 2];
                    o[3] = (o[0] == t_r && o[1] == t_g && o[2] == t_b) ? 0 : 255; break;
            case 4: o[0] = o[1] = o[2] = s[2 * x]; o[3] = s[2 * x + 1]; break;
            case 0: o[0] = o[1] = o[2] = s[x]; o[3] = (int(s[x]) == t_grey) ? 0 : 255; break;
            default: {
              const size_t idx = s[x];
              if (idx * 3 + 2 >= plte.size()) return fail(IST_E_DECODE, "palette index out of range");
              o[0] = plte[idx * 3]; o[1] = plte[idx * 3 + 1]; o[2] = plte[idx * 3 + 2]; o[3] = idx < trns.size() ? trns[idx] : 255;
            }
          }
        } else if (H.depth == 16) {                                        // keep the high byte (what 8-bit canvases do)
          const uint8_t* q = s + size_t(x) * channels * 2;
          auto v16 = [&](int ch) { return (q[2 * ch] << 8) | q[2 * ch + 1]; };
          switch (H.ctype) {
            case 6: o[0] = q[0]; o[1] = q[2]; o[2] = q[4]; o[3] = q[6]; break;
            case 2: o[0] = q[0]; o[1] = q[2]; o[2] = q[4]; o[3] = (v16(0) == t_r && v16(1) == t_g && v16(2) == t_b) ? 0 : 255; break;
            case 4: o[0] = o[1] = o[2] = q[0]; o[3] = q[2]; break;
            default: o[0] = o[1] = o[2] = q[0]; o[3] = (v16(0) == t_grey) ? 0 : 255;
          }
        } else {                                                           // 1, 2, 4 bits: grey or palette index
          const int per = 8 / H.depth, shift = (per - 1 - int(x % per)) * H.depth;
          const int v = (s[x / per] >> shift) & ((1 << H.depth) - 1);
          if (H.ctype == 3) {
            if (size_t(v) * 3 + 2 >= plte.size()) return fail(IST_E_DECODE, "palette index out of range");
            o[0] = plte[v * 3]; o[1] = plte[v * 3 + 1]; o[2] = plte[v * 3 + 2]; o[3] = size_t(v) < trns.size() ? trns[v] : 255;
          } else {
            o[0] = o[1] = o[2] = uint8_t(v * scale); o[3] = (v == t_grey) ? 0 : 255;
          }
        }
      }
      prev.swap(cur);
    }
  }
  return IST_OK;
}

}  // extern "C"
