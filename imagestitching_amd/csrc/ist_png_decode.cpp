// ist_png_decode.cpp — PNG file -> RGBA8 (straight alpha), the decode step in front of the stitch path for PNG inputs
// (SURVEY.md section 8f rank 3).
//
// Reference anchor: loadImageFrom (utils/canvas.js:27-121) sets Image.src and the platform decodes the file into a
// bitmap; 'png' is one of SUPPORTED_IMAGE_TYPES (pages/index/index.js:4).  PNG is lossless, so the result is pinned:
// every conforming decoder yields the same RGBA bytes (tests compare with PIL, bit for bit).
//
// Host code: inflate is zlib's (libz is part of the image), un-filtering is the five PNG predictors.  Un-filtering is
// row- and pixel-serial by construction (Sub/Average/Paeth depend on the reconstructed left neighbour), so it stays on
// the host and the decoded rows are what gets uploaded.  For speed the IDAT chunks are inflated in place (no
// concatenated copy) in blocks of rows that stay in cache while they are un-filtered (loops specialised per pixel size)
// and expanded to RGBA (loops specialised per colour type).
// Supported: colour types 0, 2, 3, 4, 6; bit depths 1-16; tRNS; plain and Adam7-interlaced.
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ist_internal.h"

using namespace ist;

namespace {

inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

struct Header { uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0; };

struct Span { const uint8_t* p; uint32_t n; };

// walks the chunks; collects the IDAT spans (in place), PLTE, tRNS.  header_only stops after a valid IHDR.
int parse(const uint8_t* f, int64_t n, Header* H, std::vector<Span>* idat, std::vector<uint8_t>* plte, std::vector<uint8_t>* trns, bool header_only) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (!f || n < 8 + 25 + 12) return fail(IST_E_DECODE, "not a PNG file (too short)");
  if (std::memcmp(f, sig, 8) != 0) {
    if (f[0] == 0xFF && f[1] == 0xD8) return fail(IST_E_UNSUPPORTED, "this is a JPEG file: use ist_image_decode_rgba8");
    if (n >= 12 && !std::memcmp(f, "RIFF", 4) && !std::memcmp(f + 8, "WEBP", 4)) return fail(IST_E_UNSUPPORTED, "this is a WebP file: use ist_image_decode_rgba8");
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
#ifndef IST_FUZZ_NO_CRC      // the fuzz harness builds without the check so that mutated files reach the code behind it
    if (uint32_t(crc32(crc32(0L, type, 4), data, len)) != crc) return fail(IST_E_DECODE, "PNG chunk CRC mismatch");
#else
    (void)crc;
#endif
    // IHDR is the first chunk and the only one of its kind (PNG 5.6): info and decode must agree on ONE header,
    // every buffer of the caller is sized from it
    if (pos == 8 && std::memcmp(type, "IHDR", 4) != 0) return fail(IST_E_DECODE, "PNG does not start with IHDR");
    if (!std::memcmp(type, "IHDR", 4)) {
      if (have_ihdr) return fail(IST_E_DECODE, "PNG with more than one IHDR");
      if (len != 13) return fail(IST_E_DECODE, "bad IHDR");
      H->w = be32(data); H->h = be32(data + 4); H->depth = data[8]; H->ctype = data[9]; H->interlace = data[12];
      if (data[10] != 0 || data[11] != 0) return fail(IST_E_DECODE, "unknown PNG compression / filter method");
      have_ihdr = true;
      if (header_only) break;
    } else if (!std::memcmp(type, "IDAT", 4)) { if (len) idat->push_back(Span{data, len}); }
    else if (!std::memcmp(type, "PLTE", 4)) plte->assign(data, data + len);
    else if (!std::memcmp(type, "tRNS", 4)) trns->assign(data, data + len);
    else if (!std::memcmp(type, "IEND", 4)) end = true;
    pos += 12 + int64_t(len);
  }
  if (!have_ihdr || (!end && !header_only)) return fail(IST_E_DECODE, "PNG without IHDR / IEND");
  if (H->w == 0 || H->h == 0 || H->w > (1u << 29) || H->h > 0x7FFFFFFFu) return fail(IST_E_DECODE, "bad PNG size");
  const int d = H->depth, c = H->ctype;
  const bool ok = (c == 0 && (d == 1 || d == 2 || d == 4 || d == 8 || d == 16)) || (c == 3 && (d == 1 || d == 2 || d == 4 || d == 8)) ||
                  ((c == 2 || c == 4 || c == 6) && (d == 8 || d == 16));
  if (!ok) return fail(IST_E_DECODE, "bad PNG colour type / bit depth");
  if (H->interlace > 1) return fail(IST_E_DECODE, "unknown PNG interlace method");
  if (!header_only && c == 3 && plte->size() < 3) return fail(IST_E_DECODE, "palette PNG without PLTE");
  return IST_OK;
}

// ---- un-filtering, in place: cur holds the filtered bytes of one scanline, prev the reconstructed scanline above
template <int BPP>
void unfilter_t(int ft, uint8_t* cur, const uint8_t* prev, size_t n) {
  const size_t head = n < size_t(BPP) ? n : size_t(BPP);
  switch (ft) {
    case 1:
      for (size_t i = BPP; i < n; ++i) cur[i] = uint8_t(cur[i] + cur[i - BPP]);
      break;
    case 2:
      for (size_t i = 0; i < n; ++i) cur[i] = uint8_t(cur[i] + prev[i]);
      break;
    case 3:
      for (size_t i = 0; i < head; ++i) cur[i] = uint8_t(cur[i] + (prev[i] >> 1));
      for (size_t i = BPP; i < n; ++i) cur[i] = uint8_t(cur[i] + ((cur[i - BPP] + prev[i]) >> 1));
      break;
    case 4:
      for (size_t i = 0; i < head; ++i) cur[i] = uint8_t(cur[i] + prev[i]);            // left = upper-left = 0: the predictor is `up`
      for (size_t i = BPP; i < n; ++i) {
        const int a = cur[i - BPP], b = prev[i], c = prev[i - BPP];
        const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
        const int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
        cur[i] = uint8_t(cur[i] + pr);
      }
      break;
    default: break;
  }
}

int unfilter(int ft, uint8_t* cur, const uint8_t* prev, size_t n, size_t bpp) {
  if (ft < 0 || ft > 4) return fail(IST_E_DECODE, "unknown PNG filter type");
  switch (bpp) {
    case 1: unfilter_t<1>(ft, cur, prev, n); break;
    case 2: unfilter_t<2>(ft, cur, prev, n); break;
    case 3: unfilter_t<3>(ft, cur, prev, n); break;
    case 4: unfilter_t<4>(ft, cur, prev, n); break;
    case 6: unfilter_t<6>(ft, cur, prev, n); break;
    default: unfilter_t<8>(ft, cur, prev, n); break;
  }
  return IST_OK;
}

// ---- one reconstructed scanline -> RGBA8; output pixels are `ostep` bytes apart (4 for a plain image, 4*dx for an Adam7 pass)
struct Expand {
  const Header* H; const std::vector<uint8_t>* plte; const std::vector<uint8_t>* trns;
  int t_grey = -1, t_r = -1, t_g = -1, t_b = -1, scale = 1;
};

int expand_row(const Expand& E, const uint8_t* s, uint32_t npix, uint8_t* o, size_t ostep) {
  const Header& H = *E.H;
  const std::vector<uint8_t>& plte = *E.plte; const std::vector<uint8_t>& trns = *E.trns;
  if (H.depth == 8) {
    switch (H.ctype) {
      case 6:
        if (ostep == 4) { std::memcpy(o, s, size_t(npix) * 4); break; }
        for (uint32_t x = 0; x < npix; ++x, o += ostep) std::memcpy(o, s + 4 * size_t(x), 4);
        break;
      case 2:
        if (E.t_r < 0) { for (uint32_t x = 0; x < npix; ++x, o += ostep, s += 3) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 255; } }
        else for (uint32_t x = 0; x < npix; ++x, o += ostep, s += 3) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = (s[0] == E.t_r && s[1] == E.t_g && s[2] == E.t_b) ? 0 : 255; }
        break;
      case 4: for (uint32_t x = 0; x < npix; ++x, o += ostep) { o[0] = o[1] = o[2] = s[2 * x]; o[3] = s[2 * x + 1]; } break;
      case 0: for (uint32_t x = 0; x < npix; ++x, o += ostep) { o[0] = o[1] = o[2] = s[x]; o[3] = (int(s[x]) == E.t_grey) ? 0 : 255; } break;
      default:
        for (uint32_t x = 0; x < npix; ++x, o += ostep) {
          const size_t idx = s[x];
          if (idx * 3 + 2 >= plte.size()) return fail(IST_E_DECODE, "palette index out of range");
          o[0] = plte[idx * 3]; o[1] = plte[idx * 3 + 1]; o[2] = plte[idx * 3 + 2]; o[3] = idx < trns.size() ? trns[idx] : 255;
        }
    }
  } else if (H.depth == 16) {                                        // keep the high byte (what 8-bit canvases do)
    const int channels = H.ctype == 0 ? 1 : H.ctype == 2 ? 3 : H.ctype == 4 ? 2 : 4;
    for (uint32_t x = 0; x < npix; ++x, o += ostep) {
      const uint8_t* q = s + size_t(x) * channels * 2;
      auto v16 = [&](int ch) { return (q[2 * ch] << 8) | q[2 * ch + 1]; };
      switch (H.ctype) {
        case 6: o[0] = q[0]; o[1] = q[2]; o[2] = q[4]; o[3] = q[6]; break;
        case 2: o[0] = q[0]; o[1] = q[2]; o[2] = q[4]; o[3] = (v16(0) == E.t_r && v16(1) == E.t_g && v16(2) == E.t_b) ? 0 : 255; break;
        case 4: o[0] = o[1] = o[2] = q[0]; o[3] = q[2]; break;
        default: o[0] = o[1] = o[2] = q[0]; o[3] = (v16(0) == E.t_grey) ? 0 : 255;
      }
    }
  } else {                                                           // 1, 2, 4 bits: grey or palette index
    const int per = 8 / H.depth;
    for (uint32_t x = 0; x < npix; ++x, o += ostep) {
      const int shift = (per - 1 - int(x % per)) * H.depth;
      const int v = (s[x / per] >> shift) & ((1 << H.depth) - 1);
      if (H.ctype == 3) {
        if (size_t(v) * 3 + 2 >= plte.size()) return fail(IST_E_DECODE, "palette index out of range");
        o[0] = plte[v * 3]; o[1] = plte[v * 3 + 1]; o[2] = plte[v * 3 + 2]; o[3] = size_t(v) < trns.size() ? trns[v] : 255;
      } else {
        o[0] = o[1] = o[2] = uint8_t(v * E.scale); o[3] = (v == E.t_grey) ? 0 : 255;
      }
    }
  }
  return IST_OK;
}

// zlib inflate over the IDAT spans in place: fills `out` completely or fails
struct Inflater {
  z_stream z; const std::vector<Span>* spans; size_t next = 0; bool ok = false, done = false;
  explicit Inflater(const std::vector<Span>* s) : spans(s) { std::memset(&z, 0, sizeof z); ok = inflateInit(&z) == Z_OK; }
  ~Inflater() { if (ok) inflateEnd(&z); }
  // returns the number of bytes produced (== want unless the stream ended or is damaged: then -1 on damage)
  int64_t read(uint8_t* out, size_t want) {
    z.next_out = out; z.avail_out = static_cast<uInt>(want);
    while (z.avail_out > 0 && !done) {
      if (z.avail_in == 0) {
        if (next >= spans->size()) break;
        z.next_in = const_cast<Bytef*>((*spans)[next].p); z.avail_in = (*spans)[next].n; ++next;
      }
      const int r = inflate(&z, Z_NO_FLUSH);
      if (r == Z_STREAM_END) { done = true; break; }
      if (r != Z_OK && r != Z_BUF_ERROR) return -1;
      if (r == Z_BUF_ERROR && z.avail_in == 0 && next >= spans->size()) break;
    }
    return static_cast<int64_t>(want - z.avail_out);
  }
};

}  // namespace

extern "C" {

int ist_png_info(const uint8_t* file, int64_t len, int32_t* width, int32_t* height) {
  Header H; std::vector<Span> idat; std::vector<uint8_t> plte, trns;
  const int rc = parse(file, len, &H, &idat, &plte, &trns, true);          // IHDR only: no walk over the image data
  if (rc) return rc;
  if (width) *width = static_cast<int32_t>(H.w);
  if (height) *height = static_cast<int32_t>(H.h);
  return IST_OK;
}

static int png_decode(const uint8_t* file, int64_t len, uint8_t* out, size_t out_pitch, int64_t out_rows) {
  Header H; std::vector<Span> idat; std::vector<uint8_t> plte, trns;
  int rc = parse(file, len, &H, &idat, &plte, &trns, false);
  if (rc) return rc;
  if (!out || out_pitch < size_t(H.w) * 4 || out_rows < int64_t(H.h)) return fail(IST_E_INVALID, "ist_png_decode_rgba8: output buffer too small");
  const int channels = H.ctype == 0 ? 1 : H.ctype == 2 ? 3 : H.ctype == 3 ? 1 : H.ctype == 4 ? 2 : 4;
  const int bpp_bits = channels * H.depth;
  const size_t bpp = size_t(bpp_bits + 7) / 8;                          // filter unit in bytes (>= 1)
  Expand E; E.H = &H; E.plte = &plte; E.trns = &trns;
  // tRNS for grey / RGB: one colour is fully transparent
  if (H.ctype == 0 && trns.size() >= 2) E.t_grey = (trns[0] << 8) | trns[1];
  if (H.ctype == 2 && trns.size() >= 6) { E.t_r = (trns[0] << 8) | trns[1]; E.t_g = (trns[2] << 8) | trns[3]; E.t_b = (trns[4] << 8) | trns[5]; }
  E.scale = H.depth < 8 ? 255 / ((1 << H.depth) - 1) : 1;                // 1,2,4-bit greys expand by replication
  Inflater inf(&idat);
  if (!inf.ok) return fail(IST_E_NOMEM, "zlib initialisation failed");
  static const char* kShort = "PNG image data does not inflate to the declared size";

  // passes: the whole image, or the seven Adam7 sub-images (x0, y0, dx, dy), each filtered on its own
  struct Pass { uint32_t x0, y0, dx, dy; };
  static const Pass whole[1] = {{0, 0, 1, 1}};
  static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  const Pass* passes = H.interlace ? adam7 : whole;
  const int n_passes = H.interlace ? 7 : 1;
  std::vector<uint8_t> block, prev;
  for (int k = 0; k < n_passes; ++k) {
    const Pass& P = passes[k];
    const uint32_t pw = H.w > P.x0 ? (H.w - P.x0 + P.dx - 1) / P.dx : 0u;
    const uint32_t ph = H.h > P.y0 ? (H.h - P.y0 + P.dy - 1) / P.dy : 0u;
    if (!pw || !ph) continue;
    const size_t stride = (size_t(pw) * bpp_bits + 7) / 8;
    // inflate a block of scanlines (~512 KiB: stays in cache), un-filter and expand them, repeat
    const uint32_t rows_per_block = static_cast<uint32_t>(std::max<size_t>(1, (512u << 10) / (stride + 1)));
    block.resize((stride + 1) * size_t(rows_per_block));
    prev.assign(stride, 0);
    for (uint32_t py = 0; py < ph; py += rows_per_block) {
      const uint32_t nr = std::min(rows_per_block, ph - py);
      const size_t want = (stride + 1) * size_t(nr);
      const int64_t got = inf.read(block.data(), want);
      if (got != static_cast<int64_t>(want)) return fail(IST_E_DECODE, kShort);
      const uint8_t* up = prev.data();
      for (uint32_t r = 0; r < nr; ++r) {
        uint8_t* line = block.data() + (stride + 1) * size_t(r);
        rc = unfilter(line[0], line + 1, up, stride, bpp);
        if (rc) return rc;
        rc = expand_row(E, line + 1, pw, out + size_t(P.y0 + (py + r) * P.dy) * out_pitch + size_t(P.x0) * 4, size_t(P.dx) * 4);
        if (rc) return rc;
        up = line + 1;
      }
      std::memcpy(prev.data(), block.data() + (stride + 1) * size_t(nr - 1) + 1, stride);      // the next block's row above
    }
  }
  // the stream must end here: a longer one is as wrong as a shorter one
  uint8_t extra;
  if (inf.read(&extra, 1) != 0) return fail(IST_E_DECODE, kShort);
  return IST_OK;
}

int ist_png_decode_rgba8(const uint8_t* file, int64_t len, uint8_t* out, size_t out_pitch, int64_t out_rows) {
  try { return png_decode(file, len, out, out_pitch, out_rows); }
  catch (const std::bad_alloc&) { return fail(IST_E_NOMEM, "out of memory while decoding the PNG"); }
}

}  // extern "C"
