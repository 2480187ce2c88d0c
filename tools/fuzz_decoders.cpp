// fuzz_decoders.cpp — mutation fuzzing of the host-side file parsers (PNG, JPEG container + Huffman, BMP, GIF) under
// AddressSanitizer + UBSan on the CPU.  Malformed files must come back as an error code, never as a crash or an
// out-of-bounds access.  Build + run: tools/run_fuzz.sh   (seeds are made by tests/test_fuzz_decoders.py with PIL)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "imagestitch.h"
#include "ist_jpeg.h"
#include "ist_webp.h"

extern "C" int ist_misc_info(const uint8_t*, int64_t, int32_t*, int32_t*);
extern "C" int ist_misc_decode_rgba8(const uint8_t*, int64_t, uint8_t*, size_t, int64_t);

static uint64_t s_rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { s_rng ^= s_rng << 13; s_rng ^= s_rng >> 7; s_rng ^= s_rng << 17; return uint32_t(s_rng >> 16); }

static void one(const std::vector<uint8_t>& f, long* ok, long* bad) {
  const uint8_t* p = f.data(); const int64_t n = int64_t(f.size());
  int32_t w = 0, h = 0, o = 0; int rc;
  if (n >= 2 && p[0] == 0xFF && p[1] == 0xD8) {
    ist::JpegImage J;
    rc = ist::jpeg_parse_and_entropy_decode(p, n, &J, true);
    if (rc == IST_OK && int64_t(J.width) * J.height <= (1 << 24)) rc = ist::jpeg_parse_and_entropy_decode(p, n, &J, false);
    // the container walk that feeds the GPU entropy decoder (de-stuffing, tables) - also for files the host decoder REJECTS:
    // the GPU path sees a file first, and a scan it wrongly calls eligible is decoded without the host's checks
    ist::JpegImage Jh;
    if (ist::jpeg_parse_and_entropy_decode(p, n, &Jh, true) == IST_OK && int64_t(Jh.width) * Jh.height <= (1 << 24)) {
      ist::JpegImage J2; ist::JpegGpuScan G;
      (void)ist::jpeg_parse_and_entropy_decode(p, n, &J2, false, &G);
      if (G.eligible && !G.iv.empty()) {                       // the intervals tile the frame's MCUs exactly, in order
        int64_t at = 0;
        for (const ist::JpegGpuInterval& V : G.iv) { if (V.n_mcus == 0 || int64_t(V.mcu0) != at) abort(); at += V.n_mcus; }
        if (at != int64_t(J2.mcus_x) * J2.mcus_y) abort();
      }
      if (G.eligible && (G.slots < 1 || G.slots > 10 || G.bits < 0 || G.stream.size() > G.stream.capacity())) abort();
      if (G.eligible && G.iv.empty() && G.stream.size() != size_t(G.bits / 8) + 16) abort();
      for (const ist::JpegGpuInterval& V : G.iv)                 // restart intervals: aligned, inside the scan, 16 zero bytes behind each
        if (V.byte_off % 256 || V.bits < 0 || V.bits % 8 || size_t(V.byte_off) + size_t(V.bits / 8) + 16 > G.stream.size() ||
            int64_t(V.mcu0) + V.n_mcus > int64_t(J2.mcus_x) * J2.mcus_y) abort();
    }
  } else if (ist::is_webp(p, n)) {
    rc = ist::webp_info(p, n, &w, &h, &o);
    if (rc == IST_OK && int64_t(w) * h <= (1 << 22)) {
      std::vector<uint8_t> out(size_t(w) * h * 4);
      rc = ist::webp_decode_rgba8(p, n, out.data(), size_t(w) * 4, h);
    }
  } else {
    const bool misc = n >= 4 && ((p[0] == 'B' && p[1] == 'M') || !memcmp(p, "GIF8", 4));
    rc = misc ? ist_misc_info(p, n, &w, &h) : ist_png_info(p, n, &w, &h);
    if (rc == IST_OK && int64_t(w) * h <= (1 << 24)) {
      std::vector<uint8_t> out(size_t(w) * h * 4);
      rc = misc ? ist_misc_decode_rgba8(p, n, out.data(), size_t(w) * 4, h) : ist_png_decode_rgba8(p, n, out.data(), size_t(w) * 4, h);
    }
    (void)o;
  }
  (rc == IST_OK ? *ok : *bad)++;
}

// ---- structure-aware mutations: the parsers' state machines break on VALID segments in the wrong place or number
// (a second SOF / IHDR, a scan that names a component twice), which byte noise practically never produces
struct Piece { size_t at, len; };
static std::vector<Piece> jpeg_segments(const std::vector<uint8_t>& f) {          // marker segments in front of the first scan's data
  std::vector<Piece> v; size_t pos = 2;
  while (pos + 4 <= f.size() && f[pos] == 0xFF) {
    const int m = f[pos + 1];
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) { pos += 2; continue; }
    const size_t len = (size_t(f[pos + 2]) << 8) | f[pos + 3];
    if (len < 2 || pos + 2 + len > f.size()) break;
    v.push_back(Piece{pos, 2 + len});
    pos += 2 + len;
    if (m == 0xDA) break;
  }
  return v;
}
static std::vector<Piece> png_chunks(const std::vector<uint8_t>& f) {
  std::vector<Piece> v; size_t pos = 8;
  while (pos + 12 <= f.size()) {
    const size_t len = (size_t(f[pos]) << 24) | (size_t(f[pos + 1]) << 16) | (size_t(f[pos + 2]) << 8) | f[pos + 3];
    if (pos + 12 + len > f.size()) break;
    v.push_back(Piece{pos, 12 + len});
    pos += 12 + len;
  }
  return v;
}
// duplicate / swap / drop / splice whole pieces; `donor` supplies foreign pieces of the same format (may equal f)
static bool mutate_structure(std::vector<uint8_t>* f, const std::vector<uint8_t>& donor) {
  const bool jpeg = f->size() > 2 && (*f)[0] == 0xFF && (*f)[1] == 0xD8;
  const bool png = f->size() > 8 && (*f)[1] == 'P' && (*f)[2] == 'N';
  if (!jpeg && !png) return false;
  const std::vector<Piece> mine = jpeg ? jpeg_segments(*f) : png_chunks(*f);
  const std::vector<Piece> theirs = jpeg ? jpeg_segments(donor) : png_chunks(donor);
  if (mine.empty() || theirs.empty()) return false;
  const Piece a = mine[rnd() % mine.size()], b = mine[rnd() % mine.size()], t = theirs[rnd() % theirs.size()];
  std::vector<uint8_t> out;
  switch (rnd() % 5) {
    case 0:       // duplicate piece a right behind piece b
      out.assign(f->begin(), f->begin() + b.at + b.len);
      out.insert(out.end(), f->begin() + a.at, f->begin() + a.at + a.len);
      out.insert(out.end(), f->begin() + b.at + b.len, f->end());
      break;
    case 1:       // a foreign piece behind piece b (another file's SOF / IHDR / DHT / PLTE)
      out.assign(f->begin(), f->begin() + b.at + b.len);
      out.insert(out.end(), donor.begin() + t.at, donor.begin() + t.at + t.len);
      out.insert(out.end(), f->begin() + b.at + b.len, f->end());
      break;
    case 2:       // drop piece a
      out.assign(f->begin(), f->begin() + a.at);
      out.insert(out.end(), f->begin() + a.at + a.len, f->end());
      break;
    case 3: {     // swap two pieces
      const Piece lo = a.at <= b.at ? a : b, hi = a.at <= b.at ? b : a;
      if (lo.at + lo.len > hi.at) return false;
      out.assign(f->begin(), f->begin() + lo.at);
      out.insert(out.end(), f->begin() + hi.at, f->begin() + hi.at + hi.len);
      out.insert(out.end(), f->begin() + lo.at + lo.len, f->begin() + hi.at);
      out.insert(out.end(), f->begin() + lo.at, f->begin() + lo.at + lo.len);
      out.insert(out.end(), f->begin() + hi.at + hi.len, f->end());
      break;
    }
    default: {    // a copy of piece a with one payload byte changed, behind the original (a second, different SOF / SOS header)
      std::vector<uint8_t> c(f->begin() + a.at, f->begin() + a.at + a.len);
      const size_t head = jpeg ? 4 : 8;
      if (c.size() > head) c[head + rnd() % (c.size() - head)] = uint8_t(rnd());
      out.assign(f->begin(), f->begin() + a.at + a.len);
      out.insert(out.end(), c.begin(), c.end());
      out.insert(out.end(), f->begin() + a.at + a.len, f->end());
    }
  }
  f->swap(out);
  return true;
}

// "coefs file...": print a hash of the quantised DCT coefficients the entropy decoder produced for each JPEG (the
// CPU-side check that a progressive file and its sequential twin decode to the same coefficients)
static int coef_hashes(int argc, char** argv) {
  for (int a = 2; a < argc; ++a) {
    FILE* fp = fopen(argv[a], "rb"); if (!fp) return 2;
    std::vector<uint8_t> f; uint8_t buf[65536]; size_t r;
    while ((r = fread(buf, 1, sizeof buf, fp)) > 0) f.insert(f.end(), buf, buf + r);
    fclose(fp);
    ist::JpegImage J;
    const int rc = ist::jpeg_parse_and_entropy_decode(f.data(), int64_t(f.size()), &J, false);
    if (rc != IST_OK) { printf("%s error %d %s\n", argv[a], rc, ist_last_error()); continue; }
    uint64_t h = 1469598103934665603ull;
    for (int c = 0; c < J.ncomp; ++c) for (int16_t v : ist::jpeg_dense_coefficients(J.comp[c])) { h ^= uint16_t(v); h *= 1099511628211ull; }
    printf("%s %dx%d scans=%d %016llx\n", argv[a], J.width, J.height, J.scans, (unsigned long long)h);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "coefs")) return coef_hashes(argc, argv);
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  long ok = 0, bad = 0;
  std::vector<std::vector<uint8_t>> seeds;
  for (int a = 2; a < argc; ++a) {
    FILE* fp = fopen(argv[a], "rb"); if (!fp) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
    std::vector<uint8_t> seed; uint8_t buf[65536]; size_t r;
    while ((r = fread(buf, 1, sizeof buf, fp)) > 0) seed.insert(seed.end(), buf, buf + r);
    fclose(fp);
    seeds.push_back(seed);
  }
  for (size_t a = 0; a < seeds.size(); ++a) {
    const std::vector<uint8_t>& seed = seeds[a];
    one(seed, &ok, &bad);
    for (int it = 0; it < iters; ++it) {
      std::vector<uint8_t> f = seed;
      const int kind = rnd() % 9;
      if (kind >= 6) {                                                                                    // whole segments / chunks
        const std::vector<uint8_t>& donor = seeds[rnd() % seeds.size()];
        const bool same = donor.size() > 2 && f.size() > 2 && donor[0] == f[0] && donor[1] == f[1];
        if (!mutate_structure(&f, same ? donor : seed)) f[rnd() % f.size()] ^= 0x80;
        if (kind == 8 && f.size() > 1) f[rnd() % f.size()] = uint8_t(rnd());                             // ... plus one noisy byte
      }
      else if (kind == 0 && f.size() > 1) f.resize(rnd() % f.size());                                   // truncate
      else if (kind == 1) { const int k = 1 + rnd() % 8; for (int i = 0; i < k; ++i) f[rnd() % f.size()] = uint8_t(rnd()); }        // random bytes
      else if (kind == 2) { const int k = 1 + rnd() % 4; for (int i = 0; i < k; ++i) f[rnd() % f.size()] ^= uint8_t(1u << (rnd() % 8)); } // bit flips
      else if (kind == 3) { const size_t at = rnd() % (f.size() < 64 ? f.size() : 64); f[at] = uint8_t(rnd()); }                    // header byte
      else if (kind == 4) { const size_t at = rnd() % f.size(); const uint8_t v = (rnd() & 1) ? 0xFF : 0x00; const size_t k = 1 + rnd() % 16; for (size_t i = at; i < f.size() && i < at + k; ++i) f[i] = v; }
      else { const size_t at = rnd() % f.size(), k = rnd() % 64; f.insert(f.begin() + at, k, uint8_t(rnd())); }                      // insert
      if (f.empty()) f.push_back(0);
      one(f, &ok, &bad);
    }
  }
  printf("fuzz: %ld decoded, %ld rejected, 0 crashes\n", ok, bad);
  return 0;
}
