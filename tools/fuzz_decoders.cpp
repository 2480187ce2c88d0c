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

extern "C" int ist_misc_info(const uint8_t*, int64_t, int32_t*, int32_t*);
extern "C" int ist_misc_decode_rgba8(const uint8_t*, int64_t, uint8_t*, size_t);

static uint64_t s_rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { s_rng ^= s_rng << 13; s_rng ^= s_rng >> 7; s_rng ^= s_rng << 17; return uint32_t(s_rng >> 16); }

static void one(const std::vector<uint8_t>& f, long* ok, long* bad) {
  const uint8_t* p = f.data(); const int64_t n = int64_t(f.size());
  int32_t w = 0, h = 0, o = 0; int rc;
  if (n >= 2 && p[0] == 0xFF && p[1] == 0xD8) {
    ist::JpegImage J;
    rc = ist::jpeg_parse_and_entropy_decode(p, n, &J, true);
    if (rc == IST_OK && int64_t(J.width) * J.height <= (1 << 24)) rc = ist::jpeg_parse_and_entropy_decode(p, n, &J, false);
    if (rc == IST_OK) {                                        // the container walk that feeds the GPU entropy decoder (de-stuffing, tables)
      ist::JpegImage J2; ist::JpegGpuScan G;
      (void)ist::jpeg_parse_and_entropy_decode(p, n, &J2, false, &G);
      if (G.eligible && (G.slots < 1 || G.slots > 10 || G.bits < 0 || G.stream.size() < size_t(G.bits / 8) + 16)) abort();
    }
  } else {
    const bool misc = n >= 4 && ((p[0] == 'B' && p[1] == 'M') || !memcmp(p, "GIF8", 4));
    rc = misc ? ist_misc_info(p, n, &w, &h) : ist_png_info(p, n, &w, &h);
    if (rc == IST_OK && int64_t(w) * h <= (1 << 24)) {
      std::vector<uint8_t> out(size_t(w) * h * 4);
      rc = misc ? ist_misc_decode_rgba8(p, n, out.data(), size_t(w) * 4) : ist_png_decode_rgba8(p, n, out.data(), size_t(w) * 4);
    }
    (void)o;
  }
  (rc == IST_OK ? *ok : *bad)++;
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
  for (int a = 2; a < argc; ++a) {
    FILE* fp = fopen(argv[a], "rb"); if (!fp) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
    std::vector<uint8_t> seed; uint8_t buf[65536]; size_t r;
    while ((r = fread(buf, 1, sizeof buf, fp)) > 0) seed.insert(seed.end(), buf, buf + r);
    fclose(fp);
    one(seed, &ok, &bad);
    for (int it = 0; it < iters; ++it) {
      std::vector<uint8_t> f = seed;
      const int kind = rnd() % 6;
      if (kind == 0 && f.size() > 1) f.resize(rnd() % f.size());                                        // truncate
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
