// fuzz_compile.cpp — random and hostile op lists through the host-side op-list compiler (ist_compile.cpp) under
// ASan + UBSan, checking the invariants the HIP kernel relies on for memory safety:
//   * every draw's clamp box lies inside its bitmap; every stack entry / cell.op names a resolved op
//   * cells lie inside the render region, do not overlap, and tile it completely unless HOLE ops reserve part of it
//   * every COPY cell's 1:1 index map stays inside the clamp box
//   * the LDS footprint the host asks for fits the 64 KiB a workgroup may have
// Build + run: tools/run_fuzz.sh compile ITERS
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "ist_internal.h"

using namespace ist;

static uint64_t s_rng = 0x243F6A8885A308D3ull;
static uint32_t rnd() { s_rng ^= s_rng << 13; s_rng ^= s_rng >> 7; s_rng ^= s_rng << 17; return uint32_t(s_rng >> 16); }
static double uni() { return (rnd() & 0xFFFFFF) / double(0x1000000); }

static double hostile() {
  switch (rnd() % 10) {
    case 0: return std::numeric_limits<double>::quiet_NaN();
    case 1: return std::numeric_limits<double>::infinity();
    case 2: return -std::numeric_limits<double>::infinity();
    case 3: return 1e300;
    case 4: return -1e300;
    case 5: return 1e-300;
    case 6: return 4294967296.0 * (rnd() % 1024);
    case 7: return -2147483649.0;
    case 8: return 0.0;
    default: return (uni() - 0.5) * 1e12;
  }
}

// (iter is the loop variable of the function the macro is used in)
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "invariant failed at %s:%d: %s (iteration %ld)\n", __FILE__, __LINE__, #c, iter); abort(); } } while (0)

// hostile image lists / caps through the planner, then its own op list through the compiler
static void fuzz_planner(long iters) {
  long ok = 0, bad = 0;
  for (long iter = 0; iter < iters; ++iter) {
    const int n = rnd() % 12;
    std::vector<ist_image_desc> imgs(n ? n : 1);
    for (auto& I : imgs) {
      std::memset(&I, 0, sizeof I);
      I.width = 1 + rnd() % 900; I.height = 1 + rnd() % 900; I.orientation = rnd() % 9; I.opaque = rnd() & 1;
      I.file_size = (rnd() & 3) ? int64_t(rnd()) * 16 : 0;
      if (!(rnd() & 7)) { I.bmp_width = 1 + rnd() % 5000; I.bmp_height = 1 + rnd() % 5000; }
      if (!(rnd() & 31)) { I.width = int(rnd()) ; I.height = -int(rnd() % 100); }
      if (!(rnd() & 63)) I.orientation = int(rnd());
    }
    ist_limits L;
    if (rnd() & 1) ist_limits_unlimited(&L); else ist_limits_default(rnd() % 4, &L);
    if (!(rnd() & 7)) { L.max_side = hostile(); }
    if (!(rnd() & 7)) { L.max_pixels = hostile(); }
    if (!(rnd() & 7)) { L.max_super_sample = (rnd() & 1) ? hostile() : 1.0 + 3.0 * uni(); }
    if (!(rnd() & 31)) L.platform = int(rnd());
    const double gap = (rnd() & 1) ? 0.0 : (rnd() & 7) ? double(rnd() % 60) : hostile();
    const int direction = (rnd() & 31) ? int(rnd() & 1) : int(rnd() % 7) - 3, mode = (rnd() & 31) ? int(rnd() % 3) : int(rnd() % 9) - 3;
    ist_plan P; std::memset(&P, 0, sizeof P);
    const int rc = ist_plan_compute(n ? imgs.data() : nullptr, n, direction, mode, gap, &L, &P);
    if (rc != IST_OK) { ++bad; ist_plan_free(&P); continue; }
    CHECK(P.canvas_w >= 1 && P.canvas_h >= 1 && P.n_rects == n);
    for (int i = 0; i < P.n_rects; ++i) CHECK(P.rects[i].image >= 0 && P.rects[i].image < n);     // hostile caps may give empty or non-finite rects: the compiler must drop or refuse those
    std::vector<ist_op> ops(P.n_rects + 1);
    int n_ops = 0;
    const int rc2 = ist_plan_ops(&P, imgs.data(), n, ops.data(), &n_ops);
    if (rc2 == IST_E_DECODE) { ++bad; ist_plan_free(&P); continue; }     // a bitmap without pixels is refused at draw time (index.js:1512)
    CHECK(rc2 == IST_OK && n_ops == P.n_rects + 1);
    if (P.canvas_w <= (1 << 29) && P.canvas_h < 2147483647LL) {
      Compiled C;
      const uint8_t clear[4] = {0, 0, 0, 0};
      const int rc3 = compile_ops(P.canvas_w, P.canvas_h, clear, ops.data(), n_ops, imgs.data(), n, IST_FILTER_BILINEAR, nullptr, &C);
      if (rc3 == IST_OK) {
        ++ok;
        int64_t area = 0;
        for (const DevCell& c : C.cells) area += int64_t(c.X1 - c.X0) * (c.Y1 - c.Y0);
        CHECK(area == P.canvas_w * P.canvas_h);            // the white fill alone covers the canvas
        for (const DevOp& r : C.ops) if (!(r.flags & OPF_FILL)) CHECK(r.cx0 >= 0 && r.cx1 < C.img_w[r.image] && r.cy0 >= 0 && r.cy1 < C.img_h[r.image] && r.cx0 <= r.cx1 && r.cy0 <= r.cy1);
      } else ++bad;
    }
    ist_plan_free(&P);
  }
  printf("fuzz-planner: %ld planned + compiled, %ld rejected, all invariants hold\n", ok, bad);
}

int main(int argc, char** argv) {
  const long iters = argc > 1 ? atol(argv[1]) : 20000;
  fuzz_planner(iters / 8);
  long ok = 0, bad = 0;
  for (long iter = 0; iter < iters; ++iter) {
    const int64_t cw = 1 + rnd() % ((rnd() & 7) ? 700 : 70000), ch = 1 + rnd() % ((rnd() & 7) ? 700 : 70000);
    const int n_img = 1 + rnd() % 4;
    std::vector<ist_image_desc> imgs(n_img);
    for (auto& I : imgs) { std::memset(&I, 0, sizeof I); I.width = 1 + rnd() % 500; I.height = 1 + rnd() % 500; I.opaque = rnd() & 1; if (!(rnd() & 7)) { I.bmp_width = 1 + rnd() % 500; I.bmp_height = 1 + rnd() % 500; } }
    const int n_ops = rnd() % 7;
    std::vector<ist_op> ops(n_ops);
    bool holes = false;
    for (auto& o : ops) {
      std::memset(&o, 0, sizeof o);
      const int kind = rnd() % 8;
      o.kind = kind == 0 ? IST_OP_FILL : (kind == 1 && !(rnd() & 3) ? IST_OP_HOLE : IST_OP_DRAW);
      if (!(rnd() & 63)) o.kind = int(rnd() % 5) - 1;
      holes |= o.kind == IST_OP_HOLE;
      o.image = rnd() % n_img;
      if (!(rnd() & 63)) o.image = int(rnd() % 9) - 2;
      const double sc = (rnd() & 1) ? 1.0 : 0.1 + 4.0 * uni();
      const int t = rnd() % 8;                       // the eight axis-aligned orientations
      const double sx = (t & 1) ? -sc : sc, sy = (t & 2) ? -sc : sc;
      if (t & 4) { o.m[0] = 0; o.m[1] = sx; o.m[2] = sy; o.m[3] = 0; } else { o.m[0] = sx; o.m[1] = 0; o.m[2] = 0; o.m[3] = sy; }
      o.m[4] = (rnd() & 1) ? double(int(rnd() % 1400) - 700) : (uni() - 0.5) * 1400; o.m[5] = (rnd() & 1) ? double(int(rnd() % 1400) - 700) : (uni() - 0.5) * 1400;
      const bool ints = rnd() & 1;
      for (int i = 0; i < 4; ++i) {
        o.s[i] = ints ? double(int(rnd() % 600) - (i < 2 ? 50 : 0)) : uni() * 600 - (i < 2 ? 50 : 0);
        o.d[i] = ints ? double(int(rnd() % 900) - (i < 2 ? 300 : 0)) : uni() * 900 - (i < 2 ? 300 : 0);
      }
      if (rnd() & 1) { o.s[0] = 0; o.s[1] = 0; o.s[2] = imgs[o.image >= 0 && o.image < n_img ? o.image : 0].width; o.s[3] = imgs[o.image >= 0 && o.image < n_img ? o.image : 0].height; if (rnd() & 1) { o.d[2] = o.s[2]; o.d[3] = o.s[3]; } }
      o.rgba[0] = rnd(); o.rgba[1] = rnd(); o.rgba[2] = rnd(); o.rgba[3] = (rnd() & 7) ? 255 : rnd();
      if (!(rnd() & 15)) { const int k = 1 + rnd() % 3; for (int i = 0; i < k; ++i) { const int w = rnd() % 14; (w < 6 ? o.m[w] : w < 10 ? o.s[w - 6] : o.d[w - 10]) = hostile(); } }
    }
    ist_region clip; const bool use_clip = !(rnd() & 3);
    clip.x = int(rnd() % 800) - 100; clip.y = int(rnd() % 800) - 100; clip.w = rnd() % 900; clip.h = rnd() % 900;
    const uint8_t clear[4] = {uint8_t(rnd()), uint8_t(rnd()), uint8_t(rnd()), uint8_t((rnd() & 1) ? 255 : (rnd() & 1) ? 0 : rnd())};
    const int filter = ((rnd() & 1) ? IST_FILTER_BILINEAR : IST_FILTER_NEAREST) | ((rnd() & 3) ? 0 : IST_FILTER_EDGE_AA);
    Compiled C;
    const int rc = compile_ops(cw, ch, clear, ops.data(), n_ops, imgs.data(), n_img, filter, use_clip ? &clip : nullptr, &C);
    if (rc != IST_OK) { ++bad; continue; }
    ++ok;
    CHECK(C.rx0 >= 0 && C.ry0 >= 0 && C.rx1 <= cw && C.ry1 <= ch && C.rx0 < C.rx1 && C.ry0 < C.ry1);
    CHECK(C.lds_words >= 0 && C.lds_words <= 16384);
    for (const DevOp& r : C.ops) {
      CHECK(r.X0 >= C.rx0 && r.X1 <= C.rx1 && r.Y0 >= C.ry0 && r.Y1 <= C.ry1 && r.X0 < r.X1 && r.Y0 < r.Y1);
      if (r.flags & (OPF_FILL | OPF_HOLE)) { CHECK(r.image == -1); continue; }
      CHECK(r.image >= 0 && r.image < n_img);
      CHECK(r.cx0 >= 0 && r.cx0 <= r.cx1 && r.cx1 < C.img_w[r.image] && r.cy0 >= 0 && r.cy0 <= r.cy1 && r.cy1 < C.img_h[r.image]);
      CHECK(std::isfinite(r.kx) && std::isfinite(r.ky) && std::isfinite(r.ox) && std::isfinite(r.oy));
    }
    int64_t area = 0, tiles = 0;
    for (size_t ci = 0; ci < C.cells.size(); ++ci) {
      const DevCell& c = C.cells[ci];
      CHECK(c.X0 >= C.rx0 && c.X1 <= C.rx1 && c.Y0 >= C.ry0 && c.Y1 <= C.ry1 && c.X0 < c.X1 && c.Y0 < c.Y1);
      CHECK(c.stack_off >= 0 && c.stack_len >= 0 && size_t(c.stack_off) + c.stack_len <= C.stacks.size());
      for (int k = 0; k < c.stack_len; ++k) { const int32_t e = C.stacks[c.stack_off + k]; CHECK(e >= 0 && size_t(e) < C.ops.size()); CHECK(!(C.ops[e].flags & OPF_HOLE)); }
      CHECK(c.tile_w >= 1 && c.tile_h >= 1 && c.tiles_x == (c.X1 - c.X0 + c.tile_w - 1) / c.tile_w);
      tiles += int64_t(c.tiles_x) * ((c.Y1 - c.Y0 + c.tile_h - 1) / c.tile_h);
      area += int64_t(c.X1 - c.X0) * (c.Y1 - c.Y0);
      if (c.path != PATH_FILL) { CHECK(c.stack_len >= 1 && c.op == C.stacks[c.stack_off]); }
      if (c.path == PATH_COPY || c.path == PATH_SAMPLE || c.path == PATH_SAMPLE_LDS || c.path == PATH_SAMPLE_STREAM || c.path == PATH_SWAP_LDS) { CHECK(c.stack_len == 1); CHECK(!(C.ops[c.op].flags & OPF_FILL)); }
      if (c.path == PATH_SWAP_LDS) CHECK(C.ops[c.op].flags & OPF_SWAP);
      if (c.path == PATH_COPY || c.path == PATH_SAMPLE || c.path == PATH_SAMPLE_LDS || c.path == PATH_SAMPLE_STREAM) CHECK(!(C.ops[c.op].flags & OPF_SWAP));
      if (c.path == PATH_SAMPLE_STREAM) {   // the kernel's own re-check must hold: ring fits the launch's LDS, tile fits the row-tap lanes
        CHECK(c.sub_h >= 2 && c.tile_h <= 64 && (c.tile_w == 64 || c.tile_w == 128 || c.tile_w == 256));
        const int64_t wl = (static_cast<int64_t>(std::floor((c.tile_w - 1) * std::fabs(C.ops[c.op].kx))) + 6) & ~3LL;
        CHECK(8 * c.sub_h * wl <= C.lds_words);
      }
      if (c.path == PATH_COPY) {
        const DevOp& r = C.ops[c.op];
        CHECK(r.flags & OPF_IDENTITY);
        for (int corner = 0; corner < 2; ++corner) {
          const int64_t X = corner ? c.X1 - 1 : c.X0, Y = corner ? c.Y1 - 1 : c.Y0;
          const int64_t ix = (r.flags & OPF_FLIPX) ? int64_t(r.ox) - 1 - X : X + int64_t(r.ox), iy = (r.flags & OPF_FLIPY) ? int64_t(r.oy) - 1 - Y : Y + int64_t(r.oy);
          CHECK(ix >= r.cx0 && ix <= r.cx1 && iy >= r.cy0 && iy <= r.cy1);
        }
      }
      for (size_t cj = 0; cj < ci; ++cj) {           // no overlap (cells are few: the grid is small)
        const DevCell& o = C.cells[cj];
        CHECK(!(c.X0 < o.X1 && o.X0 < c.X1 && c.Y0 < o.Y1 && o.Y0 < c.Y1));
      }
    }
    CHECK(tiles == C.info.n_tiles);
    if (!holes) CHECK(area == (C.rx1 - C.rx0) * (C.ry1 - C.ry0)); else CHECK(area <= (C.rx1 - C.rx0) * (C.ry1 - C.ry0));
    // bands partition the cells (each cell in exactly one band) and are launched in array order: expensive kinds first
    int64_t bt = 0;
    std::vector<char> in_band(C.cells.size(), 0);
    int last_weight = 3;
    for (const DevBand& b : C.bands) {
      for (int k = 0; k < b.n_cells; ++k) { CHECK(!in_band[b.first_cell + k]); in_band[b.first_cell + k] = 1; }
      const int32_t bp = C.cells[b.first_cell].path;
      const int weight = bp == PATH_FILL ? 0 : bp == PATH_COPY ? 1 : 2;
      CHECK(weight <= last_weight);
      last_weight = weight;
      CHECK(b.tile_begin == bt && b.n_cells >= 1 && size_t(b.first_cell) + b.n_cells <= C.cells.size());
      int per_row = 0;
      const DevCell& f = C.cells[b.first_cell];
      for (int k = 0; k < b.n_cells; ++k) { const DevCell& c = C.cells[b.first_cell + k]; CHECK(c.Y0 == f.Y0 && c.Y1 == f.Y1 && c.tile_h == f.tile_h && c.band_x == per_row); per_row += c.tiles_x; }
      CHECK(per_row == b.tiles_per_row);
      bt += int64_t(per_row) * ((f.Y1 - f.Y0 + f.tile_h - 1) / f.tile_h);
    }
    CHECK(bt == tiles);
    for (char v : in_band) CHECK(v);
  }
  printf("fuzz-compile: %ld compiled, %ld rejected, all invariants hold\n", ok, bad);
  return 0;
}
