// ist_internal.h — shared between the planner, the op-list compiler and the HIP launch code.
#ifndef IST_INTERNAL_H_
#define IST_INTERNAL_H_

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/imagestitch.h"

namespace ist {

extern thread_local std::string g_last_error;
extern thread_local int g_last_code;
int fail(int code, const std::string& msg);

// Canvas current transformation matrix:  X = a*u + c*v + e ;  Y = b*u + d*v + f
struct Ctm {
  double a = 1.0, b = 0.0, c = 0.0, d = 1.0, e = 0.0, f = 0.0;
  void translate(double x, double y);
  void scale(double x, double y);
  void rotate(double rad);
};

// ---- device-visible tables -----------------------------------------------------------------------------------
// A draw resolved into canvas space.  Sampling map (the arithmetic contract shared with the oracle):
//     sxf = kx * Wc + ox ,  syf = ky * Zc + oy ,  (Wc, Zc) = swap ? (Y+0.5, X+0.5) : (X+0.5, Y+0.5)
enum : int32_t { OPF_FILL = 1, OPF_SWAP = 2, OPF_OPAQUE = 4, OPF_IDENTITY = 8, OPF_HOLE = 16, OPF_FLIPX = 32, OPF_FLIPY = 64, OPF_UNIT_SWAP = 128 };
// OPF_UNIT_SWAP: a quarter turn at unit scale with integer offsets (a pure transposition-type index remap).
// OPF_IDENTITY: unit scale on both axes with integer offsets and no quarter turn, so every canvas pixel maps to exactly
// one source pixel: ix = X + ox (or ox - 1 - X with OPF_FLIPX), iy likewise.  EXIF 2/3/4 at 1:1 are such draws.

struct alignas(16) DevOp {
  double kx, ox, ky, oy;
  int32_t image;            // index into the launch's source table (-1 for fills)
  int32_t flags;
  int32_t cx0, cy0, cx1, cy1;   // inclusive clamp bounds in the source bitmap
  uint32_t rgba;            // fill colour, packed little-endian R,G,B,A
  int32_t X0, Y0, X1, Y1;   // canvas pixels the draw touches (clipped); host-side use
  int32_t pad;
  double xl, xh, yl, yh;    // continuous canvas-space extent of the destination rectangle (edge anti-aliasing)
  int32_t IX0, IY0, IX1, IY1;   // pixels covered COMPLETELY (= X0..Y1 when edge AA is off); host-side use
};

// How a cell (a canvas rectangle whose paint stack is constant) is rendered
enum : int32_t {
  PATH_FILL = 0,      // constant colour
  PATH_COPY = 1,      // opaque constant under ONE 1:1 draw with integer offset: HBM copy (+ source-over if alpha<255)
  PATH_SAMPLE = 2,    // opaque constant under ONE axis-aligned draw, source x driven by canvas x
  PATH_SAMPLE_LDS = 4, // PATH_SAMPLE (bilinear, moderate scale) with the tile's source footprint staged in LDS
  PATH_SAMPLE_STREAM = 6, // PATH_SAMPLE (bilinear): every wave streams its own rows' source row pairs through a private LDS ring
  PATH_AREA_STREAM = 7, // ONE axis-aligned draw that shrinks, IST_FILTER_AREA: box sums per output row, streamed (no barrier)
  PATH_SWAP_LDS = 5,   // ONE quarter-turned draw (EXIF 5-8), bilinear: footprint staged TRANSPOSED in LDS
  PATH_GENERAL = 3    // anything else: paint stack evaluated per pixel in canvas order (swap draws, overlaps,
                      // translucent canvas)
};

struct alignas(16) DevCell {
  int32_t X0, Y0, X1, Y1;
  int32_t path;
  int32_t op;               // the single draw for COPY/SAMPLE; first stack entry otherwise
  int32_t stack_off, stack_len;
  uint32_t bg;              // packed colour under the stack (fill colour or the canvas clear colour)
  int32_t tile_w, tile_h;
  int32_t tiles_x;
  int32_t band_x;           // tiles of this band's earlier cells in one tile row (prefix of tiles_x inside the band)
  int32_t sub_h;            // SAMPLE_LDS: rows per pipeline stage (tile_h = sub_h * stages); SAMPLE_STREAM: ring depth; 0 otherwise
  int64_t tile_begin;       // host-side bookkeeping
};

// A BAND = consecutive cells that share [Y0,Y1) and tile_h.  Tiles are enumerated band by band and, inside a band,
// canvas-row-major ACROSS its cells, so the workgroups resident at one time cover whole canvas rows (contiguous
// destination bytes) even when the strip is horizontal and every cell is a narrow column.
struct alignas(16) DevBand {
  int64_t tile_begin;       // prefix sum of tiles over bands
  int32_t first_cell, n_cells;
  int32_t tiles_per_row;    // sum of tiles_x over the band's cells
  int32_t pad[3];
};

// One entry per tile, in launch order: what a workgroup needs to start (one 16-byte scalar load instead of two binary
// searches over the band / cell prefix tables).  Built for jobs that resample and have at most kMaxTileTable tiles;
// fill/copy-only jobs and gigapixel canvases search the prefix tables instead.
struct alignas(16) DevTile { int32_t cell, op, X0, Y0; };
constexpr int64_t kMaxTileTable = 4 << 20;

struct Compiled {
  int64_t canvas_w = 0, canvas_h = 0;
  int64_t rx0 = 0, ry0 = 0, rx1 = 0, ry1 = 0;   // rendered region (the clip, or the whole canvas)
  int filter = IST_FILTER_BILINEAR;
  int32_t lds_words = 0;               // dynamic LDS per workgroup (32-bit words): max(SWAP_LDS patch, lds_half)
  int32_t lds_half = 0;                // the SAMPLE_LDS footprint buffer (the largest footprint any stage needs)
  int32_t kernel_kind = 0;             // 0: fill/copy cells only; 1: + SAMPLE / SAMPLE_LDS; 2: + SWAP_LDS / GENERAL
  std::vector<DevOp> ops;
  std::vector<DevCell> cells;
  std::vector<DevBand> bands;
  std::vector<DevTile> tiles;          // empty when the job has more than kMaxTileTable tiles
  std::vector<int32_t> stacks;
  std::vector<int32_t> img_w, img_h;   // bitmap sizes per image (for launch-time validation)
  ist_job_info info{};
};

// The same job walked as rows of kFlatPitch bytes (ist_compile.cpp, compile_flat_twin): when every op of a job covers whole canvas rows at
// unit scale, the launch moves contiguous byte ranges as long as the caller's rows are dense (pitch = 4 * width on both sides), and the
// rows the kernel walks need not be the image's.  A workgroup's eight 1 KiB stores are cheapest when their addresses are congruent mod
// 16 KiB; 32 KiB rows measured best (LAB_NOTES.md section 1.8, tools/exp/hbm_ceiling.cpp, tools/exp_flat.py).
constexpr size_t kFlatPitch = 32768;
struct FlatTwin {
  Compiled host;                         // the op list re-expressed on a canvas kFlatPitch / 4 pixels wide
  struct Src { int32_t image; int64_t delta; };   // virtual image k = the caller's image `image`, base moved by delta bytes
  std::vector<Src> src;
  int64_t dst_delta = 0;                 // the wide canvas starts at the rendered region's first row: dst + ry0 * 4 * canvas_w
};
// nothing when the job does not qualify (ist_compile.cpp)
std::unique_ptr<FlatTwin> compile_flat_twin(int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4], const ist_op* ops, int n_ops,
                                            const ist_image_desc* images, int n_images, int filter, const Compiled& primary);

// every device allocation of the library goes through these two (counted: ist_debug_device_allocs).  dev_malloc returns
// the hipError_t as an int (0 = hipSuccess) so that this header needs nothing of HIP.
int dev_malloc(void** p, size_t bytes);
void dev_free(void* p);

// true when the process was started with IST_TUNING=1 (decided once): only then are tuning knobs read from the environment
bool tuning_mode();
// Host-side marks of ONE call (IST_TUNING=1 IST_TIMELINE=1 processes only; no-ops otherwise): tl_begin() starts the call's clock,
// tl_mark() notes a point (microseconds from the start, thread-local), tl_end() prints the list to stderr - only when the call took
// at least IST_TIMELINE_SLOW_MS milliseconds (default 0: every call), so that a run of hundreds of calls names what the RARE slow
// call waited for without printing the others.
void tl_begin();
bool tl_active();                         // a call's clock is running on this thread
void tl_mark(const char* what, long a = -1);
void tl_end(const char* what);

// resolve + cell decomposition (host, pure CPU).  Returns IST_OK or an error code (g_last_error set).
int compile_ops(int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4], const ist_op* ops, int n_ops,
                const ist_image_desc* images, int n_images, int filter, const ist_region* clip, Compiled* out);

// resolve one op (exposed for tests): returns 0 ok, 1 nothing drawn, <0 error
int resolve_op(const ist_op& op, int64_t canvas_w, int64_t canvas_h, int img_w, int img_h, DevOp* out, bool edge_aa = false);

// PNG export, compressing form (ist_png_deflate.hip); ist_png_encode_device picks it when the context's level is > 0
int64_t png_deflate_bound(int64_t w, int64_t h);
// host_out (pinned, out_cap bytes) + aux stream, both optional: the file also lands in host memory, slab by slab, while
// later slabs are still being compressed
// need_rows (optional): called with (y, s) before work that reads canvas rows [0, y) is submitted to stream s (`stream` or
// `stream2`) - a producer that renders the canvas band by band submits the missing bands there / orders s behind them.
// slab_rows_hint (optional): canvas rows a slab should cover (the producer's band height), so that slab boundaries fall on
// band boundaries.  stream2 (optional, with host_out): slabs alternate between the two streams.
int png_encode_device_deflate(ist_ctx* ctx, const void* canvas, size_t pitch, int64_t w, int64_t h, void* out, int64_t out_cap,
                              int64_t* out_len, void* stream, uint8_t* host_out, void* aux,
                              const std::function<int(int64_t, void*)>& need_rows = nullptr, int64_t slab_rows_hint = 0, void* stream2 = nullptr);
int ctx_png_level(const ist_ctx* ctx);
int ctx_png_scratch(ist_ctx* ctx, size_t need, void** p);   // the context's grow-only PNG scratch (kept across calls)

}  // namespace ist

#endif  // IST_INTERNAL_H_
