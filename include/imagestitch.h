/*
 * imagestitch.h — C-ABI of the MI355X-native strip stitcher (libimagestitch.so).
 *
 * This is the drop-in boundary for ONE path of Iamctb/ImageStitching: the Canvas-2D strip concatenation in
 * Page.onStitch (plan -> per-image resample -> row/column blit into one buffer -> readback).  The reference has
 * no FFI of its own (it is a WeChat mini-program: JavaScript calling the platform Canvas); the seam is the set
 * of Canvas calls onStitch issues.  Every entry point below names the reference code it replaces; paths are
 * relative to miniprogram-stitch/miniprogram/ in the reference repository.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures (a stream is passed as void* = hipStream_t);
 *   - return 0 on success, a negative IST_E_* code on failure; ist_last_error() gives the thread-local message
 *     (the reference throws Error(msg) into one catch that toasts '拼图失败：'+msg, pages/index/index.js:1618-1624);
 *   - the caller owns every pixel buffer; the library owns only what it returns through ist_*_create /
 *     ist_plan_compute and frees through the matching destroy/free call;
 *   - pixels are RGBA8, row-major, straight (non-premultiplied) alpha, top row first, as ImageData is;
 *   - no CPU fallback exists: rendering entry points fail with IST_E_NO_DEVICE when no HIP device is present.
 */
#ifndef IMAGESTITCH_H_
#define IMAGESTITCH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IST_API __attribute__((visibility("default")))

#define IST_ABI_VERSION 2     /* 2: output capacity on the decode calls, pooled pinned results, device groups */

/* error codes */
enum {
  IST_OK = 0,
  IST_NOTHING_TO_DO = 1,        /* onStitch returns early when images is empty (pages/index/index.js:1189) */
  IST_E_INVALID = -1,           /* bad argument */
  IST_E_SIZE_UNAVAILABLE = -2,  /* '图片尺寸不可用'   (pages/index/index.js:1254) */
  IST_E_OUTPUT_SIZE = -3,       /* '输出尺寸计算失败' (pages/index/index.js:1320) */
  IST_E_NO_CONTEXT = -4,        /* '无法获取绘图上下文' (pages/index/index.js:1412) */
  IST_E_NO_DEVICE = -5,         /* 'OffscreenCanvas 不可用' analogue (utils/canvas.js:149): no HIP device / HIP failure */
  IST_E_DECODE = -6,            /* '图片N解码异常' (pages/index/index.js:1513): a source bitmap is missing or 0-sized */
  IST_E_UNSUPPORTED = -7,       /* a Canvas feature outside the path (non axis-aligned transform, translucent fill) */
  IST_E_NOMEM = -8,
  IST_E_HIP = -9
};

enum { IST_VERTICAL = 0, IST_HORIZONTAL = 1 };                    /* data.direction  (index.js:16)    */
enum { IST_MODE_MIN = 0, IST_MODE_MAX = 1, IST_MODE_ORIGINAL = 2 };/* data.*StitchMode (index.js:19-20) */
enum { IST_PLATFORM_OTHER = 0, IST_PLATFORM_IOS = 1, IST_PLATFORM_ANDROID = 2 }; /* sys.platform       */
enum { IST_OP_FILL = 0, IST_OP_DRAW = 1, IST_OP_HOLE = 2 };
enum { IST_FILTER_NEAREST = 0, IST_FILTER_BILINEAR = 1, IST_FILTER_AREA = 2 };   /* imageSmoothingEnabled false / true (index.js:1416-1418) */
/* IST_FILTER_AREA (an option; the contract's default for imageSmoothingQuality = 'high', index.js:1419, stays bilinear): on
 * every source axis that is MINIFIED (|scale| > 1 source pixel per canvas pixel) the sample is the average of the source
 * over the canvas pixel's footprint (a box of that width, pixels weighted by overlap); other axes, and any draw that
 * does not shrink, are bilinear - at |scale| = 1 the box IS the bilinear pair, so the two meet continuously.  The
 * phone-capped plans shrink 12 MP photos 2.2x (iOS) to 6.6x (Android), where point-sampled bilinear aliases. */
/* OR-ed into a `filter` argument: anti-alias FRACTIONAL rectangle edges by area coverage, as Canvas rasters do (they
 * arise from ctx.scale(superSample), index.js:1426-1428, and from the unrounded cursor, :1432).  Off: a pixel belongs
 * to a draw iff its centre is inside the rectangle.  Integer-edged plans are unaffected either way. */
enum { IST_FILTER_EDGE_AA = 0x100 };

/* per-image record: the five fields the planner reads (index.js:724-739, 1194, 1211, 1252-1253, 1522-1523, 1532) */
typedef struct ist_image_desc {
  int32_t width, height;      /* naturalWidth, naturalHeight */
  int32_t orientation;        /* EXIF 1..8; 0 = unset (treated as 1, utils/canvas.js:155) */
  int32_t bmp_width, bmp_height; /* decoded bitmap size (bmp.width/height); 0 = same as natural */
  int32_t opaque;             /* caller's hint: every alpha byte is 255 (JPEG-decoded photos). Never changes results. */
  int64_t file_size;          /* bytes; feeds bigTask (index.js:1211-1212); 0 if unknown */
} ist_image_desc;

/* device caps: this.deviceMaxCanvasSize / deviceMaxCanvasPixels + sys.platform (index.js:126-156, 1323-1336) */
typedef struct ist_limits {
  int32_t platform;
  int32_t reserved;
  double max_side;            /* 0 = unset -> reference fallback (android 4096, else 12288) */
  double max_pixels;          /* 0 = unset -> reference fallback */
  double max_super_sample;    /* <=0: reference rule (bigTask 1, ios 2.2, else 2.6; index.js:1363); >0 replaces it */
} ist_limits;

/* one drawWithOrientation call (utils/canvas.js:153): destination rectangle in user space */
typedef struct ist_rect {
  int32_t image, orientation;
  double dx, dy, dw, dh;
} ist_rect;

/* result of the planner: index.js stage 2 (1251-1386) + the rect/cursor loop (1432-1433, 1522-1554) */
typedef struct ist_plan {
  double out_w, out_h;        /* targetW/H after caps (1360-1361) */
  double scale_down;          /* 1337-1357 */
  double super_sample;        /* 1360-1386 */
  int64_t canvas_w, canvas_h; /* canvasOutW/H: offscreen canvas + export size (1373-1383, 1391, 1577-1579) */
  int32_t big_task;           /* 1212 */
  int32_t n_rects;
  ist_rect* rects;            /* library-owned; ist_plan_free */
} ist_plan;

/* One recorded Canvas call, in canvas order.  kind 0: fillRect(d) with fillStyle rgba (index.js:1423-1424);
 * kind 1: 9-argument drawImage(image, s, d) (utils/canvas.js:156); kind 2 (no Canvas analogue): rectangle d is
 * left untouched by the launch because another producer delivers those pixels (a band received in place over
 * xGMI in the multi-GPU layout).  m = CTM at the time of the call:
 * X = m0*u + m2*v + m4,  Y = m1*u + m3*v + m5  (same order as ctx.setTransform(a,b,c,d,e,f), index.js:1404). */
typedef struct ist_op {
  int32_t kind, image;
  double m[6];
  double s[4];                /* sx, sy, sw, sh */
  double d[4];                /* dx, dy, dw, dh */
  uint8_t rgba[4];
  int32_t reserved;
} ist_op;

typedef struct ist_region { int32_t x, y, w, h; } ist_region;    /* getImageData / export region (index.js:1564, 1577) */

typedef struct ist_job_info {
  int64_t canvas_w, canvas_h;
  int32_t n_ops, n_cells;
  int64_t n_tiles;
  int64_t out_pixels;         /* pixels written per launch */
  int64_t src_pixels_touched; /* distinct source pixels inside the sampling footprints */
  int64_t algorithmic_bytes;  /* 4*src_pixels_touched + 4*out_pixels */
  int64_t tiles_fill, tiles_copy, tiles_sample, tiles_general;
} ist_job_info;

typedef struct ist_ctx ist_ctx;   /* one HIP device + scratch */
typedef struct ist_job ist_job;   /* one compiled op list (device-side cell/op tables) */

/* ---- diagnostics ------------------------------------------------------------------------------------------ */
IST_API int ist_abi_version(void);
IST_API const char* ist_last_error(void);
IST_API int ist_device_count(void);                                   /* 0 when no HIP device is usable */
/* device allocations (hipMalloc calls) the library has made in this process so far.  A measurement aid for hosts and tests:
 * the steady state of every entry point allocates nothing (scratch, arenas and table blocks are kept and re-used). */
IST_API int64_t ist_debug_device_allocs(void);
/* JPEG files whose entropy-coded scan the GPU Huffman decoder has decoded AND validated in this process so far (files it
 * handed back to the host decoder do not count).  Tests use it to tell the GPU path from the silent host fall-back. */
IST_API int64_t ist_debug_gpu_entropy_files(void);
/* images the file pipeline has reconstructed STRAIGHT INTO the canvas so far (a draw that only moves an opaque image: no bitmap of
 * its own, no stitch launch for it).  Tests use it to tell that path from the general one, which makes the same pixels. */
IST_API int64_t ist_debug_direct_images(void);
/* diagnostics: ist_group_stitch_rgba8 / ist_stitch_rgba8_multi calls of this process whose result was delivered by the HOST
 * SINK (every device DMAs its bands into the pinned result; no gather, no root readback) */
IST_API int64_t ist_debug_host_sink_stitches(void);
/* launches (ist_job_launch, any caller) that took a job's flat form: every op covers whole canvas rows at unit scale and the caller's
 * rows were dense on both sides, so the same bytes were moved as rows of 32 KiB */
IST_API int64_t ist_debug_flat_launches(void);
/* ist_stitch_rgba8 calls delivered band by band with uploads and downloads overlapped (a strip of disjoint row bands >= 32 MB; DESIGN.md
 * section 4 "Host <-> device"); the others took upload-all, launch, download-all */
IST_API int64_t ist_debug_duplex_stitches(void);

/* ---- planner: pure CPU, bit-exact to index.js:1211-1216, 1251-1386, 1432-1433, 1522-1554 -------------------- */
IST_API void ist_limits_default(int platform, ist_limits* out);        /* index.js:126-156 fallback branch */
IST_API void ist_limits_unlimited(ist_limits* out);                    /* MI355X default: caps lifted, superSample 1 */
IST_API int ist_plan_compute(const ist_image_desc* images, int n_images, int direction, int mode, double gap,
                             const ist_limits* limits, ist_plan* out);
IST_API void ist_plan_free(ist_plan* plan);
/* the Canvas call sequence stage 3-4 issues for a plan: white fillRect over the canvas (index.js:1423-1424),
 * ctx.scale(ss,ss) folded into every CTM (1426-1428), one drawWithOrientation per rect (utils/canvas.js:153-202).
 * ops must hold n_rects + 1 entries. */
IST_API int ist_plan_ops(const ist_plan* plan, const ist_image_desc* images, int n_images, ist_op* ops, int* n_ops);
/* canvas pixels one op touches: box = {X0, Y0, X1, Y1} (half open, clipped to the canvas) under the coverage rule of
 * `filter` (pixel centre, or every touched pixel with IST_FILTER_EDGE_AA).  Returns 1 when the op draws nothing.
 * Pure CPU; what the multi-GPU layer uses to cut a stitch into per-image bands. */
IST_API int ist_op_box(const ist_op* op, int64_t canvas_w, int64_t canvas_h, int filter, int32_t box[4]);
/* The flat form of an op list as the kernel will walk it (pure CPU, no GPU needed; a test aid): when every op covers whole canvas
 * rows at unit scale, ist_job_launch on dense rows moves the same bytes as rows of *pitch bytes starting *dst_offset bytes into the
 * destination.  One record per cell: rows Y0..Y1-1, pixels X0..X1-1 of that wide canvas; path 0 = fill with bg (packed R,G,B,A),
 * 1 = copy from src[image] starting src_offset bytes into it (+ *pitch per row), source-over bg unless opaque.  *n_cells = 0 when
 * the op list has no flat form.  tests/test_flat_form.py replays the records in numpy against the op list. */
typedef struct ist_flat_cell {
  int32_t path, image;
  int32_t X0, Y0, X1, Y1;
  int64_t src_offset;
  uint32_t bg;
  int32_t opaque;
} ist_flat_cell;
IST_API int ist_debug_flat_form(int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4], const ist_op* ops, int n_ops,
                                const ist_image_desc* images, int n_images, int filter, const ist_region* clip, int64_t* pitch,
                                int64_t* dst_offset, ist_flat_cell* cells, int max_cells, int* n_cells);

/* ---- sharding: one stitch cut into parts for a group of GPUs (pure CPU) ----------------------------------------- */
/* The per-image iterations of onStitch are independent once the cursor is planned (index.js:1439-1554).  A PART is a
 * canvas box of ONE draw, rendered by one GPU (its slot) over the background alone; the root (slot 0) assembles them.
 * IST_SPLIT_IMAGE: image i -> slot i mod n_slots (BASELINE configs[3]).  IST_SPLIT_BAND: canvas rows dealt in canvas
 * order so that every slot renders the same number of output pixels (9 images on 8 GPUs: 1.125 images each); cuts fall
 * on multiples of 8 rows inside a draw's box.  A slot needs only source rows [sy0, sy1) of the part's image (readable
 * for 16 bytes past the last row when a partial buffer is passed, biased by -sy0 rows, to ist_job_launch).
 * Draws that overlap (edge anti-aliasing makes neighbours share a pixel row) cannot be sharded draw by draw: IST_E_UNSUPPORTED.
 * IST_SPLIT_ROWS: slot s owns canvas rows [cuts[s], cuts[s+1]) (ist_shard_row_cuts) ACROSS ALL DRAWS and renders the whole op
 * list clipped to them.  The unit that is rendered and delivered is the slot's BAND - full canvas width whatever the layout,
 * so it is a contiguous byte range of the canvas for horizontal strips (index.js:1540-1553: every rect spans the canvas
 * height, i.e. is a column band under the two cuts above) and centred 'original' rects too: received in place, no staging,
 * no placement launch, host sink always available; overlapping draws and anti-aliased seams are allowed (one owner per
 * pixel paints the whole stack).  Its parts are (slot's rows) x (one draw's box), slot by slot in op order, and say which
 * rows of which image the slot must hold (a horizontal strip on 8 slots: 1/8 of the rows of EVERY image per slot - still
 * disjoint input subsets); in_place is 1 for all of them (it describes the band).
 * IST_SPLIT_AUTO: IMAGE when that cut yields full-width parts only (vertical min / max strips: BASELINE configs[3]),
 * otherwise ROWS; ist_shard_resolve says which. */
enum { IST_SPLIT_IMAGE = 0, IST_SPLIT_BAND = 1, IST_SPLIT_ROWS = 2, IST_SPLIT_AUTO = 3 };
typedef struct ist_part {
  int32_t image, op;          /* source image; index of the draw in the op list */
  int32_t slot;               /* owner, 0 .. n_slots-1; slot 0 is the root */
  int32_t X0, Y0, X1, Y1;     /* canvas box, half open */
  int32_t sx0, sy0, sx1, sy1; /* source columns / rows the part samples, half open */
  int32_t in_place;           /* the box spans the canvas width: a contiguous byte range of the canvas */
} ist_part;
/* max_parts: n_ops + n_slots + 8 suffices for IMAGE / BAND, n_ops * n_slots for ROWS / AUTO */
IST_API int ist_shard_parts(const ist_op* ops, int n_ops, int64_t canvas_w, int64_t canvas_h, const ist_image_desc* images,
                            int n_images, int filter, int n_slots, int split, ist_part* parts, int max_parts, int* n_parts);
/* IST_SPLIT_ROWS: cuts[0 .. n_slots] - equal rows per slot, every cut on a multiple of 8 rows, cuts[n_slots] = canvas_h; a
 * canvas shorter than 8 * n_slots rows leaves some slots empty (cuts[s] == cuts[s+1]; never slot 0) */
IST_API int ist_shard_row_cuts(int64_t canvas_h, int n_slots, int32_t* cuts);
/* the split IST_SPLIT_AUTO stands for on this op list (other values are returned unchanged); negative on error */
IST_API int ist_shard_resolve(const ist_op* ops, int n_ops, int64_t canvas_w, int64_t canvas_h, const ist_image_desc* images,
                              int n_images, int filter, int split);

/* ---- device path: inputs and output already resident in HBM ------------------------------------------------- */
IST_API ist_ctx* ist_ctx_create(int device);
IST_API void ist_ctx_destroy(ist_ctx* ctx);
/* waits for everything the context itself has in flight (its streams and staging lanes).  Hosts call it before process exit so
 * that no DMA of the library is still pending when the runtime (or a profiler attached to it) shuts down. */
IST_API int ist_ctx_sync(ist_ctx* ctx);
/* PNG export form for every *_png entry point of this context.  1 (default): Paeth filter + run-length matches + a
 * dynamic Huffman code per 16 KiB, all on the GPU: photographs shrink to about 0.43 x raw (smaller than zlib level 6 on
 * the same filtered stream), flat areas (gaps, screenshots) to 2-3 per cent; random data stays 1:1.  0: stored deflate
 * blocks, file = 1.001 x raw, one HBM-bound pass (3x faster).  Both decode to the same pixels
 * (canvasToTempFilePath quality:1 is lossless, utils/canvas.js:205-242). */
IST_API int ist_ctx_set_png_level(ist_ctx* ctx, int level);
/* compile an op list for a canvas (replaces createOffscreenCanvas + the recorded draw calls; utils/canvas.js:131,
 * index.js:1391-1428, 1532-1551).  clear_rgba = canvas initial colour ({0,0,0,0} for a fresh canvas).
 * clip = NULL renders the whole canvas; otherwise only that region is written (getImageData(0,0,1,1), 1564). */
IST_API ist_job* ist_job_create(ist_ctx* ctx, int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4],
                                const ist_op* ops, int n_ops, const ist_image_desc* images, int n_images,
                                int filter, const ist_region* clip);
IST_API int ist_job_info_get(const ist_job* job, ist_job_info* out);
/* The destination row pitch (bytes) this job runs fastest on, for a host that allocates the canvas itself: 4 * canvas_w (dense rows)
 * when the job has a flat form - a strip of whole rows at unit scale, which ist_job_launch then walks as rows of 32 KiB - otherwise
 * 4 * canvas_w rounded up to a multiple of 4096 (measured: INTEGRATION.md "Row pitch").  Any pitch >= 4 * canvas_w that is a multiple
 * of 4 stays legal.  0 on a NULL job. */
IST_API size_t ist_job_preferred_dst_pitch(const ist_job* job);
/* one fused launch: every canvas pixel (in clip) is written exactly once.  src[i] / dst are DEVICE pointers.
 * stream = hipStream_t (NULL = default stream).  Asynchronous: returns after enqueueing. */
IST_API int ist_job_launch(ist_job* job, const void* const* src, const size_t* src_pitch, int n_images,
                           void* dst, size_t dst_pitch, void* stream);
/* waits for the streams the job was launched on (not for the device: other streams keep running), then recycles its tables.
 * A job launched on the legacy default stream (NULL) inherits that stream's own implicit synchronisation rules. */
IST_API void ist_job_destroy(ist_job* job);

/* ---- host path: what stitch(images, direction, opts) binds (host RGBA8 in, host RGBA8 out) ------------------- */
/* Transfers of this group: the library never page-locks or registers memory it does not own.  Caller buffers (pageable,
 * any pitch) are packed through a ring of pinned chunks by a few host threads; buffers the library RETURNS are pinned
 * blocks of a process-wide pool, filled by one DMA, and go back to the pool through ist_free (ist_pool_trim releases the
 * idle ones).
 * index.js:1251-1581 minus decode (1441-1520) and PNG encode (1579): plan, then render.
 * *out_pixels is owned by the library (canvas_w*canvas_h*4 bytes, pitch canvas_w*4); release with ist_free. */
IST_API int ist_stitch_rgba8(ist_ctx* ctx, const ist_image_desc* images, const uint8_t* const* src,
                             const size_t* src_pitch, int n_images, int direction, int mode, double gap,
                             const ist_limits* limits, int filter, ist_plan* out_plan, uint8_t** out_pixels);
/* render a recorded Canvas op list into a caller buffer (the Canvas-2D shim's export / getImageData) */
IST_API int ist_render_rgba8(ist_ctx* ctx, int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4],
                             const ist_op* ops, int n_ops, const ist_image_desc* images,
                             const uint8_t* const* src, const size_t* src_pitch, int n_images, int filter,
                             const ist_region* region, uint8_t* dst, size_t dst_pitch);
IST_API void ist_free(void* p);                /* any buffer the library returned through an out pointer */
IST_API void ist_pool_trim(void);              /* release the idle pinned blocks ist_free is keeping for reuse */

/* ---- device groups: one stitch on several GPUs from ONE process (stitch(images, direction, {devices})) ------ */
/* onStitchVertical/Horizontal -> onStitch (index.js:771-788 -> :1186) with the per-image iterations (:1439-1554) dealt to
 * the GPUs of `devices` (devices[0] = the root; a device may be listed more than once).  The job is cut by ist_shard_parts;
 * parts render on their devices, finished bands reach the root's canvas through ONE grouped ncclSend/ncclRecv batch
 * (RCCL over xGMI; librccl.so.1 is loaded on first use of a group with more than one device): full-width bands are
 * received in place, others are staged and placed by a 1:1 launch behind their receive.  Parts whose device is the
 * root's render straight into the canvas.  The one-process-per-GPU form of the same layout is imagestitching_amd/dist.py. */
typedef struct ist_group ist_group;
typedef struct ist_group_job ist_group_job;
IST_API ist_group* ist_group_create(const int* devices, int ndev);
IST_API void ist_group_destroy(ist_group* g);
IST_API int ist_group_slots(const ist_group* g);
IST_API int ist_group_device(const ist_group* g, int slot);            /* -1 when slot is out of range */
/* device-resident: compile once, launch on new buffers.  split = IST_SPLIT_IMAGE / IST_SPLIT_BAND. */
IST_API ist_group_job* ist_group_job_create(ist_group* g, int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4],
                                            const ist_op* ops, int n_ops, const ist_image_desc* images, int n_images, int filter,
                                            int split);
IST_API void ist_group_job_destroy(ist_group_job* job);
/* the part table of the job (parts == NULL: only the count); part k's owner device = ist_group_device(g, parts[k].slot) */
IST_API int ist_group_job_parts(const ist_group_job* job, ist_part* parts, int max_parts, int* n_parts);
/* src[k] / src_pitch[k] belong to PART k: the address, on the part's device, of row 0 of the part's image (a holder of rows
 * [sy0, sy1) only passes the address of row sy0 minus sy0 * pitch; 16 bytes behind its last row must be readable).
 * dst: the canvas on the root's device, dst_pitch == canvas_w * 4.  Asynchronous: ist_group_sync waits for the canvas.
 * The group runs on its own streams: whatever the caller queued on these buffers must have completed before the call. */
IST_API int ist_group_job_launch(ist_group_job* job, const void* const* src, const size_t* src_pitch, int n_parts, void* dst,
                                 size_t dst_pitch);
IST_API int ist_group_sync(ist_group* g);
/* host buffers in, host buffer out (ist_stitch_rgba8 on a group): every device uploads only the source rows its parts
 * sample, over its own PCIe link; *out_pixels is library-owned (ist_free). */
IST_API int ist_group_stitch_rgba8(ist_group* g, const ist_image_desc* images, const uint8_t* const* src, const size_t* src_pitch,
                                   int n_images, int direction, int mode, double gap, const ist_limits* limits, int filter,
                                   int split, ist_plan* out_plan, uint8_t** out_pixels);
/* the same in one call; groups are cached per device list for the life of the process */
IST_API int ist_stitch_rgba8_multi(const int* devices, int ndev, const ist_image_desc* images, const uint8_t* const* src,
                                   const size_t* src_pitch, int n_images, int direction, int mode, double gap,
                                   const ist_limits* limits, int filter, int split, ist_plan* out_plan, uint8_t** out_pixels);

/* ---- decode: PNG file -> RGBA8 (host; the Image.src step, utils/canvas.js:27-121, for 'png' inputs, index.js:4) ---- */
/* colour types 0/2/3/4/6, bit depths 1-16 (16-bit keeps the high byte), tRNS, plain or Adam7-interlaced.  JPEG / WebP /
 * HEIC return IST_E_UNSUPPORTED; damaged files IST_E_DECODE ('图片N解码异常' analogue).
 * Every decode entry point takes the CAPACITY of the output buffer (out_pitch bytes per row, out_rows rows) and fails
 * with IST_E_INVALID when the file's own header asks for more: the header is untrusted input. */
IST_API int ist_png_info(const uint8_t* file, int64_t len, int32_t* width, int32_t* height);
IST_API int ist_png_decode_rgba8(const uint8_t* file, int64_t len, uint8_t* out, size_t out_pitch, int64_t out_rows);
/* JPEG (baseline / extended sequential / progressive, 8 bit, grey or YCbCr 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0, restart
 * intervals): Huffman decoding on the host, dequantise + IDCT + upsampling + colour conversion on the GPU.  *orientation = EXIF tag 0x0112
 * (0 when absent) - what getImageInfo feeds the planner (index.js:734).  Lossless / arithmetic-coded JPEG: IST_E_UNSUPPORTED. */
IST_API int ist_jpeg_info(const uint8_t* file, int64_t len, int32_t* width, int32_t* height, int32_t* orientation);
IST_API int ist_jpeg_decode_rgba8(ist_ctx* ctx, const uint8_t* file, int64_t len, uint8_t* out, size_t out_pitch, int64_t out_rows);
/* by signature: PNG, JPEG, BMP (uncompressed 1-32 bit, bit fields, RLE8 / RLE4), GIF (first frame) or WebP (lossless VP8L and lossy
 * VP8 key frames, with or without an ALPH chunk; orientation from the container's EXIF chunk; animation: the first frame) - every
 * raster member of SUPPORTED_IMAGE_TYPES (index.js:4).  ctx may be NULL for everything except JPEG (whose reconstruction
 * runs on the GPU). */
IST_API int ist_image_info(const uint8_t* file, int64_t len, int32_t* width, int32_t* height, int32_t* orientation);
IST_API int ist_image_decode_rgba8(ist_ctx* ctx, const uint8_t* file, int64_t len, uint8_t* out, size_t out_pitch, int64_t out_rows);

/* files -> decoded bitmaps in CALLER-OWNED DEVICE memory (the Image.src step ending in HBM): baseline JPEG entropy decoding
 * (one interleaved scan, at most two DC + two AC tables, with or without restart intervals) and reconstruction on the GPU, progressive JPEG / PNG / BMP / GIF entropy stages on host threads.  dst[i] must hold
 * dst_rows[i] rows of dst_pitch[i] bytes (sizes from ist_image_info); out_descs (optional) receives what the planner needs
 * (size, EXIF orientation, opaque).  Returns when the bitmaps are complete.
 * Ordering: the library writes dst[i] from its OWN streams.  Whatever the caller has queued on those buffers (a launch
 * that still reads the previous contents, a fill) must have completed before the call - the library cannot see the
 * caller's streams.  The Python host synchronises the tensors' current torch stream before it calls. */
IST_API int ist_decode_files_device(ist_ctx* ctx, const uint8_t* const* files, const int64_t* lens, int n_images,
                                    void* const* dst, const size_t* dst_pitch, const int64_t* dst_rows, ist_image_desc* out_descs);
/* phase times of the last ist_stitch_files_png / ist_decode_files_device on this context, in milliseconds (measurement
 * aid: while on, every phase ends with a stream synchronisation) */
enum { IST_PHASE_HOST_DECODE = 0, IST_PHASE_PLAN_ARENA = 1, IST_PHASE_ENTROPY_GPU = 2, IST_PHASE_RECONSTRUCT = 3, IST_PHASE_STITCH = 4,
       IST_PHASE_PNG = 5, IST_PHASE_D2H = 6, IST_PHASE_COUNT = 8 };
IST_API int ist_ctx_set_timing(ist_ctx* ctx, int on);
IST_API int ist_ctx_last_timing(ist_ctx* ctx, double* ms, int n);

/* ---- files in, file out: the whole onStitch (decode -> plan -> resample+blit -> PNG export; index.js:1441-1581) ---- */
/* files[i] = the bytes of one image file of any type ist_image_info recognises.  Baseline JPEG: container parse + de-stuffing
 * on a host thread per image, Huffman decoding, reconstruction, stitch and PNG compression on the GPU; progressive JPEG /
 * PNG / BMP / GIF / WebP: entropy stage on that host thread, the rest on the GPU.  Decoded bitmaps, canvas and PNG stay in
 * HBM - only file bytes go in and PNG bytes come out.  A file that does not decode fails with IST_E_DECODE /
 * IST_E_UNSUPPORTED and the message '图片N解码异常: ...' (index.js:1512-1514).
 * files[i] must not change during the call: a file is parsed twice (frame header for the device layout, then the scan); a
 * JPEG whose frame layout differs between the two reads fails with IST_E_DECODE instead of overrunning the layout. */
IST_API int ist_stitch_files_png(ist_ctx* ctx, const uint8_t* const* files, const int64_t* lens, int n_images,
                                 int direction, int mode, double gap, const ist_limits* limits, int filter,
                                 ist_plan* out_plan, uint8_t** out_png, int64_t* out_len);
/* the same from file PATHS (what wx.chooseImage hands the page: tempFilePaths, index.js:1441-1450): the library READS the
 * files (one parked worker per file) into blocks the context keeps from call to call - it does not map them, so a file that
 * another process rewrites or truncates meanwhile can neither change under the parsers nor raise SIGBUS in the host
 * process.  A path that cannot be opened, is not a regular file, is empty, or shrinks while it is read fails with
 * IST_E_DECODE and '图片N解码异常: ...'. */
IST_API int ist_stitch_paths_png(ist_ctx* ctx, const char* const* paths, int n_images,
                                 int direction, int mode, double gap, const ist_limits* limits, int filter,
                                 ist_plan* out_plan, uint8_t** out_png, int64_t* out_len);

/* ---- export: lossless PNG (fileType 'png', quality 1; utils/canvas.js:205-242, index.js:1577-1579) --------------- */
/* upper bound of the file size for a w x h RGBA canvas */
IST_API int64_t ist_png_bound(int64_t w, int64_t h);
/* encode a canvas that is resident in HBM into a device buffer (16-byte aligned, ist_png_bound bytes); one HBM-bound
 * pass; the checksums are combined on the host, so the call synchronises `stream` before it returns */
IST_API int ist_png_encode_device(ist_ctx* ctx, const void* canvas, size_t pitch, int64_t w, int64_t h, void* out,
                                  int64_t out_cap, int64_t* out_len, void* stream);
/* host pixels -> PNG bytes (library-owned, release with ist_free) */
IST_API int ist_png_encode_rgba8(ist_ctx* ctx, const uint8_t* pixels, size_t pitch, int64_t w, int64_t h,
                                 uint8_t** out_png, int64_t* out_len);
/* recorded Canvas op list -> PNG bytes (wx.canvasToTempFilePath of the shim); the canvas never leaves the device */
IST_API int ist_render_png(ist_ctx* ctx, int64_t canvas_w, int64_t canvas_h, const uint8_t clear_rgba[4],
                           const ist_op* ops, int n_ops, const ist_image_desc* images, const uint8_t* const* src,
                           const size_t* src_pitch, int n_images, int filter, uint8_t** out_png, int64_t* out_len);
/* onStitch stages 2-5 including the export: plan, render, PNG */
IST_API int ist_stitch_png(ist_ctx* ctx, const ist_image_desc* images, const uint8_t* const* src,
                           const size_t* src_pitch, int n_images, int direction, int mode, double gap,
                           const ist_limits* limits, int filter, ist_plan* out_plan, uint8_t** out_png,
                           int64_t* out_len);

#ifdef __cplusplus
}
#endif
#endif /* IMAGESTITCH_H_ */
