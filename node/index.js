'use strict';
/**
 * Node host of the MI355X strip stitcher: the reference's stitch surface as one function.
 *
 * Reference: Page.onStitch (miniprogram-stitch/miniprogram/pages/index/index.js:1186-1633) reads
 * this.data.{images, direction, gap, verticalStitchMode, horizontalStitchMode}.  Here:
 *
 *   stitch(images, direction, opts?) -> Promise<{width, height, data: Buffer, plan}>
 *
 * images[i] = {width, height, data: Uint8Array (RGBA8, straight alpha, row-major), orientation?: 1..8, fileSize?, opaque?}
 * direction = 'vertical' | 'horizontal'                         (data.direction, index.js:16)
 * opts      = {mode: 'min'|'max'|'original' (default 'min', index.js:19-20), gap: 0..20 (default 0, index.js:17),
 *              filter: 'bilinear'|'nearest' (imageSmoothingEnabled, index.js:1416), platform: 'ios'|'android'|'devtools'
 *              (reproduces the phone caps, index.js:1323-1336; default: caps lifted, superSample 1),
 *              maxSide, maxPixels (deviceMaxCanvasSize/Pixels), superSample (MAX_SUPER_SAMPLE, index.js:1363),
 *              onProgress: (percent) => void  (the stitchProgress checkpoints of index.js:1193-1611),
 *              edgeAA: anti-alias fractional rectangle edges by area coverage (default: true when `platform` is given - a
 *              reference plan has fractional edges as a rule - false otherwise: pixel-centre rule),
 *              devices: number[] - shard the stitch over these GPUs from this one process (devices[0] = root; parts render
 *              on their GPUs, one grouped RCCL send/recv batch over xGMI gathers the bands into the root's canvas),
 *              split: 'image' (image i -> devices[i mod n], the BASELINE layout) | 'band' (equal output rows per GPU, cut draw by draw)
 *                     | 'rows' (GPU s owns a band of canvas rows across ALL draws: full-width bands for horizontal strips too,
 *                     index.js:1540-1553) | 'auto' (default: 'image' when its parts are full-width, else 'rows')}
 * Errors reject with Error('拼图失败：' + reason) like the reference's catch (index.js:1618-1624); err.code is the
 * C-ABI code.  No pixel arithmetic happens in JavaScript; there is no CPU fallback.
 */
const path = require('path');
const native = require(path.join(__dirname, 'imagestitch.node'));

const DIRECTION = { vertical: 0, horizontal: 1 };
const MODE = { min: 0, max: 1, original: 2 };
const FILTER = { nearest: 0, bilinear: 1, area: 2 };
const PLATFORM = { other: 0, devtools: 0, windows: 0, mac: 0, ios: 1, android: 2 };
const KNOWN = ['mode', 'gap', 'filter', 'platform', 'maxSide', 'maxPixels', 'superSample', 'onProgress', 'edgeAA', 'pngLevel', 'devices', 'split'];
const SPLIT = { image: 0, band: 1, rows: 2, auto: 3 };
const FILTER_EDGE_AA = 0x100;    // IST_FILTER_EDGE_AA: anti-alias fractional rectangle edges by area coverage

function limitsOf(opts) {
  const o = opts || {};
  const lim = {};
  if (o.platform !== undefined && o.platform !== null) {
    if (!(o.platform in PLATFORM)) throw new TypeError('unknown platform ' + o.platform);
    lim.platform = PLATFORM[o.platform];
  }
  if (typeof o.maxSide === 'number') lim.maxSide = o.maxSide;
  if (typeof o.maxPixels === 'number') lim.maxPixels = o.maxPixels;
  if (typeof o.superSample === 'number') lim.superSample = o.superSample;
  return Object.keys(lim).length ? lim : null;
}

// Coverage rule for fractional rectangle edges.  Unset: ON whenever a reference platform's plan is requested (the
// reference's default plans scale the canvas by superSample 2.2 / 2.6 for fewer than 7 images, index.js:1363,1426-1428, and
// keep an unrounded cursor when gap > 0 and scaleDown < 1, :1432, so fractional edges are normal there and a Canvas raster
// anti-aliases them); OFF for the lifted MI355X default (every output pixel owned by exactly one image).
function edgeAA(o) { return (o.edgeAA === undefined || o.edgeAA === null) ? (o.platform !== undefined && o.platform !== null) : !!o.edgeAA; }

function args(images, direction, opts) {
  const o = opts || {};
  for (const k of Object.keys(o)) if (!KNOWN.includes(k)) throw new TypeError('unknown stitch option ' + k);
  if (!(direction in DIRECTION)) throw new TypeError("direction must be 'vertical' or 'horizontal'");
  const mode = o.mode || 'min';                      // `|| 'min'` (index.js:1257)
  if (!(mode in MODE)) throw new TypeError('unknown mode ' + mode);
  const filter = o.filter || 'bilinear';
  if (!(filter in FILTER)) throw new TypeError('unknown filter ' + filter);
  return [images || [], DIRECTION[direction], MODE[mode], Number(o.gap) || 0, limitsOf(o), FILTER[filter] | (edgeAA(o) ? FILTER_EDGE_AA : 0)];
}

// The reference reports progress through setData({stitchProgress}): 1 at the start (index.js:1193), 25 when every image
// is prepared (:1247-1248), 30 after planning (:1358), 30 + 60*(i+1)/n (capped at 90) per drawn image (:1556-1557), 96
// after the export (:1581), 100 at the end (:1611).  Here the draw loop is ONE launch, so the per-image steps collapse
// into 90; opts.onProgress(percent) receives the same checkpoints.
function withProgress(opts, run) {
  const cb = opts && typeof opts.onProgress === 'function' ? opts.onProgress : null;
  if (!cb) return run();
  cb(1); cb(25); cb(30);
  return run().then((r) => { cb(90); cb(96); cb(100); return r; }, (e) => { cb(0); throw e; });       // failure resets to 0 (:1622)
}

// opts.devices -> the trailing (asPng, devices, split) arguments of the native call
function groupArgs(opts) {
  const o = opts || {};
  if (o.devices === undefined || o.devices === null) return [];
  if (!Array.isArray(o.devices) || !o.devices.length || !o.devices.every((d) => Number.isInteger(d) && d >= 0)) throw new TypeError('devices must be a non-empty array of GPU indices');
  const split = o.split || 'auto';
  if (!(split in SPLIT)) throw new TypeError('unknown split ' + split);
  return [false, o.devices, SPLIT[split]];
}
function stitch(images, direction, opts) {
  let a;
  try { a = args(images, direction, opts).concat(groupArgs(opts)); } catch (e) { return Promise.reject(e); }
  if (!a[0].length) return Promise.resolve(null);      // `if (!originalImages.length) return;` (index.js:1189): no progress, no error
  return withProgress(opts, () => native.stitch(...a));
}
function stitchSync(images, direction, opts) { const a = args(images, direction, opts).concat(groupArgs(opts)); return a[0].length ? native.stitchSync(...a) : null; }
/** stitch + the reference's export step: resolves {width, height, png: Buffer (a lossless PNG file), plan}. The canvas
 *  never leaves the GPU; only the PNG bytes cross PCIe (utils/canvas.js:205-242, index.js:1577-1579). */
function stitchPng(images, direction, opts) {
  let a;
  try { a = args(images, direction, opts); } catch (e) { return Promise.reject(e); }
  if (!a[0].length) return Promise.resolve(null);
  return withProgress(opts, () => { pngLevel(opts); return native.stitch(...a, true); });
}
/** opts.pngLevel: 0 = stored deflate blocks (file = raw size, fastest), 1 = Paeth + run-length + Huffman on the GPU
 *  (photographs about half, screenshots a few per cent). A process-wide setting of the native context. */
function pngLevel(opts) { if (opts && opts.pngLevel !== undefined && opts.pngLevel !== null) native.setPngLevel(opts.pngLevel | 0); }
function setPngLevel(level) { native.setPngLevel(level | 0); }
/** PNG file bytes -> {width, height, data} (RGBA8, straight alpha). Host decode: the Image.src step for 'png' inputs
 *  (utils/canvas.js:27-121; SUPPORTED_IMAGE_TYPES, index.js:4). JPEG / WebP / HEIC are not built: err.code '-7'. */
function decodePng(file) { return native.decodePng(file); }
/** PNG or JPEG file bytes -> {width, height, orientation, opaque, data}. JPEG (baseline): Huffman decoding on the host,
 *  IDCT / chroma upsampling / colour conversion on the GPU; orientation = EXIF tag 0x0112 (what getImageInfo feeds the
 *  planner, index.js:734). WebP / HEIC reject with err.code '-7'. */
function decodeImage(file) { return native.decodeImage(file); }
/** File to file (PNG and JPEG inputs): decode -> stitch -> PNG export. Resolves {width, height, png, plan}; writes
 *  outPath when given. A file that does not decode rejects with '图片N解码异常' like index.js:1512-1514. */
async function stitchFiles(paths, direction, opts, outPath) {
  const fs = require('fs');
  const a = args([], direction, opts);
  if (!paths || !paths.length) return null;
  const files = paths.map((p) => fs.readFileSync(p));
  // one native call: Huffman / inflate on host threads, reconstruction + stitch + PNG on the GPU, buffers stay in HBM
  const res = await withProgress(opts, () => { pngLevel(opts); return native.stitchFiles(files, a[1], a[2], a[3], a[4], a[5]); });
  if (res && outPath) fs.writeFileSync(outPath, res.png);
  return res;
}
/** Lossless PNG of RGBA8 pixels, encoded on the GPU. */
function encodePng(data, width, height, opts) { pngLevel(opts); return native.encodePng(data, width, height); }
function plan(images, direction, opts) {
  const a = args(images, direction, opts);
  return native.plan(a[0], a[1], a[2], a[3], a[4]);
}

module.exports = { stitch, stitchSync, stitchPng, stitchFiles, encodePng, setPngLevel, decodePng, decodeImage, plan, native, DIRECTION, MODE, FILTER, PLATFORM, SPLIT };
