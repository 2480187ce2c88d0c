'use strict';
/**
 * Canvas-2D shim (surface S2 of SURVEY.md section 8b): exactly the members the reference's stitch path touches, so that
 * the UNMODIFIED page script (pages/index/index.js + utils/canvas.js) can drive the HIP path.
 *
 *   canvas : createOffscreenCanvas({type,width,height}) (utils/canvas.js:134,138,144), width/height (settable; 0 frees,
 *            index.js:1587-1588), getContext('2d') (index.js:1392), createImage() (utils/canvas.js:38-40)
 *   ctx    : fillStyle / fillRect (index.js:1423-1424), scale (:1427), save / restore / translate / rotate / scale
 *            (utils/canvas.js:154-201), drawImage 3/5/9-arg (utils/canvas.js:156), getImageData (index.js:1564),
 *            imageSmoothingEnabled / imageSmoothingQuality (:1416-1420), setTransform / resetTransform (:1403-1406),
 *            clearRect (:1599, whole-canvas only)
 *   export : wx.canvasToTempFilePath({canvas,x,y,width,height,destWidth,destHeight,fileType,quality})
 *            (utils/canvas.js:211-221) and canvas.toTempFilePath (utils/canvas.js:225-236)
 *
 * drawImage / fillRect only RECORD an op (with the CTM of that moment); an export or getImageData triggers ONE fused
 * launch over the recorded list through native.render().  getImageData renders just the requested region and keeps
 * the list (the reference calls getImageData(0,0,1,1) after every image purely as a flush, index.js:1559-1566).
 * Fractional rectangle edges are anti-aliased by area coverage, as a Canvas raster does (IST_FILTER_EDGE_AA): the page's
 * default plans scale the canvas by superSample 2.2 / 2.6 (index.js:1363,1426-1428), so its edges are fractional as a rule.
 * makeEnvironment({edgeAA: false}) selects the pixel-centre rule instead.
 */
const path = require('path');
const native = require(path.join(__dirname, 'imagestitch.node'));

const QUARTER = 1.5707963267948966;

function parseColor(style) {
  const s = String(style).trim().toLowerCase();
  let m;
  if ((m = /^#([0-9a-f]{3})$/.exec(s))) return m[1].split('').map((c) => parseInt(c + c, 16)).concat(255);
  if ((m = /^#([0-9a-f]{6})$/.exec(s))) return [0, 2, 4].map((i) => parseInt(m[1].substr(i, 2), 16)).concat(255);
  if (s === 'white') return [255, 255, 255, 255];
  if (s === 'black') return [0, 0, 0, 255];
  throw new Error('unsupported fillStyle ' + style);
}

class Context2D {
  constructor(canvas) {
    this.canvas = canvas;
    this.imageSmoothingEnabled = true;       // Canvas default
    this.imageSmoothingQuality = 'low';
    this.fillStyle = '#000000';
    this._m = [1, 0, 0, 1, 0, 0];
    this._stack = [];
  }
  save() { this._stack.push({ m: this._m.slice(), fillStyle: this.fillStyle }); }
  restore() { const s = this._stack.pop(); if (s) { this._m = s.m; this.fillStyle = s.fillStyle; } }
  setTransform(a, b, c, d, e, f) { this._m = [a, b, c, d, e, f]; }
  resetTransform() { this._m = [1, 0, 0, 1, 0, 0]; }
  translate(x, y) { const [a, b, c, d, e, f] = this._m; this._m = [a, b, c, d, a * x + c * y + e, b * x + d * y + f]; }
  scale(x, y) { const [a, b, c, d, e, f] = this._m; this._m = [a * x, b * x, c * y, d * y, e, f]; }
  rotate(r) {
    // quarter turns use exact cos/sin (DESIGN.md raster contract): 0.5*Math.PI is not pi/2 and cos() of it is 6e-17
    const q = r / QUARTER, qr = Math.floor(q + 0.5);
    let co, si;
    if (Math.abs(q - qr) < 1e-9) { const k = ((qr % 4) + 4) % 4; co = [1, 0, -1, 0][k]; si = [0, 1, 0, -1][k]; }
    else { co = Math.cos(r); si = Math.sin(r); }
    const [a, b, c, d, e, f] = this._m;
    this._m = [a * co + c * si, b * co + d * si, c * co - a * si, d * co - b * si, e, f];
  }
  fillRect(x, y, w, h) { this.canvas._ops.push({ kind: 0, image: -1, m: this._m.slice(), s: [0, 0, 0, 0], d: [x, y, w, h], rgba: parseColor(this.fillStyle) }); }
  clearRect(x, y, w, h) {
    const [a, b, c, d, e, f] = this._m;
    if (b !== 0 || c !== 0) throw new Error('clearRect under a rotated transform is outside the stitch path');
    const x0 = Math.min(a * x + e, a * (x + w) + e), x1 = Math.max(a * x + e, a * (x + w) + e);
    const y0 = Math.min(d * y + f, d * (y + h) + f), y1 = Math.max(d * y + f, d * (y + h) + f);
    if (x0 <= 0.5 && y0 <= 0.5 && x1 >= this.canvas.width - 0.5 && y1 >= this.canvas.height - 0.5) { this.canvas._ops = []; return; }
    throw new Error('clearRect of a sub-rectangle is outside the stitch path');
  }
  drawImage(img, ...a) {
    if (!img || !img.width || !img.height || !img._pixels) throw new Error('drawImage: image is not decoded');
    let s, d;
    if (a.length === 2) { s = [0, 0, img.width, img.height]; d = [a[0], a[1], img.width, img.height]; }
    else if (a.length === 4) { s = [0, 0, img.width, img.height]; d = a; }
    else if (a.length === 8) { s = a.slice(0, 4); d = a.slice(4); }
    else throw new TypeError('drawImage expects 3, 5 or 9 arguments');
    this.canvas._ops.push({ kind: 1, bitmap: { width: img.width, height: img.height, data: img._pixels, opaque: !!img._opaque }, m: this._m.slice(), s, d, rgba: [0, 0, 0, 0] });
  }
  getImageData(x, y, w, h) {
    const data = this.canvas._render({ x, y, w, h }, this.imageSmoothingEnabled);
    return { width: w, height: h, data: new Uint8ClampedArray(data.buffer, data.byteOffset, data.length) };
  }
}

class ShimImage {
  constructor(files) { this._files = files; this.width = 0; this.height = 0; this.onload = null; this.onerror = null; this._src = ''; this._pixels = null; }
  get src() { return this._src; }
  set src(v) {
    this._src = v;
    if (!v) { this._pixels = null; this.width = 0; this.height = 0; return; }     // `bmp.src = ''` releases (index.js:1569)
    let f = this._files[v];
    if (!f && /\.(png|jpe?g|bmp|gif)$/i.test(v)) {  // a real file on disk: decode it (the platform's Image.src does the same)
      try { f = this._files[v] = native.decodeImage(require('fs').readFileSync(v)); } catch (e) { f = null; }
    }
    setImmediate(() => {
      if (this._src !== v) return;
      if (f && f.data) { this.width = f.width; this.height = f.height; this._pixels = f.data; this._opaque = !!f.opaque; if (this.onload) this.onload(); }
      else if (this.onerror) this.onerror(new Error('decode failed: ' + v));
    });
  }
  close() { this._pixels = null; }
}

class ShimCanvas {
  constructor(env, width, height) { this._env = env; this._w = Math.floor(width || 0); this._h = Math.floor(height || 0); this._ops = []; this._ctx = null; }
  get width() { return this._w; }
  set width(v) { this._w = Math.floor(v || 0); this._ops = []; if (this._ctx) this._ctx.resetTransform(); }   // resizing clears a canvas
  get height() { return this._h; }
  set height(v) { this._h = Math.floor(v || 0); this._ops = []; if (this._ctx) this._ctx.resetTransform(); }
  getContext(kind) { if (kind !== '2d') return null; if (!this._ctx) this._ctx = new Context2D(this); return this._ctx; }
  createImage() { return new ShimImage(this._env.files); }
  createOffscreenCanvas(o) {
    if (!this._env.recordOnly && !native.deviceCount()) throw new Error('OffscreenCanvas 不可用');   // utils/canvas.js:149
    return new ShimCanvas(this._env, o && o.width, o && o.height);
  }
  toTempFilePath(o) {
    try { const r = this._env.exportCanvas(this, o); if (o.success) o.success(r); } catch (e) { if (o.fail) o.fail(e); }
  }
  /** one fused launch over the recorded ops; returns the region's RGBA bytes */
  _render(region, smoothing, asPng) {
    if (this._w < 1 || this._h < 1) throw new Error('canvas has no size');
    const bitmaps = [], index = new Map();
    const packed = new Float64Array(this._ops.length * 18);
    this._ops.forEach((op, i) => {
      let image = -1;
      if (op.kind === 1) {
        if (!index.has(op.bitmap.data)) { index.set(op.bitmap.data, bitmaps.length); bitmaps.push(op.bitmap); }
        image = index.get(op.bitmap.data);
      }
      const c = op.rgba;
      packed.set([op.kind, image, ...op.m, ...op.s, ...op.d, (c[0] | (c[1] << 8) | (c[2] << 16)) + c[3] * 16777216, 0], i * 18);
    });
    const reg = region ? { x: Math.floor(region.x), y: Math.floor(region.y), w: Math.floor(region.w), h: Math.floor(region.h) } : null;
    if (this._env.recordOnly) {     // test hook: capture the op list instead of launching (no GPU needed)
      this._env.recorded.push({ canvasW: this._w, canvasH: this._h, region: reg, smoothing: !!smoothing, ops: Array.from(packed),
                                bitmaps: bitmaps.map((b) => [b.width, b.height]) });
      return Buffer.alloc(Math.min((reg ? reg.w * reg.h : this._w * this._h) * 4, 1 << 16));   // placeholder pixels
    }
    return native.render(this._w, this._h, new Uint8Array([0, 0, 0, 0]), packed, bitmaps, (smoothing ? 1 : 0) | (this._env.edgeAA === false ? 0 : 0x100), reg, !!asPng);
  }
}

/**
 * A `wx` + canvas-node environment for running the reference page.  files: {path: {width, height, data, opaque?,
 * orientation?}} are the "decoded bitmaps" (decode is outside the path: SURVEY.md section 8f rank 3).
 */
function makeEnvironment({ platform = 'devtools', files = {}, storage = {}, recordOnly = false, outDir = null, edgeAA = true } = {}) {
  const env = { platform, files, storage, exports: {}, toasts: [], nextExport: 0, recordOnly, recorded: [], outDir, edgeAA };
  env.exportCanvas = (canvas, o) => {
    const w = Math.max(1, Math.floor(o.width || canvas.width)), h = Math.max(1, Math.floor(o.height || canvas.height));
    if ((o.destWidth && o.destWidth !== w) || (o.destHeight && o.destHeight !== h)) throw new Error('export rescale is outside the stitch path');
    const smoothing = canvas._ctx ? canvas._ctx.imageSmoothingEnabled : true;
    const data = canvas._render({ x: o.x || 0, y: o.y || 0, w, h }, smoothing);
    const p = 'shim://export/' + (env.nextExport++) + '.' + (o.fileType || 'png');
    env.exports[p] = { width: w, height: h, data };
    // fileType 'png' of the whole canvas: also produce the file itself (GPU encoder), written out when outDir is set
    if (!env.recordOnly && (o.fileType || 'png') === 'png' && (o.x || 0) === 0 && (o.y || 0) === 0 && w === canvas.width && h === canvas.height) {
      env.exports[p].png = canvas._render(null, smoothing, true);
      if (env.outDir) {
        const file = require('path').join(env.outDir, 'export' + (env.nextExport - 1) + '.png');
        require('fs').writeFileSync(file, env.exports[p].png);
        env.exports[p].file = file;
      }
    }
    env.files[p] = { width: w, height: h, data, opaque: true };
    return { tempFilePath: p };
  };
  env.mainCanvas = new ShimCanvas(env, 0, 0);
  env.wx = {
    env: { USER_DATA_PATH: '/tmp/imagestitch_shim' },
    getSystemInfoSync: () => ({ platform, pixelRatio: 2, windowWidth: 375, windowHeight: 667, model: 'MI355X', brand: 'AMD', system: 'linux', SDKVersion: '3.10.3' }),
    getWindowInfo: () => ({ pixelRatio: 2, windowWidth: 375, windowHeight: 667 }),
    getFileSystemManager: () => ({ statSync() { throw new Error('nofile'); }, appendFileSync() {}, writeFileSync() {}, getFileInfo(o) { if (o && o.success) o.success({ size: 0 }); } }),
    setStorageSync: (k, v) => { storage[k] = v; },
    getStorageSync: (k) => storage[k],
    getImageInfo: (o) => { const f = files[o.src]; if (f) o.success({ width: f.width, height: f.height, type: 'png', orientation: 'up', path: o.src }); else o.fail(new Error('nofile')); },
    saveFile: (o) => o.success({ savedFilePath: o.tempFilePath }),
    removeSavedFile() {},
    createSelectorQuery() { const q = { select: () => q, fields: () => q, exec: (cb) => setImmediate(() => cb([{ node: env.mainCanvas, width: 343, height: 457 }])) }; return q; },
    createOffscreenCanvas: (o) => env.mainCanvas.createOffscreenCanvas(o),
    canvasToTempFilePath: (o) => { try { return Promise.resolve(env.exportCanvas(o.canvas, o)); } catch (e) { return Promise.reject(e); } },
    previewImage() {}, showModal() {}, showLoading() {}, hideLoading() {},
    showToast: (o) => { env.toasts.push(o && o.title); },
  };
  return env;
}

module.exports = { ShimCanvas, ShimImage, Context2D, makeEnvironment, parseColor };
