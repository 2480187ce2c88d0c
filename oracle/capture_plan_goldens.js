#!/usr/bin/env node
/*
 * TEST INFRASTRUCTURE — golden-vector generator (never shipped, never on the product path).
 *
 * Runs the reference page script (miniprogram/pages/index/index.js) UNMODIFIED under Node with a
 * stub `wx` + a recording Canvas-2D, drives Page.onStitch() for a list of cases, and writes the exact
 * ordered Canvas call sequence (fillRect / scale / save / translate / rotate / drawImage / restore /
 * getImageData / export, with every numeric argument) to tests/golden/plan_goldens.json.
 *
 * Only runs where /root/reference exists (the authoring container). The reference sources are read in
 * place via require(); nothing from them is copied into this repo - only call traces (data) are stored.
 *
 *   node oracle/capture_plan_goldens.js [/root/reference] [tests/golden/plan_goldens.json]
 *   node oracle/capture_plan_goldens.js /root/reference tests/golden/plan_goldens_random.json --random 400 20261004
 *       instead of the named cases: N seeded random cases (1-12 images of 1..9000 px, EXIF orientations, file sizes around
 *       the bigTask threshold, every platform, stored canvas limits from tiny to lifted, both directions, the three modes,
 *       integer and fractional gaps)
 *   ... --check   (last argument, either form): regenerate in memory and compare with the file at the output path as PARSED
 *       JSON (whitespace-insensitive: the committed random set is minified); the "reference" field, which names the Node
 *       version that produced the file, is not compared.  Exit status 0 = identical, 2 = different; nothing is written.
 */
'use strict';
const path = require('path');
const fs = require('fs');

const REF = process.argv[2] || '/root/reference';
const OUT = process.argv[3] || path.join(__dirname, '..', 'tests', 'golden', 'plan_goldens.json');
const PAGE_JS = path.join(REF, 'miniprogram-stitch', 'miniprogram', 'pages', 'index', 'index.js');

// ---- silence the reference's console chatter while keeping our own output -------------------------
const realLog = console.log.bind(console);
console.log = () => {}; console.warn = () => {}; console.error = () => {};

// ---- recording Canvas-2D --------------------------------------------------------------------------
function makeCtx(trace) {
  const ctx = {
    imageSmoothingEnabled: false,
    imageSmoothingQuality: 'low',
    _fillStyle: '#000000',
    set fillStyle(v) { this._fillStyle = v; trace.push({ op: 'fillStyle', v }); },
    get fillStyle() { return this._fillStyle; },
    fillRect(x, y, w, h) { trace.push({ op: 'fillRect', a: [x, y, w, h] }); },
    clearRect(x, y, w, h) { trace.push({ op: 'clearRect', a: [x, y, w, h] }); },
    scale(x, y) { trace.push({ op: 'scale', a: [x, y] }); },
    translate(x, y) { trace.push({ op: 'translate', a: [x, y] }); },
    rotate(r) { trace.push({ op: 'rotate', a: [r] }); },
    save() { trace.push({ op: 'save' }); },
    restore() { trace.push({ op: 'restore' }); },
    setTransform(a, b, c, d, e, f) { trace.push({ op: 'setTransform', a: [a, b, c, d, e, f] }); },
    drawImage(img, ...a) { trace.push({ op: 'drawImage', img: img._id, bmp: [img.width, img.height], a }); },
    getImageData(x, y, w, h) { trace.push({ op: 'getImageData', a: [x, y, w, h] }); return { data: new Uint8ClampedArray(4 * w * h), width: w, height: h }; },
  };
  return ctx;
}

function makeCanvas(env, label, width, height) {
  const canvas = {
    _label: label, _w: width || 0, _h: height || 0, _ctx: null,
    get width() { return this._w; },
    set width(v) { this._w = v; env.trace.push({ op: 'canvas.width', canvas: label, v }); },
    get height() { return this._h; },
    set height(v) { this._h = v; env.trace.push({ op: 'canvas.height', canvas: label, v }); },
    getContext() {
      if (!this._ctx) {
        const t = [];
        this._ctx = makeCtx({ push: (e) => env.trace.push(Object.assign({ canvas: label }, e)) });
        this._ctx._t = t;
      }
      return this._ctx;
    },
    createImage() { return makeImage(env); },
    createOffscreenCanvas(o) {
      env.trace.push({ op: 'createOffscreenCanvas', a: o ? [o.width, o.height] : [] });
      return makeCanvas(env, 'off' + (env.offCount++), o && o.width, o && o.height);
    },
  };
  return canvas;
}

function makeImage(env) {
  const img = { width: 0, height: 0, onload: null, onerror: null, _src: '', _id: -1 };
  Object.defineProperty(img, 'src', {
    get() { return this._src; },
    set(v) {
      this._src = v;
      if (!v) return;
      const meta = env.files[v];
      setImmediate(() => {
        if (meta) { this.width = meta.bmpW; this.height = meta.bmpH; this._id = meta.id; if (this.onload) this.onload(); }
        else if (this.onerror) this.onerror(new Error('no such file ' + v));
      });
    },
  });
  return img;
}

// ---- stub wx ----------------------------------------------------------------------------------------
function makeWx(env) {
  return {
    env: { USER_DATA_PATH: '/tmp/ist_stub_user' },
    getSystemInfoSync() { return { platform: env.platform, pixelRatio: 2, windowWidth: 375, windowHeight: 667, model: 'stub', brand: 'stub', system: 'stub', SDKVersion: '3.10.3' }; },
    getWindowInfo() { return { pixelRatio: 2, windowWidth: 375, windowHeight: 667 }; },
    getFileSystemManager() { return { statSync() { throw new Error('nofile'); }, appendFileSync() {}, writeFileSync() {}, getFileInfo(o) { if (o && o.success) o.success({ size: 0 }); } }; },
    setStorageSync(k, v) { env.storage[k] = v; },
    getStorageSync(k) { return env.storage[k]; },
    getImageInfo(o) { const m = env.files[o.src]; if (m) o.success({ width: m.w, height: m.h, type: 'jpeg', orientation: m.orientationName || 'up', path: o.src }); else o.fail(new Error('nofile')); },
    saveFile(o) { o.success({ savedFilePath: o.tempFilePath }); },
    removeSavedFile() {},
    createSelectorQuery() {
      const q = { select() { return q; }, fields() { return q; }, exec(cb) { setImmediate(() => cb([{ node: env.mainCanvas, width: 343, height: 457 }])); } };
      return q;
    },
    canvasToTempFilePath(o) {
      env.trace.push({ op: 'export', canvas: o.canvas && o.canvas._label, a: [o.x, o.y, o.width, o.height, o.destWidth, o.destHeight], fileType: o.fileType, quality: o.quality });
      return Promise.resolve({ tempFilePath: '/tmp/ist_stub_export.png' });
    },
    previewImage() {}, showToast(o) { env.trace.push({ op: 'toast', title: o && o.title }); }, showModal() {}, showLoading() {}, hideLoading() {},
  };
}

async function runCase(c) {
  const env = { platform: c.platform || 'devtools', storage: {}, files: {}, trace: [], offCount: 0 };
  if (c.canvasLimit) env.storage.canvasLimit = Object.assign({ platform: env.platform }, c.canvasLimit);
  env.mainCanvas = makeCanvas(env, 'main', 0, 0);
  env.files['/tmp/ist_stub_export.png'] = { id: -2, w: 1, h: 1, bmpW: 1, bmpH: 1 };
  let page = null;
  global.Page = (o) => { page = o; };
  global.wx = makeWx(env);
  delete require.cache[require.resolve(PAGE_JS)];
  delete require.cache[require.resolve(path.join(REF, 'miniprogram-stitch', 'miniprogram', 'utils', 'canvas.js'))];
  require(PAGE_JS);
  page.setData = function (d) { Object.assign(this.data, d); };
  page.onLoad();
  const images = c.images.map((im, i) => {
    const p = 'wxfile://usr/img' + i + '.jpg';
    env.files[p] = { id: i, w: im.w, h: im.h, bmpW: im.bmpW || im.w, bmpH: im.bmpH || im.h };
    return { id: 'i' + i, tempFilePath: p, preparedPath: p, prepared: true, naturalWidth: im.w, naturalHeight: im.h,
             width: im.w, height: im.h, orientation: im.orientation || 1, fileSize: im.fileSize || 0 };
  });
  page.data.images = images;
  page.data.direction = c.direction;
  page.data.gap = c.gap || 0;
  if (c.mode) { page.data.verticalStitchMode = c.mode; page.data.horizontalStitchMode = c.mode; }
  env.trace.length = 0;           // drop onLoad noise
  await page.onStitch();
  const t0 = Date.now();
  while (page.data.isStitching && Date.now() - t0 < 30000) await new Promise((r) => setTimeout(r, 1));
  if (page.data.isStitching) throw new Error('stitch did not finish: ' + c.name);
  // keep the stitch portion only: from the offscreen creation to the export (the preview redraw that follows
  // is UI, SURVEY.md section 2 row 14)
  const t = env.trace;
  const start = t.findIndex((e) => e.op === 'createOffscreenCanvas');
  const end = t.findIndex((e) => e.op === 'export');
  const failed = t.find((e) => e.op === 'toast');
  return {
    name: c.name, input: c,
    limits: { deviceMaxCanvasSize: page.deviceMaxCanvasSize, deviceMaxCanvasPixels: page.deviceMaxCanvasPixels },
    error: failed ? failed.title : null,
    calls: (start >= 0 && end >= 0) ? t.slice(start, end + 1).filter((e) => !['fillStyle'].includes(e.op) || true) : t,
  };
}

const IM12 = { w: 4032, h: 3024 };
const MIXED4 = [{ w: 4032, h: 3024 }, { w: 3024, h: 4032 }, { w: 4000, h: 3000 }, { w: 1920, h: 1080 }];
const MIXED7 = [{ w: 4032, h: 3024 }, { w: 1080, h: 1920 }, { w: 4000, h: 3000 }, { w: 1920, h: 1080 }, { w: 4032, h: 3024 }, { w: 1080, h: 1920 }, { w: 4000, h: 3000 }];
const MIXED9 = [{ w: 4032, h: 3024 }, { w: 3024, h: 4032 }, { w: 4000, h: 3000 }, { w: 3840, h: 2160 }, { w: 4032, h: 3024 }, { w: 3024, h: 4032 }, { w: 4000, h: 3000 }, { w: 3840, h: 2160 }, { w: 4032, h: 3024 }];
const LIFT = { size: 1048576, pixels: Math.pow(2, 40) };
const rep = (o, n) => Array.from({ length: n }, () => Object.assign({}, o));

const cases = [];
for (const platform of ['devtools', 'android', 'ios']) {
  cases.push({ name: `G1_3x640x480_v_min_${platform}`, platform, direction: 'vertical', images: rep({ w: 640, h: 480 }, 3) });
  cases.push({ name: `G2_9x12MP_v_${platform}`, platform, direction: 'vertical', images: rep(IM12, 9) });
  cases.push({ name: `G3_9x12MP_h_${platform}`, platform, direction: 'horizontal', images: rep(IM12, 9) });
}
cases.push({ name: 'G1_3x640x480_v_lifted', platform: 'devtools', canvasLimit: LIFT, direction: 'vertical', images: rep({ w: 640, h: 480 }, 3) });
cases.push({ name: 'G2_9x12MP_v_lifted', platform: 'devtools', canvasLimit: LIFT, direction: 'vertical', images: rep(IM12, 9) });
cases.push({ name: 'G3_9x12MP_h_lifted', platform: 'devtools', canvasLimit: LIFT, direction: 'horizontal', images: rep(IM12, 9) });
cases.push({ name: 'G4_64x48MP_v_lifted', platform: 'devtools', canvasLimit: LIFT, direction: 'vertical', images: rep({ w: 8000, h: 6000 }, 64) });
for (const mode of ['min', 'max', 'original']) {
  for (const direction of ['vertical', 'horizontal']) {
    cases.push({ name: `G5_mixed4_${direction[0]}_${mode}_gap10_lifted`, platform: 'devtools', canvasLimit: LIFT, direction, mode, gap: 10, images: MIXED4 });
    cases.push({ name: `G5_mixed4_${direction[0]}_${mode}_gap0_devtools`, platform: 'devtools', direction, mode, gap: 0, images: MIXED4 });
    cases.push({ name: `G5_mixed9_${direction[0]}_${mode}_gap0_lifted`, platform: 'devtools', canvasLimit: LIFT, direction, mode, gap: 0, images: MIXED9 });
    cases.push({ name: `G5_mixed9_${direction[0]}_${mode}_gap7_ios`, platform: 'ios', direction, mode, gap: 7, images: MIXED9 });
  }
}
cases.push({ name: 'G6_mixed7_h_min_gap8_ios', platform: 'ios', direction: 'horizontal', mode: 'min', gap: 8, images: MIXED7 });
cases.push({ name: 'G7_mixed7_v_min_gap10_android', platform: 'android', direction: 'vertical', mode: 'min', gap: 10, images: MIXED7 });
cases.push({ name: 'G7_mixed7_v_original_gap10_android', platform: 'android', direction: 'vertical', mode: 'original', gap: 10, images: MIXED7 });
for (let o = 1; o <= 8; o++) {
  cases.push({ name: `G8_orient${o}_v_lifted`, platform: 'devtools', canvasLimit: LIFT, direction: 'vertical',
               images: [{ w: 3024, h: 4032, orientation: o }, { w: 4032, h: 3024, orientation: o }] });
}
cases.push({ name: 'G9_bigbytes_2img_ios', platform: 'ios', direction: 'vertical', images: [{ w: 3000, h: 2000, fileSize: 20 * 1024 * 1024 }, { w: 2000, h: 3000, fileSize: 6 * 1024 * 1024 }] });
cases.push({ name: 'G9_tiny_1x1_and_wide', platform: 'devtools', canvasLimit: LIFT, direction: 'vertical', gap: 3, images: [{ w: 1, h: 1 }, { w: 5000, h: 3 }, { w: 7, h: 9000 }] });
cases.push({ name: 'G9_single_android', platform: 'android', direction: 'horizontal', images: [{ w: 6000, h: 4000 }] });
cases.push({ name: 'G9_gap20_h_original_ios', platform: 'ios', direction: 'horizontal', mode: 'original', gap: 20, images: MIXED7 });
cases.push({ name: 'G9_stored_limit_android_8192', platform: 'android', canvasLimit: { size: 8192, pixels: 8192 * 8192 }, direction: 'vertical', gap: 5, images: MIXED9 });

// ---- seeded random cases (xorshift32: the same list on every run) ---------------------------------------------
function randomCases(n, seed) {
  let s = seed >>> 0 || 1;
  const rnd = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s / 4294967296; };
  const ri = (lo, hi) => lo + Math.floor(rnd() * (hi - lo + 1));
  const pick = (a) => a[Math.floor(rnd() * a.length)];
  const dim = () => { const r = rnd(); return r < 0.1 ? ri(1, 8) : r < 0.55 ? ri(9, 1200) : ri(1201, 9000); };
  const out = [];
  for (let k = 0; k < n; k++) {
    const count = rnd() < 0.25 ? ri(7, 12) : ri(1, 6);
    const same = rnd() < 0.2 ? { w: dim(), h: dim() } : null;
    const images = [];
    for (let i = 0; i < count; i++) {
      const im = same ? Object.assign({}, same) : { w: dim(), h: dim() };
      if (rnd() < 0.3) im.orientation = ri(1, 8);
      if (rnd() < 0.3) im.fileSize = pick([0, 1, 3 << 20, 9 << 20, 13 << 20, (25 << 20) - 1, 25 << 20, 40 << 20]);
      images.push(im);
    }
    const c = { name: 'R' + k, platform: pick(['devtools', 'android', 'ios']), direction: pick(['vertical', 'horizontal']), images };
    const m = pick(['', 'min', 'max', 'original']);
    if (m) c.mode = m;
    const g = rnd();
    if (g < 0.5) c.gap = ri(0, 20); else if (g < 0.65) c.gap = Math.round(rnd() * 2000) / 100;
    const l = rnd();
    if (l < 0.2) c.canvasLimit = LIFT;
    else if (l < 0.45) { const size = pick([64, 500, 1024, 2048, 4096, 8192, 12288, 16384, 32767]); c.canvasLimit = { size, pixels: pick([size * size, size * 1024, 4096 * 2048, 1 << 20, 1 << 24, 1 << 28]) }; }
    out.push(c);
  }
  return out;
}
const CHECK = process.argv[process.argv.length - 1] === '--check';
if (process.argv[4] === '--random') {
  const list = randomCases(parseInt(process.argv[5] || '400', 10), parseInt(process.argv[6] || '1', 10));
  cases.length = 0;
  for (const c of list) cases.push(c);
}

(async () => {
  const out = { generator: 'oracle/capture_plan_goldens.js', reference: 'Iamctb/ImageStitching miniprogram/pages/index/index.js (run unmodified under Node ' + process.version + ' with a stub wx)', cases: [] };
  for (const c of cases) out.cases.push(await runCase(c));
  if (CHECK) {
    const have = JSON.parse(fs.readFileSync(OUT, 'utf8'));
    const now = JSON.parse(JSON.stringify(out));         // what a reader of the file would get (undefined dropped, -0 -> 0, ...)
    let bad = have.cases.length === now.cases.length ? 0 : 1;
    if (bad) realLog('case count differs:', have.cases.length, 'stored,', now.cases.length, 'regenerated');
    for (let k = 0; k < Math.min(have.cases.length, now.cases.length); k++)
      if (JSON.stringify(have.cases[k]) !== JSON.stringify(now.cases[k])) { bad++; realLog('case differs:', now.cases[k].name); }
    realLog(bad ? 'CHECK FAILED' : 'check ok:', OUT, now.cases.length, 'cases regenerate identically from the reference');
    process.exit(bad ? 2 : 0);
  }
  fs.mkdirSync(path.dirname(OUT), { recursive: true });
  fs.writeFileSync(OUT, JSON.stringify(out, null, 1));
  realLog('wrote', OUT, out.cases.length, 'cases');
  for (const c of out.cases) {
    const off = c.calls.find((e) => e.op === 'createOffscreenCanvas');
    const draws = c.calls.filter((e) => e.op === 'drawImage').length;
    realLog(' ', c.name, off ? off.a.join('x') : '-', 'draws', draws, c.error ? 'ERROR ' + c.error : '');
  }
})().catch((e) => { realLog('FAILED', e && e.stack || e); process.exit(1); });
