#!/usr/bin/env node
'use strict';
/**
 * Command-line driver used by the parity tests and by hand:
 *   node node/cli.js job.json
 * job = { mode: 'stitch' | 'shim' | 'reference', direction, opts, out: 'file.rgba',
 *         images: [{width, height, orientation?, fileSize?, file: 'raw RGBA8 file'}],
 *         platform?, canvasLimit?, gap?, stitchMode?, referenceRoot? }
 *   stitch    : index.js stitch(images, direction, opts)                (surface S1)
 *   shim      : replays the Canvas calls of a plan through canvas_shim.js (surface S2, no reference needed)
 *   reference : runs the reference's UNMODIFIED pages/index/index.js onStitch against canvas_shim.js
 *               (only where the reference checkout exists; nothing of it is copied)
 * Writes raw RGBA8 to job.out and prints {width,height,...} as JSON.
 */
const fs = require('fs');
const path = require('path');

async function main() {
  const job = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
  const images = job.images.map((im) => Object.assign({}, im, { data: im.file ? new Uint8Array(fs.readFileSync(im.file)) : undefined }));
  let result;
  if (job.mode === 'stitch') {
    const api = require('./index.js');
    if (job.png) { const r = await api.stitchPng(images, job.direction, job.opts); fs.writeFileSync(job.out, r.png); console.log(JSON.stringify({ width: r.width, height: r.height, bytes: r.png.length, png: true })); return; }
    result = job.sync ? api.stitchSync(images, job.direction, job.opts) : await api.stitch(images, job.direction, job.opts);
  } else if (job.mode === 'shim') {
    const api = require('./index.js');
    const shim = require('./canvas_shim.js');
    const p = api.plan(images, job.direction, job.opts);
    const files = {};
    images.forEach((im, i) => { files['img' + i] = im; });
    const env = shim.makeEnvironment({ files, outDir: job.outDir || null });
    const off = env.mainCanvas.createOffscreenCanvas({ type: '2d', width: p.canvasW, height: p.canvasH });
    const ctx = off.getContext('2d');
    ctx.imageSmoothingEnabled = (job.opts && job.opts.filter) !== 'nearest';
    ctx.fillStyle = '#ffffff';
    ctx.fillRect(0, 0, p.canvasW, p.canvasH);
    if (p.superSample !== 1) ctx.scale(p.superSample, p.superSample);
    for (const r of p.rects) {
      const bmp = off.createImage();
      await new Promise((res, rej) => { bmp.onload = res; bmp.onerror = rej; bmp.src = 'img' + r.image; });
      ctx.drawImage(bmp, 0, 0, bmp.width, bmp.height, r.dx, r.dy, r.dw, r.dh);       // orientation 1 only in this mode
      ctx.getImageData(0, 0, 1, 1);
    }
    const exp = await env.wx.canvasToTempFilePath({ canvas: off, x: 0, y: 0, width: p.canvasW, height: p.canvasH, destWidth: p.canvasW, destHeight: p.canvasH, fileType: 'png', quality: 1 });
    result = env.exports[exp.tempFilePath];
    if (result.file) result.plan = { file: result.file };
  } else if (job.mode === 'reference') {
    const shim = require('./canvas_shim.js');
    const root = job.referenceRoot || '/root/reference';
    const pageJs = path.join(root, 'miniprogram-stitch', 'miniprogram', 'pages', 'index', 'index.js');
    const files = {};
    const pageImages = images.map((im, i) => {
      const p = 'wxfile://usr/img' + i + '.png';
      files[p] = { width: im.bmpWidth || im.width, height: im.bmpHeight || im.height, data: im.data || new Uint8Array(job.recordOnly ? 4 : 0), opaque: !!im.opaque };
      return { id: 'i' + i, tempFilePath: p, preparedPath: p, prepared: true, naturalWidth: im.width, naturalHeight: im.height,
               width: im.width, height: im.height, orientation: im.orientation || 1, fileSize: im.fileSize || 0 };
    });
    const storage = {};
    if (job.canvasLimit) storage.canvasLimit = Object.assign({ platform: job.platform || 'devtools' }, job.canvasLimit);
    const env = shim.makeEnvironment({ platform: job.platform || 'devtools', files, storage, recordOnly: !!job.recordOnly });
    const log = console.log; console.log = () => {}; console.warn = () => {}; const cerr = console.error; console.error = () => {};
    let page = null;
    global.Page = (o) => { page = o; };
    global.wx = env.wx;
    require(pageJs);
    page.setData = function (d) { Object.assign(this.data, d); };
    page.onLoad();
    page.data.images = pageImages;
    page.data.direction = job.direction;
    page.data.gap = job.gap || 0;
    if (job.stitchMode) { page.data.verticalStitchMode = job.stitchMode; page.data.horizontalStitchMode = job.stitchMode; }
    const origCreate = env.mainCanvas.createOffscreenCanvas.bind(env.mainCanvas);
    env.mainCanvas.createOffscreenCanvas = (o) => { const c = origCreate(o); if (job.filter === 'nearest') { const g = c.getContext.bind(c); c.getContext = (k) => { const x = g(k); Object.defineProperty(x, 'imageSmoothingEnabled', { get: () => false, set: () => {} }); return x; }; } return c; };
    await page.onStitch();
    const t0 = Date.now();
    while (page.data.isStitching && Date.now() - t0 < 120000) await new Promise((r) => setTimeout(r, 2));
    console.log = log; console.error = cerr;
    if (!page.data.stitchedTempPath) throw new Error('reference page failed: ' + JSON.stringify(env.toasts));
    result = env.exports[page.data.stitchedTempPath];
    result.progress = page.data.stitchProgress;
    if (job.recordOnly) result.recorded = env.recorded;
  } else throw new Error('unknown mode ' + job.mode);
  if (!result) { console.log(JSON.stringify({ empty: true })); return; }
  if (!job.recordOnly) fs.writeFileSync(job.out, Buffer.from(result.data.buffer, result.data.byteOffset, result.data.length));
  console.log(JSON.stringify({ width: result.width, height: result.height, bytes: result.data.length, plan: result.plan || null, progress: result.progress, recorded: result.recorded }));
}
main().catch((e) => { console.error(String(e && e.message || e)); console.log(JSON.stringify({ error: String(e && e.message || e), code: e && e.code })); process.exit(3); });
