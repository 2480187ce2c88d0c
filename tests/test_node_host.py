"""The TypeScript/Node host over the N-API addon (node/): plan parity on CPU; on a GPU, pixels through
stitch(images, direction, opts) (surface S1) and through the Canvas-2D shim (surface S2) against the oracle."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import imagestitching_amd as ist
from tests import golden_util as G
from tests import util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "node", "imagestitch.node")
REF = "/root/reference"
needs_node = pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or the built addon is missing")


def _cli(job, tmp_path):
    jp = tmp_path / "job.json"
    jp.write_text(json.dumps(job))
    r = subprocess.run([NODE, os.path.join(ROOT, "node", "cli.js"), str(jp)], capture_output=True, text=True, timeout=300)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "{}"
    return r.returncode, json.loads(line), r.stderr


def _write_images(pixels, tmp_path, orientations=None):
    out = []
    for i, a in enumerate(pixels):
        f = tmp_path / ("img%d.rgba" % i)
        np.ascontiguousarray(a).tofile(f)
        out.append({"width": a.shape[1], "height": a.shape[0], "file": str(f), "orientation": (orientations[i] if orientations else 1)})
    return out


@needs_node
def test_addon_loads_and_plans_like_the_c_abi():
    code = ("const api=require('%s/node/index.js');"
            "const c=JSON.parse(process.argv[1]);"
            "console.log(JSON.stringify(c.map(x=>api.plan(x.images,x.direction,x.opts))));") % ROOT
    cases = [
        {"images": [{"width": 4032, "height": 3024}] * 9, "direction": "vertical", "opts": {}},
        {"images": [{"width": 4032, "height": 3024}, {"width": 3024, "height": 4032}, {"width": 1920, "height": 1080}], "direction": "horizontal",
         "opts": {"mode": "original", "gap": 10, "platform": "ios"}},
        {"images": [{"width": 640, "height": 480, "orientation": 6}] * 3, "direction": "vertical", "opts": {"platform": "android", "gap": 3}},
        {"images": [{"width": 4000, "height": 3000, "fileSize": 30000000}], "direction": "vertical", "opts": {"platform": "devtools", "mode": "max"}},
    ]
    out = subprocess.run([NODE, "-e", code, json.dumps(cases)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    plans = json.loads(out.stdout)
    for c, jp in zip(cases, plans):
        p = ist.plan(c["images"], c["direction"], c["opts"])
        assert (jp["canvasW"], jp["canvasH"], jp["superSample"], jp["scaleDown"], jp["bigTask"]) == (p.canvas_w, p.canvas_h, p.super_sample, p.scale_down, p.big_task)
        assert jp["rects"] == p.rects


@needs_node
def test_node_host_rejects_like_the_reference_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    imgs = _write_images([U.rand_image(1, 4, 4)], tmp_path)
    rc, out, err = _cli({"mode": "stitch", "direction": "vertical", "opts": {}, "images": imgs, "out": str(tmp_path / "o.rgba")}, tmp_path)
    assert rc == 3 and out["code"] == "-5" and out["error"].startswith("拼图失败：")


@needs_node
@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout only exists in the authoring container")
@pytest.mark.parametrize("name", ["G1_3x640x480_v_min_ios", "G2_9x12MP_v_lifted", "G3_9x12MP_h_devtools", "G5_mixed4_h_original_gap10_lifted",
                                  "G7_mixed7_v_min_gap10_android", "G8_orient5_v_lifted", "G8_orient6_v_lifted", "G8_orient7_v_lifted",
                                  "G8_orient8_v_lifted", "G9_gap20_h_original_ios"])
def test_unmodified_reference_page_drives_the_shim_to_the_same_ops(name, goldens, tmp_path):
    """Run the reference's own onStitch (read in place, nothing copied) against node/canvas_shim.js in record-only mode:
    the op list it records must be exactly what ist_plan_ops produces for the same inputs (surface S2 == surface S1)."""
    case = goldens[name]
    inp, lim, _ = G.case_inputs(case)
    job = {"mode": "reference", "recordOnly": True, "direction": inp["direction"], "gap": inp.get("gap", 0),
           "stitchMode": inp.get("mode"), "platform": inp.get("platform", "devtools"), "out": str(tmp_path / "o.rgba"),
           "images": [{"width": im["w"], "height": im["h"], "orientation": im.get("orientation", 1), "fileSize": im.get("fileSize", 0)} for im in inp["images"]]}
    if inp.get("canvasLimit"):
        job["canvasLimit"] = inp["canvasLimit"]
    rc, out, err = _cli(job, tmp_path)
    assert rc == 0, err
    assert out["progress"] == 100
    export = out["recorded"][-1]                    # the export launch (earlier entries are the 1x1 flushes)
    opts = {"mode": inp.get("mode", "min"), "gap": inp.get("gap", 0), "platform": inp.get("platform", "devtools"),
            "maxSide": lim["deviceMaxCanvasSize"], "maxPixels": lim["deviceMaxCanvasPixels"]}
    p = ist.plan([{"width": im["w"], "height": im["h"], "orientation": im.get("orientation", 1), "fileSize": im.get("fileSize", 0)} for im in inp["images"]],
                 inp["direction"], opts)
    assert (export["canvasW"], export["canvasH"]) == (p.canvas_w, p.canvas_h)
    ops = p.ops_as_dicts()
    packed = export["ops"]
    assert len(packed) == 18 * len(ops)
    for i, o in enumerate(ops):
        q = packed[18 * i:18 * i + 18]
        assert q[0] == (0 if o["kind"] == "fill" else 1)
        assert q[2:8] == o["m"] and q[12:16] == o["d"]
        if o["kind"] == "draw":
            assert q[8:12] == o["s"]
    if p.big_task:                                  # bigTask flushes after every image (index.js:1559-1566)
        assert len(out["recorded"]) == len(inp["images"]) + 1
        assert all(r["region"] == {"x": 0, "y": 0, "w": 1, "h": 1} for r in out["recorded"][:-1])


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("filt", ["nearest", "bilinear"])
def test_node_stitch_pixels_match_oracle(filt, tmp_path):
    px = [U.rand_image(100, 48, 64), U.smooth_image(101, 80, 50), U.rand_image(102, 33, 77, opaque=False)]
    ori = [1, 6, 3]
    imgs = _write_images(px, tmp_path, ori)
    opts = {"filter": filt, "mode": "max", "gap": 5}
    for direction in ("vertical", "horizontal"):
        for sync in (False, True):
            rc, out, err = _cli({"mode": "stitch", "sync": sync, "direction": direction, "opts": opts, "images": imgs, "out": str(tmp_path / "o.rgba")}, tmp_path)
            assert rc == 0, err
            ref, pd, _ = U.oracle_stitch(px, direction, opts, ori)
            got = np.fromfile(tmp_path / "o.rgba", np.uint8).reshape(out["height"], out["width"], 4)
            assert got.shape == ref.shape
            assert U.max_abs_diff(got, ref) <= (0 if filt == "nearest" else 1)


@needs_node
@pytest.mark.gpu
def test_canvas_shim_pixels_match_oracle(tmp_path):
    """Surface S2: fillRect + scale + drawImage + getImageData + canvasToTempFilePath recorded by the shim, one fused launch."""
    px = [U.rand_image(110 + i, 48, 64) for i in range(3)]
    imgs = _write_images(px, tmp_path)
    for opts in ({"filter": "nearest"}, {"filter": "bilinear", "platform": "devtools"}, {"filter": "bilinear", "mode": "original", "gap": 9}):
        rc, out, err = _cli({"mode": "shim", "direction": "vertical", "opts": opts, "images": imgs, "out": str(tmp_path / "o.rgba")}, tmp_path)
        assert rc == 0, err
        ref, pd, _ = U.oracle_stitch(px, "vertical", dict(opts, edgeAA=True))      # the shim is a Canvas: fractional edges are anti-aliased
        got = np.fromfile(tmp_path / "o.rgba", np.uint8).reshape(out["height"], out["width"], 4)
        assert got.shape == ref.shape
        assert U.max_abs_diff(got, ref) <= (0 if opts["filter"] == "nearest" else 1)


@needs_node
@pytest.mark.gpu
def test_node_png_export(tmp_path):
    """stitchPng (S1 + export) and the shim's canvasToTempFilePath({fileType:'png'}) write real PNG files."""
    import io
    from PIL import Image
    px = [U.rand_image(500 + i, 48, 64) for i in range(3)]
    imgs = _write_images(px, tmp_path)
    ref, pd, _ = U.oracle_stitch(px, "vertical", {"filter": "nearest"})
    out = tmp_path / "o.png"
    rc, meta, err = _cli({"mode": "stitch", "png": True, "direction": "vertical", "opts": {"filter": "nearest"}, "images": imgs, "out": str(out)}, tmp_path)
    assert rc == 0 and meta["png"], err
    assert np.array_equal(np.asarray(Image.open(out).convert("RGBA")), ref)
    rc, meta, err = _cli({"mode": "shim", "direction": "vertical", "opts": {"filter": "nearest"}, "images": imgs, "out": str(tmp_path / "o.rgba"),
                          "outDir": str(tmp_path)}, tmp_path)
    assert rc == 0, err
    assert np.array_equal(np.asarray(Image.open(meta["plan"]["file"]).convert("RGBA")), ref)
    # opts.pngLevel: the same pixels, stored (0) or compressed on the GPU (1, the default)
    flat = [np.full((48, 64, 4), 255, np.uint8) for _ in range(3)]
    fimgs = _write_images(flat, tmp_path)
    sizes = {}
    for level in (0, 1):
        o = tmp_path / ("lvl%d.png" % level)
        rc, meta, err = _cli({"mode": "stitch", "png": True, "direction": "vertical", "opts": {"filter": "nearest", "pngLevel": level}, "images": fimgs, "out": str(o)}, tmp_path)
        assert rc == 0, err
        assert np.array_equal(np.asarray(Image.open(o).convert("RGBA")), np.concatenate(flat, 0))
        sizes[level] = os.path.getsize(o)
    assert sizes[0] > 3 * 48 * 64 * 4 and sizes[1] < 0.1 * sizes[0]


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("devices,split", [([0], "image"), ([0, 0, 0], "image"), ([0, 0], "band"), ([0, 0, 0], "rows"), ([0] * 4, "auto")])
def test_node_stitch_on_a_device_list(devices, split, tmp_path):
    """opts.devices: the Node host shards the stitch over a device list through ist_stitch_rgba8_multi; on a
    one-GPU box the listed device serves every slot and the result is the single-device result, bit for bit"""
    px = [U.rand_image(160, 48, 64), U.smooth_image(161, 80, 50), U.rand_image(162, 33, 77, opaque=False), U.rand_image(163, 40, 64)]
    imgs = _write_images(px, tmp_path)
    opts = {"filter": "bilinear", "mode": "max", "gap": 3}
    for direction in ("vertical", "horizontal"):
        rc, out, err = _cli({"mode": "stitch", "direction": direction, "opts": dict(opts, devices=devices, split=split), "images": imgs, "out": str(tmp_path / "m.rgba")}, tmp_path)
        assert rc == 0, err
        rc, one, err = _cli({"mode": "stitch", "direction": direction, "opts": opts, "images": imgs, "out": str(tmp_path / "o.rgba")}, tmp_path)
        assert rc == 0, err
        assert (out["width"], out["height"]) == (one["width"], one["height"])
        assert np.array_equal(np.fromfile(tmp_path / "m.rgba", np.uint8), np.fromfile(tmp_path / "o.rgba", np.uint8))
        ref, _, _ = U.oracle_stitch(px, direction, opts)
        assert U.max_abs_diff(np.fromfile(tmp_path / "m.rgba", np.uint8).reshape(ref.shape), ref) <= 1


@needs_node
def test_node_device_list_validation():
    code = ("const api=require('%s/node/index.js'); let bad=0;"
            "for (const o of [{devices:[]},{devices:'0'},{devices:[0.5]},{devices:[-1]},{devices:[0],split:'diagonal'}])"
            "{ try{api.stitchSync([{width:1,height:1,data:new Uint8Array(4)}],'vertical',o)}catch(e){ if (e instanceof TypeError) bad++ } }"
            "console.log(JSON.stringify({bad}));") % ROOT
    out = subprocess.run([NODE, "-e", code], capture_output=True, text=True, timeout=60)
    assert json.loads(out.stdout.strip().splitlines()[-1])["bad"] == 5, out.stdout + out.stderr


@needs_node
def test_node_progress_checkpoints_and_option_validation():
    code = ("const api=require('%s/node/index.js'); const seen=[];"
            "let bad=0; try{api.plan([],'diagonal')}catch(e){bad++} try{api.plan([],'vertical',{bogus:1})}catch(e){bad++}"
            "api.stitch([], 'vertical', {onProgress:p=>seen.push(p)}).then(r=>console.log(JSON.stringify({r, seen, bad})), e=>console.log(JSON.stringify({err:e.message, code:e.code, seen, bad})));") % ROOT
    out = subprocess.run([NODE, "-e", code], capture_output=True, text=True, timeout=60)
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["bad"] == 2
    # empty image list: the reference returns before touching any state (index.js:1189) -> resolves null, no progress
    assert res.get("r", "x") is None and res["seen"] == []


@needs_node
def test_node_decode_png_matches_pil(tmp_path):
    from PIL import Image
    a = U.rand_image(600, 23, 31, opaque=False)
    Image.fromarray(a, "RGBA").save(tmp_path / "a.png")
    code = ("const api=require('%s/node/index.js'); const fs=require('fs');"
            "const r=api.decodePng(fs.readFileSync(process.argv[1])); fs.writeFileSync(process.argv[2], r.data); console.log(JSON.stringify([r.width,r.height]));"
            "try{api.decodePng(Buffer.from([0xff,0xd8,0xff,0xe0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46]))}catch(e){console.log(JSON.stringify({code:e.code,msg:e.message}))}") % ROOT
    out = subprocess.run([NODE, "-e", code, str(tmp_path / "a.png"), str(tmp_path / "a.rgba")], capture_output=True, text=True, timeout=60)
    lines = out.stdout.strip().splitlines()
    assert json.loads(lines[0]) == [31, 23], out.stderr
    assert np.array_equal(np.fromfile(tmp_path / "a.rgba", np.uint8).reshape(23, 31, 4), a)
    err = json.loads(lines[1])
    assert err["code"] == "-7" and "JPEG" in err["msg"]


@needs_node
@pytest.mark.gpu
def test_node_stitch_files_png_in_png_out(tmp_path):
    """File to file: PNG inputs -> decode -> stitch -> GPU PNG export (the whole onStitch for 'png' images)."""
    from PIL import Image
    px = [U.rand_image(610 + i, h, w) for i, (w, h) in enumerate([(64, 48), (50, 80), (33, 20)])]
    paths = []
    for i, a in enumerate(px):
        p = tmp_path / ("in%d.png" % i)
        Image.fromarray(a, "RGBA").save(p)
        paths.append(str(p))
    code = ("const api=require('%s/node/index.js');"
            "api.stitchFiles(JSON.parse(process.argv[1]),'vertical',{mode:'max',gap:3},process.argv[2]).then(r=>console.log(JSON.stringify([r.width,r.height,r.png.length])),e=>{console.log(e.message);process.exit(3)});") % ROOT
    out = subprocess.run([NODE, "-e", code, json.dumps(paths), str(tmp_path / "out.png")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    ref, pd, _ = U.oracle_stitch(px, "vertical", {"mode": "max", "gap": 3})
    got = np.asarray(Image.open(tmp_path / "out.png").convert("RGBA"))
    assert got.shape == ref.shape and U.max_abs_diff(got, ref) <= 1


@needs_node
@pytest.mark.gpu
def test_node_stitch_files_with_jpeg_inputs(tmp_path):
    """JPEG + PNG files in, PNG out, through the Node host; EXIF orientation comes from the file."""
    from PIL import Image
    a = U.smooth_image(620, 48, 64)[..., :3]
    b = U.smooth_image(621, 64, 48)
    exif = Image.Exif()
    exif[0x0112] = 3
    Image.fromarray(a).save(tmp_path / "a.jpg", quality=90, exif=exif)
    Image.fromarray(b, "RGBA").save(tmp_path / "b.png")
    paths = [str(tmp_path / "a.jpg"), str(tmp_path / "b.png")]
    code = ("const api=require('%s/node/index.js');"
            "api.stitchFiles(JSON.parse(process.argv[1]),'horizontal',{filter:'nearest'},process.argv[2]).then(r=>console.log(JSON.stringify([r.width,r.height])),e=>{console.log(e.message);process.exit(3)});") % ROOT
    out = subprocess.run([NODE, "-e", code, json.dumps(paths), str(tmp_path / "out.png")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    bitmaps = [np.asarray(Image.open(tmp_path / "a.jpg").convert("RGBA")), b]
    ref, pd, _ = U.oracle_stitch(bitmaps, "horizontal", {"filter": "nearest"}, orientations=[3, 1])
    got = np.asarray(Image.open(tmp_path / "out.png").convert("RGBA"))
    assert np.array_equal(got, ref)
