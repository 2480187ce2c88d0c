"""Mutation fuzzing of the host-side file parsers under AddressSanitizer + UBSan (CPU build only; tools/run_fuzz.sh).
A malformed PNG / JPEG / BMP / GIF must come back as an error code, never as a crash or an out-of-bounds access."""
import os
import shutil
import subprocess

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _seeds(d):
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:48, 0:64]
    a = np.stack([(xx * 4) % 256, (yy * 5) % 256, ((xx + yy) * 3) % 256], -1)
    a = (a + rng.integers(-20, 20, a.shape)).clip(0, 255).astype(np.uint8)
    im = Image.fromarray(a, "RGB")
    ex = Image.Exif()
    ex[0x0112] = 6
    todo = [("rgb.png", im, "PNG", {}), ("rgba.png", im.convert("RGBA"), "PNG", {}), ("pal.png", im.convert("P"), "PNG", {}),
            ("g16.png", Image.fromarray(a[..., 0].astype(np.uint16) * 257), "PNG", {}),
            ("444.jpg", im, "JPEG", {"subsampling": 0}), ("420.jpg", im, "JPEG", {"subsampling": 2, "quality": 70}),
            ("422.jpg", im, "JPEG", {"subsampling": 1}), ("grey.jpg", im.convert("L"), "JPEG", {}),
            ("rst.jpg", im, "JPEG", {"restart_marker_blocks": 2}), ("exif.jpg", im, "JPEG", {"exif": ex}),
            ("prog.jpg", im, "JPEG", {"progressive": True, "subsampling": 2}), ("prog444.jpg", im, "JPEG", {"progressive": True, "subsampling": 0}),
            ("24.bmp", im, "BMP", {}), ("8.bmp", im.convert("P"), "BMP", {}), ("1.bmp", im.convert("1"), "BMP", {}),
            ("a.gif", im.convert("P"), "GIF", {}), ("i.gif", im.convert("P"), "GIF", {"interlace": True}),
            ("t.gif", im.convert("P"), "GIF", {"transparency": 3}),
            ("ll.webp", im, "WEBP", {"lossless": True}), ("lla.webp", im.convert("RGBA"), "WEBP", {"lossless": True, "quality": 100, "method": 6}),
            ("llp.webp", im.convert("P").convert("RGB"), "WEBP", {"lossless": True}), ("lossy.webp", im, "WEBP", {"quality": 80}),
            ("lossya.webp", im.convert("RGBA"), "WEBP", {"quality": 60})]
    out = []
    from test_png_decode import _adam7_png
    with open(os.path.join(d, "adam7.png"), "wb") as f:
        f.write(_adam7_png(np.concatenate([a, a[..., :1]], -1), 6))
    out.append(os.path.join(d, "adam7.png"))
    for name, img, fmt, kw in todo:
        p = os.path.join(d, name)
        img.save(p, fmt, **kw)
        out.append(p)
    # round 3 parsers: run-length BMP (hand-made: PIL does not write them) and an animated WebP
    from test_png_decode import _rle_bmp
    idx8 = np.repeat(rng.integers(0, 200, (20, 8), dtype=np.uint8), 4, axis=1)
    for name, bpp, idx in (("rle8.bmp", 8, idx8), ("rle4.bmp", 4, (idx8 % 16).astype(np.uint8))):
        p = os.path.join(d, name)
        with open(p, "wb") as f:
            f.write(_rle_bmp(idx, bpp, rng.integers(0, 256, 3 * (200 if bpp == 8 else 16), dtype=np.uint8).tolist(), with_delta=True))
        out.append(p)
    p = os.path.join(d, "anim.webp")
    im.convert("RGBA").save(p, "WEBP", save_all=True, append_images=[im.convert("RGBA").rotate(180)], duration=40, lossless=True)
    out.append(p)
    return out


@pytest.fixture(scope="module")
def fuzz_bin(tmp_path_factory):
    """the sanitizer build of the decoder harness: compiled by the first test that runs it, reused by the others (20 s each time)"""
    return str(tmp_path_factory.mktemp("fuzzbin") / "fuzz")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ for the sanitizer build")
def test_decoders_survive_mutated_files(tmp_path, fuzz_bin):
    seeds = _seeds(str(tmp_path))
    env = dict(os.environ, IST_FUZZ_BIN=fuzz_bin, IST_FUZZ_REUSE="1")
    r = subprocess.run([os.path.join(ROOT, "tools", "run_fuzz.sh"), "400"] + seeds, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "0 crashes" in r.stdout
    decoded = int(r.stdout.split("fuzz:")[1].split("decoded")[0])
    assert decoded >= len(seeds)          # every unmutated seed decodes


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ for the sanitizer build")
def test_compiler_invariants_under_hostile_op_lists(tmp_path):
    """tools/fuzz_compile.cpp: NaN / infinite / huge transforms and rectangles, hostile caps through the planner; the
    clamp boxes, cells, bands and LDS sizes the kernel trusts must stay in range (or the list is refused)."""
    env = dict(os.environ, IST_FUZZ_BIN=str(tmp_path / "fuzzc"))
    r = subprocess.run([os.path.join(ROOT, "tools", "run_fuzz.sh"), "compile", "16000"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("all invariants hold") == 2


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_progressive_jpeg_gives_the_coefficients_of_its_sequential_twin(tmp_path, fuzz_bin):
    """PIL writes the same quantised coefficients whether a file is saved sequential or progressive; the host entropy
    decoder must recover the same planes from both (the GPU stages after it are shared)."""
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:203, 0:317]
    a = np.stack([(xx * 2 + yy) % 256, (yy * 3) % 256, ((xx + yy) * 2) % 256], -1) + rng.integers(-25, 25, (203, 317, 3))
    im = Image.fromarray(a.clip(0, 255).astype(np.uint8), "RGB")
    files = []
    for name, kw in (("444", {"subsampling": 0}), ("422", {"subsampling": 1}), ("420", {"subsampling": 2}), ("grey", None)):
        img = im.convert("L") if kw is None else im
        for tag, extra in (("b", {"optimize": True}), ("p", {"progressive": True})):
            p = str(tmp_path / ("%s_%s.jpg" % (name, tag)))
            img.save(p, "JPEG", quality=83, **dict(kw or {}, **extra))
            files.append(p)
    env = dict(os.environ, IST_FUZZ_BIN=fuzz_bin, IST_FUZZ_REUSE="1")
    r = subprocess.run([os.path.join(ROOT, "tools", "run_fuzz.sh"), "coefs"] + files, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rows = [l.split() for l in r.stdout.strip().splitlines()]
    assert len(rows) == 8 and all("error" not in l for l in r.stdout.splitlines())
    for b, p in zip(rows[0::2], rows[1::2]):
        assert b[-1] == p[-1], (b, p)                 # same coefficient hash
        assert b[2] == "scans=1" and int(p[2].split("=")[1]) > 1


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ for the sanitizer build")
def test_a_scan_with_too_few_restart_intervals_is_not_eligible_for_the_gpu_decoder(tmp_path, fuzz_bin):
    """a DRI scan that ends (EOI) at an RSTn boundary after k < n intervals must not reach the GPU entropy
    decoder - its block count is per interval, and the MCUs nobody writes would keep an earlier call's coefficients.  The
    harness aborts when an eligible scan's intervals do not tile the frame's MCUs; it runs that check on files the host
    decoder rejects too."""
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, (160, 240, 3), dtype=np.uint8)
    import io
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", quality=90, subsampling=2, restart_marker_rows=1)
    good = b.getvalue()
    marks = [i for i in range(good.find(b"\xff\xda"), len(good) - 1) if good[i] == 0xFF and 0xD0 <= good[i + 1] <= 0xD7]
    assert len(marks) == 9
    files = []
    for k, cut in enumerate((marks[0], marks[3], marks[3] + 2, marks[8], marks[8] + 2)):
        p = tmp_path / ("cut%d.jpg" % k)
        p.write_bytes(good[:cut] + b"\xff\xd9")
        files.append(str(p))
    (tmp_path / "whole.jpg").write_bytes(good)
    env = dict(os.environ, IST_FUZZ_BIN=fuzz_bin, IST_FUZZ_REUSE="1")
    r = subprocess.run([os.path.join(ROOT, "tools", "run_fuzz.sh"), "0", str(tmp_path / "whole.jpg")] + files, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "0 crashes" in r.stdout
