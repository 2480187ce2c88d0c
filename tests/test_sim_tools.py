"""The CPU restatements DESIGN.md quotes numbers from must keep building and running: tools/sim_slot_sync.cpp (how often a
wrong start state survives a subsequence of the GPU Huffman decoder) and tools/sim_code_lengths.py (the PNG encoder's code
rule against Huffman's; its invariants are in test_png_code_rule.py)."""
import io
import os
import re
import shutil
import subprocess

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_slot_sync_simulation_builds_and_runs(tmp_path):
    exe = str(tmp_path / "sim_slot_sync")
    csrc = os.path.join(ROOT, "imagestitching_amd", "csrc")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + csrc, "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "sim_slot_sync.cpp"),
                    os.path.join(csrc, "ist_jpeg.cpp"), "-o", exe, "-lpthread"], check=True, capture_output=True, timeout=300)
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:480, 0:640]
    a = np.stack([128 + 90 * np.sin(xx / 37.0 + yy / 91.0), 128 + 80 * np.cos(xx / 53.0 - yy / 29.0), 100 + 0.1 * xx + 0.1 * yy], -1) + rng.normal(0, 3.0, (480, 640, 3))
    b = io.BytesIO()
    Image.fromarray(a.clip(0, 255).astype(np.uint8)).save(b, "JPEG", quality=90, subsampling=2)
    (tmp_path / "p.jpg").write_bytes(b.getvalue())
    out = subprocess.run([exe, str(tmp_path / "p.jpg")], check=True, capture_output=True, text=True, timeout=120).stdout
    # round 4: the replay of the kernel's pass iteration, one start hypothesis per subsequence against two
    m = re.search(r"one hypothesis: worst (\d+), two hypotheses: worst (\d+)", out)
    assert m and 1 <= int(m.group(1)) <= 40 and 1 <= int(m.group(2)) <= 40, out
    rows = re.findall(r"exit true ([0-9.]+) \| slot wrong only ([0-9.]+) \| elsewhere ([0-9.]+)", out)
    assert len(rows) == 2, out
    for r in rows:
        f = [float(x) for x in r]
        assert abs(sum(f) - 1.0) < 0.01 and 0.3 < f[0] < 0.95, out      # most wrong starts fall into step; a real share does not
