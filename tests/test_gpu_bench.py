"""The bench contract on a real GPU (kept short): the N = 1 line carries what the driver and the judge read, and the
single-process device-group leg (what the N > 1 lines embed as extra.single_process_group) runs through ist_group_* on the
devices that exist.  Reference anchor of the workload: BASELINE configs[1], pages/index/index.js:1251-1581."""
import json
import os
import subprocess
import sys

import pytest

from tests import util as U

pytestmark = pytest.mark.gpu
BENCH = os.path.join(U.ROOT, "bench.py")


def _run(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    line = _run(["--quick", "--no-cpu", "--kernels-only", "--steps", "10", "--warmup", "2"])
    assert line["n_gpus"] == 1 and line["unit"] == "MP/s" and line["dtype"] == "u8" and line["higher_is_better"] is True
    assert "configs[1]" in line["config"]["workload"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0.3 < roof["frac"] <= 1.0
    assert roof["algorithmic_bytes_per_launch"] == 2 * 9 * 4032 * 3024 * 4
    assert abs(line["value"] - 109.734912 / (line["ms_per_step"] * 1e-3)) / line["value"] < 0.02


def test_single_process_group_leg_on_one_device():
    leg = _run(["--group-leg", "1", "--steps", "8"])
    assert leg["devices"] == [0] and leg["checked"] is True
    for key in ("resident/image", "resident/band", "host_in_host_out/image", "host_in_host_out/band"):
        assert leg["regions"][key]["MPs"] > 0
    assert leg["checked_horizontal"] is True
    h = leg["regions"]["host_in_host_out/rows_horizontal"]
    assert h["MPs"] > 0 and "error" not in h
    c5 = leg["config5"]                                   # BASELINE configs[4] through the group, on the devices that exist
    assert "error" not in c5
    if "skipped" not in c5:
        assert c5["resident/image"]["checked"] is True and c5["host_in_host_out/image"]["checked"] is True


def _errors(node, path=""):
    """every {"error": ...} object below `node` (bench.py swallows an informational leg's exception into one)"""
    found = []
    if isinstance(node, dict):
        if "error" in node:
            found.append((path, node["error"]))
        for k, v in node.items():
            found += _errors(v, path + "/" + str(k))
    elif isinstance(node, list):
        for i, v in enumerate(node):
            found += _errors(v, path + "/" + str(i))
    return found


def test_no_informational_leg_of_the_full_line_failed():
    """the informational legs catch their exceptions so that the headline survives; here the FULL line (no
    --kernels-only) runs once and none of them may have failed; the file pipeline reports its four stages with their bounds; the
    BASELINE configs[4] leg ran at its own size and made the strip."""
    line = _run(["--no-cpu", "--steps", "20", "--warmup", "2"], timeout=1500)
    assert _errors(line) == []
    ex = line["extra"]
    for key in ("uniform_vertical", "uniform_horizontal", "mixed_vertical", "mixed_horizontal", "regions", "end_to_end_host_path", "file_pipeline", "config5_single_gpu",
                "canvas_rows_padded_to_4KiB"):
        assert key in ex, key
    assert ex["uniform_vertical"]["rows_walked"].startswith("flat form") and ex["mixed_horizontal"]["rows_walked"] == "the canvas's rows"
    assert 0.3 < ex["canvas_rows_padded_to_4KiB"]["mixed_horizontal"]["frac"] <= 1.0 and ex["canvas_rows_padded_to_4KiB"]["mixed_horizontal"]["canvas_row_bytes"] % 4096 == 0
    assert set(ex["file_pipeline"]["stage_rooflines"]) == {"entropy_gpu", "reconstruct", "stitch", "png"}
    for st in ex["file_pipeline"]["stage_rooflines"].values():
        assert st["bound_ms"] > 0 and st["achieved_ms"] > 0 and 0 < st["frac"] <= 1.5
    c5 = ex["config5_single_gpu"]
    assert "skipped" not in c5 and c5["canvas"] == [8000, 384000] and c5["strip_is_the_images_in_order"] is True
    assert c5["algorithmic_bytes"] == 2 * 64 * 8000 * 6000 * 4 and 0.3 < c5["frac"] <= 1.0
    assert set(ex["regions"]) >= {"resident", "from_pinned_host", "host_in_host_out", "from_jpeg"}
    assert line["d2d_copy_yardstick"]["GBs"] > 0


def test_the_sharded_line_on_a_world_of_one():
    """the N > 1 code path (run_sharded: regions, same-lease one-GPU comparators, configs[4] legs, horizontal strip by rows) with
    WORLD_SIZE = 1 on the GPU that exists - everything but the cross-device transfer"""
    env = dict(os.environ, IST_BENCH_FORCE_SHARDED="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29671")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "20", "--warmup", "2", "--no-cpu", "--kernels-only"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert _errors(line) == []
    ex = line["extra"]
    want = {"resident/image", "from_pinned_host/image", "host_in_host_out/image", "from_jpeg/image", "resident/band", "from_pinned_host/band", "host_in_host_out/band",
            "resident/rows_horizontal", "host_in_host_out/rows_horizontal"}
    assert want <= set(ex["regions"]) and want <= set(ex["scaling"])
    for sc in ex["scaling"].values():
        # (one rank against itself; with 20 steps the contract's barrier + all-reduce bracket is a visible part of the sub-millisecond regions)
        assert sc["ms_1gpu"] > 0 and sc["ms_Ngpu"] > 0 and abs(sc["speedup"] - sc["ms_1gpu"] / sc["ms_Ngpu"]) <= 0.01 * max(1.0, sc["speedup"])
    for name in ("host_in_host_out/image", "host_in_host_out/band", "host_in_host_out/rows_horizontal", "from_pinned_host/image"):
        assert 0.6 < ex["scaling"][name]["speedup"] < 1.6, (name, ex["scaling"][name])              # PCIe-bound regions: the same bytes over the same link
    assert ex["gather"]["rows_horizontal"]["split"] == "rows" and ex["gather"]["rows_horizontal"]["host_sink_available"] is True
    c5 = ex["config5"]
    assert "skipped" not in c5 and {"resident/image", "host_in_host_out/image"} <= set(c5["regions"]) <= set(c5["scaling"]) | set(c5["regions"])
    assert set(c5["scaling"]) == {"resident/image", "host_in_host_out/image"}
    assert "CANNOT scale" in line["config"]["timed_region"]


def test_the_sharded_line_without_rccl_falls_back_to_gloo():
    """SURVEY 8e fallback: when RCCL cannot be initialised the ranks keep a gloo control plane (timing brackets, flags on CPU tensors), the
    regions that exchange nothing still run, and the line says which collectives it used.  --force-gloo rehearses it on a world of one."""
    env = dict(os.environ, IST_BENCH_FORCE_SHARDED="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29674")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "10", "--warmup", "2", "--no-cpu", "--kernels-only", "--no-config5", "--force-gloo"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["extra"]["collectives"].startswith("gloo") and "forced by --force-gloo" in line["extra"]["collectives"]
    assert _errors(line) == []
    assert line["extra"]["regions"]["host_in_host_out/band"]["ms_per_step"] > 0 and line["value"] > 0
