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
