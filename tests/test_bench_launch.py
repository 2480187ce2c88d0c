"""bench.py --gpus N must run N ranks however it is started: invoked plainly it starts the ranks
itself before touching torch/HIP; under torchrun the ranks exist already.  --dry-launch runs the same launch path on CPU
(gloo, a stub render that involves neither the GPU nor the oracle) and the line proves the rank count.  Reference anchor
of what is spread over the ranks: the independent per-image iterations, pages/index/index.js:1439-1554."""
import json
import os
import subprocess
import sys

from tests import util as U

BENCH = os.path.join(U.ROOT, "bench.py")


def _bench_free_port():
    """a port nobody holds right now (a hard-coded port fails when another run or a TIME_WAIT socket has it)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ist_bench_for_port", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.free_port()


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "IST_BENCH_CHILD")}


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout          # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


def test_plain_invocation_starts_the_ranks_itself():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch", "--steps", "3", "--warmup", "1"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["extra"]["world_size"] == 2
    seen = line["extra"]["ranks_seen"]
    assert sorted(s["rank"] for s in seen) == [0, 1]
    assert len({s["pid"] for s in seen}) == 2                 # two processes, not one rank counted twice
    assert line["extra"]["strip_assembled"] is True           # every cut: every part landed where the plan puts it
    assert "started 2 rank processes" in line["extra"]["launcher"]
    _check_scaling_keys(line)


def _check_scaling_keys(line):
    """the N > 1 line carries, per region, the N-rank time AND the one-GPU form of the same region timed by
    rank 0 alone in the same process; BASELINE configs[4] (64 images) has its own legs; configs[2]'s horizontal strip is cut by
    rows (full-width bands: nothing staged, host sink available)"""
    ex = line["extra"]
    assert {"resident/image", "resident/band", "resident/rows_horizontal"} <= set(ex["regions"])
    h = ex["regions"]["resident/rows_horizontal"]
    assert h["split"] == "rows" and h["bands_staged"] == 0 and h["host_sink_available"] is True
    assert ex["regions"]["resident/image"]["split"] == "image"
    for name in ("resident/image", "resident/band", "resident/rows_horizontal"):
        sc = ex["scaling"][name]
        assert set(sc) == {"ms_1gpu", "ms_Ngpu", "speedup"} and sc["ms_1gpu"] > 0 and sc["ms_Ngpu"] > 0
        assert abs(sc["speedup"] - sc["ms_1gpu"] / sc["ms_Ngpu"]) < 0.01 * max(1.0, sc["speedup"])
    assert ex["scaling"]["resident/image"]["ms_Ngpu"] == ex["regions"]["resident/image"]["ms_per_step"]
    assert ex["scaling"]["resident/image"]["ms_1gpu"] == ex["one_gpu_same_lease"]["vertical"]["resident"]["ms_per_step"]
    rank0 = [s["pid"] for s in ex["ranks_seen"] if s["rank"] == 0]
    assert ex["one_gpu_same_lease"]["timed_by_pids"] == rank0          # the comparators come from rank 0's own process
    c5 = ex["config5"]
    assert "resident/image" in c5["regions"] and "resident" in c5["one_gpu"] and set(c5["scaling"]["resident/image"]) == {"ms_1gpu", "ms_Ngpu", "speedup"}
    assert "host_in_host_out/band" in line["config"]["timed_region"] and "CANNOT scale" in line["config"]["timed_region"]


def test_under_torchrun_the_existing_ranks_are_used():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(_bench_free_port()), BENCH, "--gpus", "3", "--dry-launch", "--steps", "2", "--warmup", "1"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 3 and len({s["pid"] for s in line["extra"]["ranks_seen"]}) == 3
    assert "launcher" not in line["extra"]
    _check_scaling_keys(line)


def test_a_rank_count_that_does_not_match_gpus_is_refused():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_a_failing_rank_fails_the_launch():
    """no GPU here: every self-started rank refuses ('needs GPU r'), the parent reports it and exits non-zero without a line"""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--no-cpu"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "exited with status" in r.stderr
