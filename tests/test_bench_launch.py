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
    assert line["extra"]["strip_assembled"] is True           # both splits: every part landed where the plan puts it
    assert "started 2 rank processes" in line["extra"]["launcher"]


def test_under_torchrun_the_existing_ranks_are_used():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", "29683", BENCH, "--gpus", "3", "--dry-launch", "--steps", "2", "--warmup", "1"], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 3 and len({s["pid"] for s in line["extra"]["ranks_seen"]}) == 3
    assert "launcher" not in line["extra"]


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
