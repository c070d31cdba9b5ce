"""bench.py --gpus N must start its N ranks itself when no launcher did, and must refuse when the GPUs are not there
(round-1 finding: `python bench.py --gpus 8` silently ran one rank and printed n_gpus: 1)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_self_launch_starts_the_ranks_over_gloo():
    """The self-launch path end to end without a GPU: two rank processes rendezvous over gloo and the line says n_gpus 2."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--launch-check"], env=_env(), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["launch_check"] and d["n_gpus"] == 2 and d["nranks"] == 2 and d["sum"] == 2.0


def test_more_gpus_than_visible_is_refused():
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("a box with 64 GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "refusing" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())


@pytest.mark.gpu
def test_two_rank_rehearsal_on_one_gpu_reports_the_exchange():
    """Both ranks on cuda:0, collectives over gloo: the whole N > 1 path of bench.py through the self-launch, with the exchange
    block on the line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--single-device", "--steps", "3", "--warmup", "1",
                        "--config", "2", "--no-extras"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["exchange"]["nranks"] == 2
    assert d["exchange"]["compute_ms"] > 0 and d["exchange"]["allreduce_bytes_per_step"] == 44 * 100_000
