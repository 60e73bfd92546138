"""The benchmark entry points honour the launcher contract on CPU: one JSON line from rank 0 with the agreed keys,
under ``torch.distributed.run`` with several ranks (gloo), for the per-rank engine mode and the worker-pool mode."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config"}


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(script, nproc, *flags, timeout=600):
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / script), *flags]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 only, exactly one line
    return json.loads(lines[0])


def test_bench_under_torchrun_four_ranks_gloo():
    out = _run("bench.py", 4, "--cpu", "--gpus", "4", "--steps", "2", "--warmup", "1", "--pages-per-doc", "2")
    assert KEYS <= set(out)
    assert out["n_gpus"] == 4 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp4"
    # whole-job aggregate: pages of all ranks over the slowest rank's time
    assert out["value"] == pytest.approx(2 * 2 * 4 / (out["ms_per_step"] * 2 / 1000.0), rel=1e-3)
    assert "without restart markers" in out["data"]
    assert out["config"]["host_cpu_s_per_100_pages"] > 0          # CPU seconds of a rank per 100 pages (all its threads)


def test_bench_via_pool_under_torchrun():
    out = _run("bench.py", 2, "--cpu", "--via-pool", "--gpus", "2", "--steps", "1", "--warmup", "1", "--pages-per-doc", "3")
    assert KEYS <= set(out)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "pool2"
    # two front-end processes, each reaching both engine workers; rank 0's own documents: (warm-up + 1 step) x 3 pages
    pool = out["config"]["pool"]
    assert pool["workers"] == 2 and pool["shared_workers"] and pool["pages_done"] == 2 * 3 and pool["restarts"] == 0
    assert out["value"] == pytest.approx(2 * 3 / (out["ms_per_step"] / 1000.0), rel=1e-3)
    assert out["config"]["front_end_cpu_s_per_100_pages"] > 0


@pytest.mark.parametrize("script,flags", [
    ("bench/allreduce_bench.py", ("--cpu", "--steps", "2", "--warmup", "1", "--mb", "1")),
    ("bench/train_bench.py", ("--cpu", "--steps", "2", "--warmup", "1", "--batch", "8", "--width", "128")),
])
def test_auxiliary_benches_follow_the_same_contract(script, flags):
    out = _run(script, 2, "--gpus", "2", *flags)
    assert KEYS <= set(out)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
