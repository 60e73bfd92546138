"""Tests that need two or more GPUs: one rank per device, RCCL as the process group, HIP IPC across xGMI.

They skip on a one-GPU box and run unchanged on a multi-GPU node.  The rank functions are the ones of
tests/test_direct_allreduce.py, switched (DLAMD_TEST_MULTI_GPU=1) from "N processes on cuda:0 over gloo" to "rank r on
cuda:r over RCCL"; the oracles are the same fp32 sums.  The worker-pool test gives every engine process its own device
and kills one of them mid-document."""
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
N_GPUS = torch.cuda.device_count() if torch.cuda.is_available() else 0

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(N_GPUS < 2, reason="needs at least two GPUs (one rank per device)")]


@pytest.fixture()
def multi_gpu_env(monkeypatch):
    monkeypatch.setenv("DLAMD_TEST_MULTI_GPU", "1")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _worlds():
    return [w for w in (2, 4, 8) if w <= max(N_GPUS, 2)]


@pytest.mark.parametrize("world", _worlds())
def test_direct_allreduce_across_devices_matches_fp32_oracle(world, multi_gpu_env):
    import test_direct_allreduce as t

    res = t._spawn(world)
    for r in res:
        bad = [c for c in r["cases"] if not c[4]]
        assert not bad, bad[:4]
        assert r["side_stream"] and r["status"] == 0
    assert len(res[0]["cases"]) == 21


@pytest.mark.parametrize("world", _worlds())
def test_registered_buckets_across_devices_are_reduced_in_place(world, multi_gpu_env):
    import test_direct_allreduce as t

    with tempfile.TemporaryDirectory() as d:
        mp.spawn(t._region_rank, args=(world, t._free_port(), d), nprocs=world, join=True)
        res = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(world)]
    for r in res:
        assert r["ids"] == list(range(7)) and r["zeroed"]
        assert not [c for c in r["cases"] if not c[4]]
        assert r["status"] == 0 and r["unregistered"] == "refused" and r["oversized_one_shot"] == "refused"


def test_bucketed_ddp_direct_equals_rccl_allreduce(multi_gpu_env):
    import test_direct_allreduce as t

    out = {}
    for mode in ("direct", "direct_staged", "allreduce"):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(t._ddp_rank, args=(2, t._free_port(), d, mode), nprocs=2, join=True)
            out[mode] = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(2)]
    for mode in ("direct", "direct_staged"):
        assert out[mode][0]["mode"] == "direct" and out[mode][0]["status"] == 0 and out[mode][0]["calls"] > 0
        assert torch.equal(out[mode][0]["flat"], out[mode][1]["flat"]), "ranks must stay bitwise in step"
        torch.testing.assert_close(out[mode][0]["flat"], out["allreduce"][0]["flat"], rtol=1e-5, atol=1e-6)
    assert torch.equal(out["direct"][0]["flat"], out["direct_staged"][0]["flat"])


def test_ctc_trainer_dp2_over_rccl(multi_gpu_env):
    import test_direct_allreduce as t

    out = {}
    for mode in ("direct", "allreduce"):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(t._trainer_rank, args=(2, t._free_port(), d, mode), nprocs=2, join=True)
            out[mode] = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(2)]
    assert torch.equal(out["direct"][0]["flat"], out["direct"][1]["flat"])
    a, b = out["direct"][0]["flat"], out["allreduce"][0]["flat"]
    assert float((a - b).norm() / b.norm()) < 2e-3


def _torchrun(script, nproc, *flags, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(__import__("test_direct_allreduce")._free_port()), str(ROOT / script), *flags]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    return [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("mb", [1, 20, 64, 668])
def test_allreduce_bench_direct_against_rccl(mb):
    """Direct (all-links) kernel against RCCL for a 1 MB latency-bound call, a CRNN-sized gradient (20 MB), one
    bucket (64 MB) and a TrOCR-base-sized gradient (668 MB); the numbers land in the JSON line (per algorithm: time and
    bus bandwidth), the test only checks that every path ran - their results are compared by the oracle tests above."""
    n = min(N_GPUS, 8)
    out = _torchrun("bench/allreduce_bench.py", n, "--gpus", str(n), "--steps", "10", "--warmup", "3", "--mb", str(mb))
    assert len(out) == 1 and out[0]["n_gpus"] == n and out[0]["value"] > 0
    per = out[0]["config"]["ms_per_algorithm"]
    assert {"rccl", "direct_two_shot", "direct_in_place"} <= set(per) and all(v > 0 for v in per.values()), per


def test_bench_scales_over_devices():
    n = min(N_GPUS, 8)
    out = _torchrun("bench.py", n, "--gpus", str(n), "--steps", "3", "--warmup", "1", "--pages-per-doc", "36")
    assert len(out) == 1 and out[0]["n_gpus"] == n and out[0]["config"]["parallelism"] == f"dp{n}"
    pool = _torchrun("bench.py", n, "--via-pool", "--gpus", str(n), "--steps", "3", "--warmup", "1", "--pages-per-doc", "36")
    assert pool[0]["config"]["pool"]["workers"] == n and pool[0]["config"]["pool"]["restarts"] == 0


def test_worker_pool_over_two_devices_survives_a_crash():
    """One engine process per device; worker 0 dies after its first shard (fault injection exits the process from the
    host side, no kernel is disturbed); the document still completes on the surviving device and worker 0 is replaced."""
    import time

    from distributed_llms_amd.pdf import synth
    from distributed_llms_amd.runtime.workers import WorkerPool

    wd = ROOT / "weights"
    spec = {"model": "full", "cfg": dict(max_batch_pages=4), "weights_dir": str(wd) if wd.exists() else None}
    rendered = [synth.render_page(700 + i)[0] for i in range(2)]
    pdf = synth.make_scanned_pdf([rendered[i % 2] for i in range(8)], codec="jpeg")
    pool = WorkerPool(["cuda:0", "cuda:1"], spec, fault_inject="crash_after_tasks=1", max_wait_ms=5.0, min_pages_per_shard=4,
                      doc_deadline_s=300.0).start(300.0)
    try:
        first = pool.submit(pdf, 8).result(timeout=300)
        assert sorted(first) == list(range(8))
        second = pool.submit(pdf, 8).result(timeout=300)         # worker 0 is gone or restarting: re-queued / respawned
        assert sorted(second) == list(range(8))
        assert [len(first[i].lines) for i in range(8)] == [len(second[i].lines) for i in range(8)]
        t_end = time.time() + 120
        while pool.describe()["restarts"] < 1 and time.time() < t_end:
            time.sleep(0.2)
        assert pool.describe()["restarts"] >= 1
    finally:
        pool.stop()
