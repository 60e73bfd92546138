"""Direct all-reduce over HIP IPC peer buffers (parallel/direct.py, csrc/kernels/direct_allreduce.hip).

The GPU tests run N ranks as N processes that all use cuda:0 (the test box has one GPU): buffers still
travel through hipIpcGetMemHandle / hipIpcOpenMemHandle and the kernels of different processes
synchronise through flags in each other's memory, exactly as across xGMI.  The process group that
carries the handles is gloo (RCCL refuses two ranks on one device).  Oracle: the same sum in plain
PyTorch fp32, in rank order.
"""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port, **env):
    """One-GPU box: every rank uses cuda:0 and the handles travel over gloo (RCCL refuses two ranks on one device).
    With DLAMD_TEST_MULTI_GPU=1 (tests/test_multigpu.py, needs >= world devices): one rank per device over RCCL - the
    layout the kernels are built for.  Returns the rank's device index."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    multi = os.environ.get("DLAMD_TEST_MULTI_GPU") == "1"
    dev = rank if multi else 0
    torch.cuda.set_device(dev)
    if multi:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dev


def _oracle(parts, scale, dtype):
    acc = torch.zeros_like(parts[0], dtype=torch.float32)
    for p in parts:
        acc += p.float()
    return (acc * scale).to(dtype)


def _rank_main(rank, world, port, out_dir, starve):
    dev = _init(rank, world, port)
    from distributed_llms_amd.parallel.direct import DirectAllReduce

    ar = DirectAllReduce(max_bytes=8 << 20, blocks=32)
    res = {"uncached_flags": ar.flags_uncached, "cases": []}
    g = torch.Generator().manual_seed(1234)
    cases = [(torch.float32, 4096, "one_shot"), (torch.float32, 4096, "two_shot"), (torch.bfloat16, 8, "auto"),
             (torch.bfloat16, 1 << 20, "one_shot"), (torch.bfloat16, (1 << 20) + 8 * 37, "two_shot"),
             (torch.float32, (2 << 20) - 4, "auto"), (torch.bfloat16, 4 << 20, "auto")]
    for rep in range(3):                       # repeats reuse both slots and both flag phases
        for dtype, n, algo in cases:
            parts = [(torch.randn(n, generator=g) * (p + 1)).to(dtype) for p in range(world)]   # same stream on every rank
            x = parts[rank].cuda()
            scale = 1.0 / world if rep == 1 else 1.0
            y = ar.all_reduce_(x, scale=scale, algo=algo)
            torch.cuda.synchronize()
            want = _oracle(parts, scale, dtype)
            got = y.cpu()
            res["cases"].append((str(dtype), n, algo, rep, bool(torch.equal(got, want)),
                                 float((got.float() - want.float()).abs().max())))
    # out-of-place on a side stream, ordered by events like the DDP hook does
    st = torch.cuda.Stream()
    parts = [torch.full((4096,), float(p + 1)) for p in range(world)]
    x, out = parts[rank].cuda(), torch.empty(4096, device="cuda")
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        ar.all_reduce_(x, out=out)
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    res["side_stream"] = bool(torch.equal(out.cpu(), _oracle(parts, 1.0, torch.float32)))
    res["status"] = ar.status()
    if starve:
        # rank 1 skips a call: rank 0's bounded spin must run out, set the status word and let the grid drain
        dist.barrier()
        if rank == 0:
            ar.set_spin_limit(20000)
            ar.all_reduce_(torch.ones(4096, device="cuda"))
            torch.cuda.synchronize()
            res["starved_status"] = ar.status()
        dist.barrier()
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(world, starve=False):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rank_main, args=(world, _free_port(), d, starve), nprocs=world, join=True)
        return [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(world)]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 3])
def test_direct_allreduce_matches_fp32_oracle(world):
    res = _spawn(world)
    for r in res:
        bad = [c for c in r["cases"] if not c[4]]
        assert not bad, bad[:4]
        assert r["side_stream"] and r["status"] == 0
    assert len(res[0]["cases"]) == 21


@pytest.mark.gpu
def test_direct_allreduce_bounded_spin_reports_missing_peer():
    res = _spawn(2, starve=True)
    assert res[0]["status"] == 0 and res[0]["starved_status"] == 1


def _region_rank(rank, world, port, out_dir):
    dev = _init(rank, world, port, DLAMD_DAR_SPIN_LIMIT=str(1 << 22))
    from distributed_llms_amd.parallel.direct import DirectAllReduce, DirectAllReduceError

    ar = DirectAllReduce(max_bytes=4096, blocks=32)
    # one-shot holds one 16-byte vector per thread: 32 blocks x 512 threads = 256 KiB at most
    cases = [(torch.float32, 4096, "one_shot"), (torch.float32, 4096, "two_shot"), (torch.bfloat16, 8, "auto"),
             (torch.bfloat16, 1 << 17, "one_shot"), (torch.bfloat16, (1 << 20) + 8 * 37, "two_shot"),
             (torch.float32, (2 << 20) - 4, "auto"), (torch.bfloat16, 4 << 20, "auto")]
    bufs = [ar.alloc(n, dtype) for dtype, n, _ in cases]
    ids = ar.register(bufs[:3]) + ar.register(bufs[3:])          # regions can be added later
    res = {"ids": ids, "zeroed": all(bool((b == 0).all()) for b in bufs), "cases": []}
    g = torch.Generator().manual_seed(4321)
    for rep in range(3):
        for (dtype, n, algo), buf in zip(cases, bufs):
            # two reductions back to back with no host synchronisation: the refill of the second runs as soon as this
            # rank's first kernel has ended, i.e. only after every peer has read the first input
            parts1 = [(torch.randn(n, generator=g) * (p + 1)).to(dtype) for p in range(world)]
            parts2 = [(torch.randn(n, generator=g) - p).to(dtype) for p in range(world)]
            x1, x2 = parts1[rank].cuda(), parts2[rank].cuda()
            first = torch.empty_like(buf)
            scale = 1.0 / world if rep == 1 else 1.0
            buf.copy_(x1)
            ar.all_reduce_region_(buf, scale=scale, algo=algo)
            first.copy_(buf)
            buf.copy_(x2)
            ar.all_reduce_region_(buf, scale=scale, algo=algo)
            torch.cuda.synchronize()
            ok = torch.equal(first.cpu(), _oracle(parts1, scale, dtype)) and torch.equal(buf.cpu(), _oracle(parts2, scale, dtype))
            res["cases"].append((str(dtype), n, algo, rep, bool(ok)))
    res["status"] = ar.status()
    try:
        ar.all_reduce_region_(torch.zeros(64, device="cuda"))
        res["unregistered"] = "accepted"
    except DirectAllReduceError:
        res["unregistered"] = "refused"
    try:
        ar.all_reduce_region_(bufs[4], algo="one_shot")           # larger than the one-shot form can hold
        res["oversized_one_shot"] = "accepted"
    except Exception:  # noqa: BLE001
        res["oversized_one_shot"] = "refused"
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_registered_buffers_are_reduced_in_place(world):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_region_rank, args=(world, _free_port(), d), nprocs=world, join=True)
        res = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(world)]
    for r in res:
        assert r["ids"] == list(range(7)) and r["zeroed"]
        bad = [c for c in r["cases"] if not c[4]]
        assert not bad, bad[:4]
        assert r["status"] == 0 and r["unregistered"] == "refused" and r["oversized_one_shot"] == "refused"
    assert len(res[0]["cases"]) == 21


def _ddp_rank(rank, world, port, out_dir, mode):
    dev = _init(rank, world, port)
    from distributed_llms_amd.parallel.ddp import BucketedDDP

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(),
                              torch.nn.Linear(256, 8)).cuda()
    ddp = BucketedDDP(net, bucket_mb=0.1, mode=mode.split("_")[0], register_buckets=mode != "direct_staged")
    assert len(ddp.flat.buckets) > 1
    if ddp._direct is not None:
        # registered: every gradient bucket is a region of the all-reduce (reduced where autograd wrote it)
        regions = [ddp._direct.region_of(b.flat_grad) for b in ddp.flat.buckets]
        assert (regions == list(range(len(regions)))) == (mode == "direct") and ddp.registered == (mode == "direct")
    g = torch.Generator().manual_seed(7 + rank)
    for _ in range(3):
        ddp.zero_grad()
        x = torch.randn(32, 64, generator=g).cuda()
        net(x).square().mean().backward()
        ddp.finish_backward()
        with torch.no_grad():
            for b in ddp.flat.buckets:
                b.flat_param.add_(b.flat_grad, alpha=-0.1)
    torch.cuda.synchronize()
    torch.save({"flat": torch.cat([b.flat_param for b in ddp.flat.buckets]).cpu(), "mode": ddp.mode,
                "calls": ddp._direct.calls if ddp._direct is not None else 0,
                "status": ddp._direct.status() if ddp._direct is not None else 0}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_bucketed_ddp_direct_mode_equals_allreduce_mode():
    out = {}
    for mode in ("direct", "direct_staged", "allreduce"):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_ddp_rank, args=(2, _free_port(), d, mode), nprocs=2, join=True)
            out[mode] = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(2)]
    for mode in ("direct", "direct_staged"):
        assert out[mode][0]["mode"] == "direct" and out[mode][0]["status"] == 0 and out[mode][0]["calls"] > 0
        assert torch.equal(out[mode][0]["flat"], out[mode][1]["flat"]), "ranks must stay bitwise in step"
        torch.testing.assert_close(out[mode][0]["flat"], out["allreduce"][0]["flat"], rtol=1e-5, atol=1e-6)
    assert torch.equal(out["direct"][0]["flat"], out["direct_staged"][0]["flat"]), "same kernel, same summation order"


def test_direct_mode_falls_back_on_cpu_groups():
    """CPU process groups have no IPC buffers: mode='direct' must quietly become 'allreduce'."""
    from distributed_llms_amd.parallel.ddp import BucketedDDP

    net = torch.nn.Linear(8, 8)
    ddp = BucketedDDP(net, mode="direct")
    assert ddp.mode == "allreduce" and ddp._direct is None


def test_supported_predicate():
    from distributed_llms_amd.parallel.direct import supported

    assert not supported(torch.zeros(8))                      # CPU tensor


def _trainer_rank(rank, world, port, out_dir, mode):
    dev = _init(rank, world, port)
    from distributed_llms_amd.models.crnn import CRNN
    from distributed_llms_amd.parallel.comm import DistEnv
    from distributed_llms_amd.train.trainer import CTCTrainer, TrainConfig

    torch.manual_seed(0)
    model = CRNN(hidden=64, widths=(8, 16, 16, 16, 32, 32))
    cfg = TrainConfig(batch_size=8, width=96, lr=3e-3, vocab=3, max_chars=5, font_px=28, warmup_steps=0, bucket_mb=0.05,
                      dp_mode=mode, amp=False)
    tr = CTCTrainer(model, cfg, device=f"cuda:{dev}",
                    env=DistEnv(rank, world, dev, torch.device(f"cuda:{dev}"),
                                "nccl" if os.environ.get("DLAMD_TEST_MULTI_GPU") == "1" else "gloo"))
    losses = []
    for s in range(4):
        losses.append(tr.train_step(*tr.synthetic_batch(s)))
        print(f"[dp-trainer {mode}] rank {rank} step {s} loss {losses[-1]:.4f}", flush=True)
    torch.cuda.synchronize()
    torch.save({"flat": torch.cat([b.flat_param for b in tr.ddp.flat.buckets]).cpu(), "losses": losses, "mode": tr.ddp.mode,
                "buckets": len(tr.ddp.flat.buckets)}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_ctc_trainer_dp2_direct_allreduce_matches_process_group_allreduce():
    """Whole training steps (fused LSTM recurrences, CTC kernel, flat AdamW) with gradients averaged by the IPC kernel."""
    out = {}
    for mode in ("direct", "allreduce"):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_trainer_rank, args=(2, _free_port(), d, mode), nprocs=2, join=True)
            out[mode] = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(2)]
    assert out["direct"][0]["mode"] == "direct" and out["direct"][0]["buckets"] > 1
    assert torch.equal(out["direct"][0]["flat"], out["direct"][1]["flat"]), "ranks must hold identical parameters"
    # two separate runs: per-rank gradients are not bitwise reproducible (float atomics in the CTC kernel, MIOpen's
    # algorithm choice), and Adam turns a sign flip of a near-zero gradient into an lr-sized step - compare in norm
    a, b = out["direct"][0]["flat"], out["allreduce"][0]["flat"]
    assert float((a - b).norm() / b.norm()) < 2e-3
    assert float(((a - b).abs() > 2e-3).float().mean()) < 1e-3
    assert all(l == l for l in out["direct"][0]["losses"])      # finite
    for l1, l2 in zip(out["direct"][0]["losses"], out["allreduce"][0]["losses"]):
        assert abs(l1 - l2) < 0.05 * max(1.0, abs(l2))


def _late_peer_rank(rank, world, port, out_dir):
    dev = _init(rank, world, port)
    import time

    from distributed_llms_amd.parallel.ddp import BucketedDDP
    from distributed_llms_amd.parallel.direct import DirectAllReduceError

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8)).cuda()
    ddp = BucketedDDP(net, mode="direct")
    assert ddp.mode == "direct"
    x = torch.randn(16, 64, device="cuda")
    # a normal step first: both ranks arrive together, the check passes
    net(x).sum().backward()
    ddp.finish_backward()
    ddp.zero_grad()
    dist.barrier()
    # now the barriers give up after a few thousand polls (microseconds) and rank 1 is a second late: rank 0's
    # kernel leaves with its status word set (it still exits normally - nothing hangs) and the step must raise
    ddp._direct.set_spin_limit(2000)
    if rank == 1:
        time.sleep(1.0)
    net(x).sum().backward()
    raised = False
    try:
        ddp.finish_backward()
    except DirectAllReduceError:
        raised = True
    torch.cuda.synchronize()
    with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
        f.write("raised" if raised else "ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_bucketed_ddp_raises_when_a_peer_misses_the_barrier():
    """A rank that arrives after the bounded spin has run out must fail the step on the rank that waited: silently
    proceeding would feed unreduced gradients to the optimizer and leave the ranks out of sync for good."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_late_peer_rank, args=(2, _free_port(), d), nprocs=2, join=True)
        got = [open(os.path.join(d, f"r{r}.txt")).read() for r in range(2)]
    assert got[0] == "raised", got
