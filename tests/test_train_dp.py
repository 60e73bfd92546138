"""Data-parallel training over a 2-rank gloo group on CPU:
bitwise DP=2-with-identical-shards == DP=1, bucketed all-reduce vs reduce-scatter/all-gather with a
sharded optimizer, checkpoint / resume."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from distributed_llms_amd.models.crnn import CRNN
from distributed_llms_amd.parallel.comm import DistEnv
from distributed_llms_amd.parallel.ddp import BucketedDDP, FlatBuckets
from distributed_llms_amd.train.trainer import CTCTrainer, TrainConfig

CFG = dict(batch_size=4, width=96, lr=3e-3, vocab=3, max_chars=5, font_px=28, warmup_steps=0, bucket_mb=0.05)


def _tiny():
    torch.manual_seed(0)
    return CRNN(hidden=64, widths=(8, 16, 16, 16, 32, 32))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_rank(rank, world, port, mode, out_dir, identical):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        env = DistEnv(rank, world, rank, torch.device("cpu"), "gloo")
        tr = CTCTrainer(_tiny(), TrainConfig(dp_mode=mode, **CFG), device="cpu", env=env)
        assert len(tr.ddp.flat.buckets) > 1, "test must exercise several buckets"
        losses = []
        for step in range(3):
            if identical:
                batch = tr.data.batch(step, CFG["batch_size"], 0, 1)[:4]      # every rank sees the same samples
            else:
                batch = tr.synthetic_batch(step)                               # rank-specific slice of the global batch
            losses.append(tr.train_step(*batch))
        # parameter values in model order (bucket padding depends on the world size, the parameters do not)
        flat = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()])
        torch.save({"flat": flat, "losses": losses, "comm_calls": tr.ddp.comm_calls}, os.path.join(out_dir, f"r{rank}.pt"))
        tr.save_checkpoint(os.path.join(out_dir, "ckpt"))
        dist.barrier()
        tr2 = CTCTrainer(_tiny(), TrainConfig(dp_mode=mode, **CFG), device="cpu", env=env)
        assert tr2.load_checkpoint(os.path.join(out_dir, "ckpt")) == 3
        b = tr.data.batch(7, CFG["batch_size"], 0, 1)[:4]
        l1, l2 = tr.train_step(*b), tr2.train_step(*b)
        assert l1 == l2, (l1, l2)
        f1 = torch.cat([x.flat_param for x in tr.ddp.flat.buckets])
        f2 = torch.cat([x.flat_param for x in tr2.ddp.flat.buckets])
        assert torch.equal(f1, f2), "resume must reproduce the next step bitwise"
    finally:
        dist.destroy_process_group()


def _single(identical, world=2):
    torch.set_num_threads(1)
    tr = CTCTrainer(_tiny(), TrainConfig(**{**CFG, "batch_size": CFG["batch_size"] * (1 if identical else world)}), device="cpu")
    losses = []
    for step in range(3):
        if identical:
            batch = tr.data.batch(step, CFG["batch_size"], 0, 1)[:4]
        else:
            # the global batch = concatenation of both ranks' slices
            parts = [tr.data.batch(step, CFG["batch_size"], r, world) for r in range(world)]
            batch = (torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]),
                     torch.cat([p[2] for p in parts]), torch.cat([p[3] for p in parts]))
        losses.append(tr.train_step(*batch))
    return torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()]), losses


@pytest.mark.parametrize("mode", ["allreduce", "rs_ag"])
def test_dp2_identical_shards_is_bitwise_dp1(mode):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_run_rank, args=(2, _free_port(), mode, d, True), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(d, "r0.pt")), torch.load(os.path.join(d, "r1.pt"))
    ref, ref_losses = _single(True)
    assert torch.equal(r0["flat"], r1["flat"]), "ranks must hold identical parameters"
    n = ref.numel()
    assert torch.equal(r0["flat"][:n], ref), "DP=2 on identical shards must equal DP=1 bitwise"
    assert r0["losses"] == ref_losses
    assert r0["comm_calls"] >= 3 * 2


def test_dp2_different_shards_matches_global_batch():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_run_rank, args=(2, _free_port(), "allreduce", d, False), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(d, "r0.pt")), torch.load(os.path.join(d, "r1.pt"))
    assert torch.equal(r0["flat"], r1["flat"])
    ref, ref_losses = _single(False)
    # BatchNorm statistics are per rank (local batch), so this is close but not bitwise
    rel = (r0["flat"][: ref.numel()] - ref).norm() / ref.norm()
    assert rel < 0.05, rel
    assert abs(0.5 * (r0["losses"][0] + r1["losses"][0]) - ref_losses[0]) < 0.05 * abs(ref_losses[0])


def test_flat_buckets_views_and_grad_reattach():
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    before = [p.detach().clone() for p in m.parameters()]
    fb = FlatBuckets(m.parameters(), bucket_bytes=64)
    assert len(fb.buckets) >= 2
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p, b)
    m(torch.randn(2, 5)).sum().backward()
    for bk in fb.buckets:
        for p, o in zip(bk.params, bk.offsets):
            assert p.grad.data_ptr() == bk.flat_grad.data_ptr() + 4 * o      # grads accumulate inside the flat buffer
            assert torch.equal(bk.flat_grad[o : o + p.numel()].view_as(p), p.grad)
    for p in m.parameters():
        p.grad = None
    fb.zero_grad()
    m(torch.randn(2, 5)).sum().backward()
    assert all(float(bk.flat_grad.abs().sum()) > 0 for bk in fb.buckets)
    ddp = BucketedDDP(torch.nn.Linear(3, 3))
    ddp.finish_backward()   # world size 1: a no-op


def test_flat_buckets_take_gradient_memory_from_the_given_allocator():
    """``grad_alloc`` is how the direct all-reduce hands out IPC-exportable memory for the gradient buckets: every
    bucket's gradient must be exactly the tensor the allocator returned, and the parameters' ``.grad`` views of it."""
    handed_out = []

    def alloc(numel, dtype):
        t = torch.zeros(numel, dtype=dtype)
        handed_out.append(t)
        return t

    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    fb = FlatBuckets(m.parameters(), bucket_bytes=64, pad_to=16, grad_alloc=alloc)
    assert len(handed_out) == len(fb.buckets) >= 2
    for bk, t in zip(fb.buckets, handed_out):
        assert bk.flat_grad is t and t.numel() % 16 == 0 and t.numel() == bk.flat_param.numel()
    m(torch.randn(2, 5)).sum().backward()
    for bk, t in zip(fb.buckets, handed_out):
        assert float(t.abs().sum()) > 0
        for p, o in zip(bk.params, bk.offsets):
            assert p.grad.data_ptr() == t.data_ptr() + 4 * o
