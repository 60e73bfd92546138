"""Direct all-reduce (parallel/direct.py) timing, per message size and algorithm.

Three ways to run:
* ``--steps K --warmup W [--gpus N] [--mb M]`` under torchrun (one rank per GPU): the launcher contract - W untimed
  and exactly K timed all-reduces of one M-MB bf16 gradient bucket, bracketed by barrier + synchronise, MAX over
  ranks, ONE JSON line from rank 0 (bus bandwidth of the faster of the direct kernel and RCCL; both are listed).
  ``--cpu`` times gloo on host tensors (contract check on machines without GPUs);
* ``--sweep`` under torchrun on a multi-GPU node (one rank per GPU, RCCL group): also times ``dist.all_reduce`` (RCCL)
  on the same buffers;
* plain ``python bench/allreduce_bench.py --ranks 4`` on a one-GPU box: the ranks are processes that share
  cuda:0 (gloo carries the IPC handles).  That measures the barrier / launch latency floor and an HBM-bound
  bandwidth, not xGMI; RCCL cannot be compared there (it refuses two ranks on one device).
The last two print one JSON line per (dtype, bytes, algo) from rank 0.
"""
import argparse
import json
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _time(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dist.barrier()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters   # us


def run(rank, world, shared_gpu, iters, max_mb, blocks):
    from distributed_llms_amd.parallel.direct import DirectAllReduce

    ar = DirectAllReduce(max_bytes=max_mb << 20, blocks=blocks)
    sizes = [4 << 10, 64 << 10, 512 << 10, 2 << 20, 16 << 20, 64 << 20]
    for nbytes in [s for s in sizes if s <= max_mb << 20]:
        x = torch.randn(nbytes // 2, device="cuda").to(torch.bfloat16)
        row = {"bytes": nbytes, "dtype": "bf16", "ranks": world, "shared_gpu": shared_gpu, "blocks": blocks}
        for algo in ("one_shot", "two_shot"):
            if algo == "one_shot" and nbytes > 2 << 20:
                continue
            us = _time(lambda: ar.all_reduce_(x, scale=1.0 / world, algo=algo), iters)
            row[algo + "_us"] = round(us, 1)
            row[algo + "_algbw_GBps"] = round(nbytes / us / 1e3, 1)
        # the same bucket as a registered region, reduced in place (what BucketedDDP does): no staging copy
        reg = ar.alloc(nbytes // 2, torch.bfloat16)
        ar.register([reg])
        reg.copy_(x)
        us = _time(lambda: ar.all_reduce_region_(reg, scale=1.0 / world), iters)
        row["in_place_us"] = round(us, 1)
        row["in_place_algbw_GBps"] = round(nbytes / us / 1e3, 1)
        if not shared_gpu:
            us = _time(lambda: dist.all_reduce(x), iters)
            row["rccl_us"] = round(us, 1)
            row["rccl_algbw_GBps"] = round(nbytes / us / 1e3, 1)
        st = ar.status()
        row["status"] = st
        if rank == 0:
            print(json.dumps(row), flush=True)
    ar.close()


def _spawned(rank, world, port, iters, max_mb, blocks):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    run(rank, world, True, iters, max_mb, blocks)
    dist.destroy_process_group()


def contract(args):
    """One BENCH-shaped line: all-reduce of one gradient bucket."""
    import time

    world = int(os.environ.get("WORLD_SIZE", "1"))
    nbytes = int(args.mb * (1 << 20))
    if args.cpu or not torch.cuda.is_available():
        if world > 1:
            dist.init_process_group("gloo")
        rank = dist.get_rank() if world > 1 else 0
        x = torch.randn(nbytes // 4)
        algos = {"gloo": (lambda: dist.all_reduce(x)) if world > 1 else (lambda: x.mul_(1.0))}
        sync = lambda: None
        dtype, dev = "fp32-cpu-debug", torch.device("cpu")
        ar = None
    else:
        from distributed_llms_amd.parallel.comm import init_distributed
        from distributed_llms_amd.parallel.direct import DirectAllReduce

        env = init_distributed()
        rank, dev = env.rank, env.device
        x = torch.randn(nbytes // 2, device=dev).to(torch.bfloat16)
        ar = DirectAllReduce(max_bytes=nbytes, blocks=args.blocks) if world > 1 else None
        algos = {"rccl": (lambda: dist.all_reduce(x)) if world > 1 else (lambda: x.mul_(1.0))}
        if ar is not None:
            algos["direct_two_shot"] = lambda: ar.all_reduce_(x, scale=1.0 / world, algo="two_shot")
            reg = ar.alloc(nbytes // 2, torch.bfloat16)
            ar.register([reg])
            reg.copy_(x)
            algos["direct_in_place"] = lambda: ar.all_reduce_region_(reg, scale=1.0 / world)
        sync = torch.cuda.synchronize
        dtype = "bf16"
    res = {}
    for name, fn in algos.items():
        for _ in range(args.warmup):
            fn()
        sync()
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        sync()
        if world > 1:
            dist.barrier()
        sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if world > 1:
            t = t.to(dev) if dev.type == "cuda" else t
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res[name] = float(t.item()) / args.steps
    if rank == 0:
        best = min(res, key=res.get)
        bus = lambda sec: nbytes * 2 * (world - 1) / max(world, 1) / sec / 1e9 if world > 1 else nbytes / sec / 1e9
        print(json.dumps({
            "metric": "gradient all-reduce bus bandwidth, one %g MB bucket" % args.mb, "value": round(bus(res[best]), 2),
            "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * res[best], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic gradient bucket (random values)",
            "config": {"model": "one flat gradient bucket", "global_batch": world, "seq_len": nbytes,
                       "parallelism": f"dp{world}", "algorithm": best,
                       "ms_per_algorithm": {k: round(1e3 * v, 4) for k, v in res.items()},
                       "busbw_GBps_per_algorithm": {k: round(bus(v), 2) for k, v in res.items()}},
        }), flush=True)
    if ar is not None:
        ar.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--max-mb", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=128)
    ap.add_argument("--gpus", type=int, default=0, help="launcher contract: number of ranks (informational)")
    ap.add_argument("--steps", type=int, default=0, help="launcher contract: timed all-reduces (0: sweep modes)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mb", type=float, default=64.0, help="bucket size of the contract mode")
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--sweep", action="store_true")
    args = ap.parse_args()
    if args.steps > 0 and not args.sweep:
        return contract(args)
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from distributed_llms_amd.parallel.comm import init_distributed

        env = init_distributed()
        run(env.rank, env.world, False, args.iters, args.max_mb, args.blocks)
        dist.destroy_process_group()
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_spawned, args=(args.ranks, port, args.iters, args.max_mb, args.blocks), nprocs=args.ranks, join=True)


if __name__ == "__main__":
    main()
