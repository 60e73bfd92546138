"""Direct all-reduce (parallel/direct.py) timing, per message size and algorithm.

Two ways to run:
* under torchrun on a multi-GPU node (one rank per GPU, RCCL group): also times ``dist.all_reduce`` (RCCL)
  on the same buffers, which is the comparison SURVEY 5.8 asks for;
* plain ``python bench/allreduce_bench.py --ranks 4`` on a one-GPU box: the ranks are processes that share
  cuda:0 (gloo carries the IPC handles).  That measures the barrier / launch latency floor and an HBM-bound
  bandwidth, not xGMI; RCCL cannot be compared there (it refuses two ranks on one device).
Prints one JSON line per (dtype, bytes, algo) from rank 0.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--max-mb", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=128)
    args = ap.parse_args()
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
