#!/usr/bin/env python3
"""Recognizer training benchmark (the reference workload's fine-tuning configuration): CRNN + CTC on synthetic line crops, DP over RCCL.

    python bench/train_bench.py --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 bench/train_bench.py

Reports samples/sec (whole job), ms/step, time blocked on gradient collectives, and the final loss.
Host-side line rendering is excluded: batches are pre-rendered before the timed region."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch

from distributed_llms_amd.models.crnn import CRNN
from distributed_llms_amd.parallel import comm
from distributed_llms_amd.train.trainer import CTCTrainer, TrainConfig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="launcher contract: number of ranks (informational)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--mode", default="allreduce", choices=["allreduce", "rs_ag", "direct"])
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    ap.add_argument("--library-convs", action="store_true", help="convolutions through the library (MIOpen) instead of the in-repo training kernels (ops/conv_train.py)")
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--distinct", type=int, default=4, help="pre-rendered batches cycled through")
    args = ap.parse_args()
    env = comm.init_distributed(force_cpu=args.cpu)
    cfg = TrainConfig(batch_size=args.batch, width=args.width, dp_mode=args.mode, bucket_mb=args.bucket_mb,
                      max_chars=max(8, args.width // 22), own_convs=not args.library_convs)
    torch.manual_seed(0)
    model = CRNN() if env.device.type == "cuda" else CRNN(hidden=64, widths=(16, 32, 32, 32, 64, 64))
    tr = CTCTrainer(model, cfg, env.device, env)
    batches = [tr.synthetic_batch(k) for k in range(args.distinct)]
    loss = float("nan")
    for k in range(args.warmup):
        loss = tr.train_step(*batches[k % len(batches)])
    if env.device.type == "cuda":
        torch.cuda.synchronize()
    comm.barrier()
    tr.comm_time = 0.0
    t0 = time.perf_counter()
    for k in range(args.steps):
        loss = tr.train_step(*batches[k % len(batches)])
    if env.device.type == "cuda":
        torch.cuda.synchronize()
    comm.barrier()
    dt = comm.all_reduce_max_scalar(time.perf_counter() - t0, env.device)
    if env.is_main:
        n_params = sum(p.numel() for p in model.parameters())
        print(json.dumps({
            "metric": "recognizer training samples/sec (CTC, synthetic line crops)",
            "value": round(args.steps * args.batch * env.world / dt, 2), "unit": "samples/s",
            "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "data": "synthetic rendered line crops, pre-rendered outside the timed region; random-init weights",
            "allreduce_wait_ms_per_step": round(1e3 * tr.comm_time / args.steps, 3),
            "grad_bytes_per_step": n_params * 4, "buckets": len(tr.ddp.flat.buckets), "dp_mode": args.mode,
            "final_loss": round(loss, 4), "dtype": "bf16 autocast, fp32 master" if env.device.type == "cuda" else "fp32",
            "config": {"model": "CRNN", "convolutions": "library (MIOpen)" if args.library_convs or env.device.type != "cuda" else "in-repo kernels (ops/conv_train.py)", "params": n_params, "per_gpu_batch": args.batch, "width": args.width,
                       "global_batch": args.batch * env.world, "seq_len": args.width // 4,
                       "parallelism": f"dp{env.world}"},
        }), flush=True)
    comm.shutdown()


if __name__ == "__main__":
    main()
