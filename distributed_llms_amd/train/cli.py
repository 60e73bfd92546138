"""Train the CRNN recognizer on synthetic lines (single GPU or DP under torch.distributed.run).

    python -m distributed_llms_amd.train.cli --steps 2000 --out weights/
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m distributed_llms_amd.train.cli ...

Writes ``<out>/crnn.safetensors`` (inference weights) and ``<out>/ckpt/`` (resumable: model, AdamW moments,
step, RNG).  ``--resume`` continues from ``<out>/ckpt``."""
from __future__ import annotations

import argparse
import json
import os

import torch

from ..models.crnn import CRNN
from ..parallel import comm
from ..utils.weights import save_model_weights
from .trainer import CTCTrainer, TrainConfig


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--mode", default="allreduce", choices=["allreduce", "rs_ag", "direct"])
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    ap.add_argument("--out", default="weights")
    ap.add_argument("--ckpt-every", type=int, default=500)
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--vocab", type=int, default=0)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="small model (CPU smoke runs)")
    args = ap.parse_args(argv)
    env = comm.init_distributed(force_cpu=args.cpu)
    cfg = TrainConfig(batch_size=args.batch, width=args.width, steps=args.steps, lr=args.lr, dp_mode=args.mode,
                      bucket_mb=args.bucket_mb, ckpt_dir=os.path.join(args.out, "ckpt"), ckpt_every=args.ckpt_every,
                      vocab=args.vocab, max_chars=max(6, args.width // 22))
    model = CRNN(hidden=64, widths=(16, 32, 32, 32, 64, 64)) if args.tiny else CRNN()
    tr = CTCTrainer(model, cfg, env.device, env)
    if args.resume and os.path.exists(os.path.join(cfg.ckpt_dir, "meta.json")):
        print(f"resumed at step {tr.load_checkpoint(cfg.ckpt_dir)}")
    remaining = max(0, args.steps - tr.step)
    tr.fit(remaining, verbose=True)
    tr.save_checkpoint(cfg.ckpt_dir)
    acc = tr.evaluate(2)
    if env.is_main:
        os.makedirs(args.out, exist_ok=True)
        save_model_weights(tr.model, os.path.join(args.out, "crnn.safetensors"))
        print(json.dumps({"steps": tr.step, "final_loss": tr.history[-1]["loss"] if tr.history else None, **acc}))
    comm.shutdown()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
