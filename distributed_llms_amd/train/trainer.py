"""Recognizer training: CTC on synthetic line crops, data-parallel over RCCL.

The reference has no training code; this subsystem exists so the
random-init recognizer can actually be taught, and to exercise the training-side
MI355X pieces end to end:

* forward/backward through the model's PyTorch path under bf16 autocast,
* ``ops.ctc_loss``   - fused CTC forward + gradient HIP kernel,
* ``parallel.ddp``   - flat buckets, all-reduce (or reduce-scatter + all-gather)
  over RCCL/xGMI launched from autograd hooks so it overlaps backward,
* ``ops.optim``      - one fused AdamW launch per bucket, clipping folded in,
* checkpoints        - safetensors weights + optimizer moments + step/RNG, atomic
  rename, written by rank 0 (rank-local files for sharded optimizer state).
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import asdict, dataclass
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from ..ops import ctc_loss as ctc_ops
from ..ops.optim import FlatAdamW
from ..parallel.comm import DistEnv
from ..parallel.ddp import BucketedDDP
from .data import SyntheticLines


@dataclass
class TrainConfig:
    batch_size: int = 64          # per rank
    width: int = 512
    height: int = 48
    steps: int = 100
    lr: float = 1e-3
    warmup_steps: int = 20
    weight_decay: float = 0.01
    max_grad_norm: Optional[float] = 5.0
    bucket_mb: float = 64.0
    dp_mode: str = "allreduce"    # allreduce | rs_ag (sharded optimizer step between the halves)
    amp: bool = True
    seed: int = 0
    max_chars: int = 40
    vocab: int = 0                # > 0: draw lines from a fixed set of strings (overfit tests)
    font_px: int = 34
    ckpt_dir: Optional[str] = None
    ckpt_every: int = 0
    log_every: int = 10
    own_convs: bool = True        # convolutions (forward, dgrad, wgrad) on the in-repo kernels (ops/conv_train.py); GPU only


class CTCTrainer:
    def __init__(self, model: torch.nn.Module, cfg: TrainConfig, device: torch.device | str = "cuda:0",
                 env: Optional[DistEnv] = None):
        self.cfg = cfg
        self.device = torch.device(device)
        self.env = env or DistEnv(device=self.device)
        torch.manual_seed(cfg.seed)
        self.model = model.to(self.device).float()
        self.model.train()
        if cfg.own_convs and hasattr(self.model, "own_conv_train"):
            self.model.own_conv_train = self.device.type == "cuda"
        self.ddp = BucketedDDP(self.model, cfg.bucket_mb, cfg.dp_mode)
        flats = [(b.flat_param, b.flat_grad) for b in self.ddp.flat.buckets]
        # unsharded: update exactly the occupied prefix (the padding depends on the world size; keeping it out
        # of the norm reduction makes DP=N on identical shards bitwise equal to DP=1)
        shards = [(0, b.used) for b in self.ddp.flat.buckets]
        if cfg.dp_mode == "rs_ag" and self.ddp.world > 1:
            shards = [self.ddp.shard_range(b) for b in self.ddp.flat.buckets]
        self.opt = FlatAdamW(flats, lr=cfg.lr, weight_decay=cfg.weight_decay, max_grad_norm=cfg.max_grad_norm,
                             shards=shards)
        self.data = SyntheticLines(model.charset, cfg.height, cfg.width, cfg.max_chars, cfg.seed, cfg.font_px, cfg.vocab)
        self.step = 0
        self.history: List[Dict[str, float]] = []
        self.comm_time = 0.0

    # ------------------------------------------------------------------ data
    def synthetic_batch(self, step: int):
        x, il, tg, tl, texts = self.data.batch(step, self.cfg.batch_size, self.env.rank, self.env.world)
        if self.device.type == "cuda":
            x = x.pin_memory().to(self.device, non_blocking=True)
        return x, il, tg, tl

    # ------------------------------------------------------------------ one optimisation step
    def _lr_at(self, step: int) -> float:
        w = self.cfg.warmup_steps
        return self.cfg.lr * min(1.0, (step + 1) / w) if w > 0 else self.cfg.lr

    def train_step(self, x: torch.Tensor, in_lens: torch.Tensor, targets: torch.Tensor, tgt_lens: torch.Tensor) -> float:
        cfg = self.cfg
        self.ddp.zero_grad()
        use_amp = cfg.amp and self.device.type == "cuda"
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_amp):
            logits = self.model(x, in_lens)
        logits = logits.float().contiguous()
        nll = ctc_ops.ctc_loss(logits, targets, in_lens, tgt_lens, blank=0, zero_infinity=True)
        denom = tgt_lens.to(nll.device).clamp(min=1).float()
        loss = (nll / denom).mean()
        loss.backward()                      # bucket all-reduces start from the autograd hooks
        t0 = time.perf_counter()
        self.ddp.finish_backward()
        self.comm_time += time.perf_counter() - t0
        self.opt.lr = self._lr_at(self.step)
        if cfg.dp_mode == "rs_ag" and self.ddp.world > 1:
            self.opt.step(all_reduce_norm=lambda t: dist.all_reduce(t))
            self.ddp.all_gather_params()
        else:
            self.opt.step()
        self.step += 1
        return float(loss.detach().item())

    def fit(self, steps: Optional[int] = None, verbose: bool = False) -> List[Dict[str, float]]:
        steps = steps or self.cfg.steps
        t0 = time.perf_counter()
        for _ in range(steps):
            batch = self.synthetic_batch(self.step)
            loss = self.train_step(*batch)
            rec = {"step": self.step, "loss": loss, "time": time.perf_counter() - t0}
            self.history.append(rec)
            if verbose and self.env.is_main and self.step % self.cfg.log_every == 0:
                print(f"[train] step {self.step} loss {loss:.4f} ({rec['time']:.1f}s)", flush=True)
            if self.cfg.ckpt_dir and self.cfg.ckpt_every and self.step % self.cfg.ckpt_every == 0:
                self.save_checkpoint(self.cfg.ckpt_dir)
                self._checkpoint_barrier()
        return self.history

    # ------------------------------------------------------------------ evaluation through the inference path
    @torch.no_grad()
    def evaluate(self, n_batches: int = 2, step0: int = 10_000_000) -> Dict[str, float]:
        """Greedy-decode fresh synthetic lines with the *training path* logits; exact-match + char accuracy."""
        from ..ops.ctc import ctc_greedy_reference

        self.model.eval()
        n_ok = n_tot = ch_ok = ch_tot = 0
        for k in range(n_batches):
            x, il, tg, tl, texts = self.data.batch(step0 + k, self.cfg.batch_size, self.env.rank, self.env.world)
            logits = self.model(x.to(self.device), il).float().cpu()
            ids, ln, _ = ctc_greedy_reference(logits, il.to(torch.int32), blank=0)
            preds = self.model.decode_strings(ids, ln)
            for p, t in zip(preds, texts):
                t2 = "".join(c for c in t if c in self.data.idx)
                n_ok += int(p == t2)
                n_tot += 1
                ch_tot += max(len(t2), 1)
                ch_ok += max(0, len(t2) - _edit_distance(p, t2))
        self.model.train()
        return {"exact": n_ok / max(n_tot, 1), "char_acc": ch_ok / max(ch_tot, 1)}

    # ------------------------------------------------------------------ checkpoint / resume
    def save_checkpoint(self, directory: str) -> Optional[Path]:
        from safetensors.torch import save_file

        d = Path(directory)
        d.mkdir(parents=True, exist_ok=True)
        sharded = self.cfg.dp_mode == "rs_ag" and self.ddp.world > 1
        if sharded or self.env.is_main:
            osd = self.opt.state_dict()
            tensors = {f"m.{k}": t.contiguous() for k, t in enumerate(osd["m"])}
            tensors.update({f"v.{k}": t.contiguous() for k, t in enumerate(osd["v"])})
            name = f"optim_rank{self.env.rank}.safetensors" if sharded else "optim.safetensors"
            _atomic(d / name, lambda p: save_file(tensors, str(p)))
        if not self.env.is_main:
            return None
        sd = {k: v.detach().cpu().contiguous().clone() for k, v in self.model.state_dict().items()}
        _atomic(d / "model.safetensors", lambda p: save_file(sd, str(p)))
        meta = {"step": self.step, "opt_step": self.opt.step_count, "config": asdict(self.cfg),
                "world": self.env.world, "torch_rng": torch.random.get_rng_state().tolist(),
                "charset": self.model.charset}
        _atomic(d / "meta.json", lambda p: p.write_text(json.dumps(meta)))
        return d

    def _checkpoint_barrier(self) -> None:
        """Every rank waits until rank 0 has written the checkpoint: without it the other ranks enter the next step's
        all-reduce up to a whole file write ahead of rank 0, which a bounded-spin collective reads as a dead peer."""
        import torch.distributed as dist

        if self.env.world > 1 and dist.is_initialized():
            dist.barrier()

    def load_checkpoint(self, directory: str) -> int:
        from safetensors.torch import load_file

        d = Path(directory)
        meta = json.loads((d / "meta.json").read_text())
        sd = load_file(str(d / "model.safetensors"))
        # parameters are views into the flat buckets: copy in place so the views stay intact
        own = self.model.state_dict()
        for k, v in sd.items():
            own[k].copy_(v)
        sharded = self.cfg.dp_mode == "rs_ag" and self.ddp.world > 1
        name = f"optim_rank{self.env.rank}.safetensors" if sharded else "optim.safetensors"
        if (d / name).exists():
            ot = load_file(str(d / name))
            n = len(self.opt.m)
            self.opt.load_state_dict({"step": meta["opt_step"], "m": [ot[f"m.{k}"] for k in range(n)],
                                      "v": [ot[f"v.{k}"] for k in range(n)]})
        self.step = int(meta["step"])
        torch.random.set_rng_state(torch.tensor(meta["torch_rng"], dtype=torch.uint8))
        return self.step


def _atomic(path: Path, writer) -> None:
    tmp = path.with_name(path.name + f".tmp{os.getpid()}")
    writer(tmp)
    os.replace(tmp, path)


def _edit_distance(a: str, b: str) -> int:
    if a == b:
        return 0
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]
