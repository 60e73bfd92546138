"""Cross-entropy (teacher-forcing) trainer for the transformer recognizer (models/trocr.py), data-parallel.

The reference has no training code; its transformer recognizer would be a downloaded checkpoint
(services/ocr_service/download_models_hf.py:39-41).  None is reachable offline, so this trainer is what gives the
encoder-decoder non-random weights here, and what makes the large-gradient side of data-parallel training real: the
base-sized model has 334 M parameters, i.e. a 1.3 GB fp32 gradient per step through ``parallel.ddp.BucketedDDP``.

* input: synthetic line crops (``train.data.SyntheticLines``) resized to the encoder's S x S, as the engine does at
  inference (runtime/engine.py::_submit_rec_trocr);
* targets: byte-level token ids from the native vocabulary (csrc/runtime/bpe_tokenizer.cpp): bos, bytes, eos, pad;
* forward: the model's PyTorch path (``encode_reference`` + ``decode_logits_reference``, causal self-attention and
  cross-attention over all encoder tokens) under bf16 autocast; activations of the encoder layers are recomputed in
  backward (``torch.utils.checkpoint``) when ``recompute`` is set, so the base model fits large batches;
* loss: token cross-entropy ignoring pad; optimizer: fused flat AdamW over the DDP buckets.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from ..models.trocr import TrOCR
from ..ops.optim import FlatAdamW
from ..parallel.comm import DistEnv
from ..parallel.ddp import BucketedDDP
from ..runtime.tokenizer import ByteLevelVocab
from .data import SyntheticLines
from .trainer import _edit_distance

_PRINTABLE = "".join(chr(c) for c in range(32, 127))


@dataclass
class Seq2SeqConfig:
    batch_size: int = 32          # per rank
    width: int = 384              # rendered line width before the resize to S x S
    steps: int = 100
    lr: float = 3e-4
    warmup_steps: int = 20
    decay_steps: int = 0          # > 0: cosine decay to 5 % of lr over this many steps
    weight_decay: float = 0.01
    max_grad_norm: Optional[float] = 1.0
    bucket_mb: float = 64.0
    dp_mode: str = "allreduce"
    amp: bool = True
    seed: int = 0
    max_chars: int = 24
    vocab: int = 0                # > 0: draw lines from a fixed set of strings (overfit tests)
    font_px: int = 30
    recompute: bool = False
    log_every: int = 10


class Seq2SeqTrainer:
    def __init__(self, model: TrOCR, cfg: Seq2SeqConfig, device: torch.device | str = "cuda:0",
                 env: Optional[DistEnv] = None):
        self.cfg = cfg
        self.device = torch.device(device)
        self.env = env or DistEnv(device=self.device)
        torch.manual_seed(cfg.seed)
        self.model = model.to(self.device).float()
        self.model.train()
        self.vocab = ByteLevelVocab()
        assert model.cfg.vocab >= self.vocab.size, "model vocabulary is smaller than the byte-level vocabulary"
        assert (model.cfg.bos, model.cfg.pad, model.cfg.eos) == (self.vocab.bos, self.vocab.pad, self.vocab.eos)
        self.ddp = BucketedDDP(self.model, cfg.bucket_mb, cfg.dp_mode)
        flats = [(b.flat_param, b.flat_grad) for b in self.ddp.flat.buckets]
        shards = [(0, b.used) for b in self.ddp.flat.buckets]
        if cfg.dp_mode == "rs_ag" and self.ddp.world > 1:
            shards = [self.ddp.shard_range(b) for b in self.ddp.flat.buckets]
        self.opt = FlatAdamW(flats, lr=cfg.lr, weight_decay=cfg.weight_decay, max_grad_norm=cfg.max_grad_norm, shards=shards)
        self.data = SyntheticLines(_PRINTABLE, 48, cfg.width, cfg.max_chars, cfg.seed, cfg.font_px, cfg.vocab)
        self.step = 0
        self.comm_time = 0.0
        self.history: List[Dict[str, float]] = []

    # ------------------------------------------------------------------ data
    def batch(self, step: int) -> Tuple[torch.Tensor, torch.Tensor, List[str]]:
        """-> (images [N,3,S,S] in [-1,1], tokens [N, L] = bos, bytes, eos, pad..., texts)."""
        x, _, _, _, texts = self.data.batch(step, self.cfg.batch_size, self.env.rank, self.env.world)
        S = self.model.cfg.image_size
        x = F.interpolate(x, size=(S, S), mode="bilinear", align_corners=False)
        ids = [self.vocab.encode(t, add_special=True)[: self.model.cfg.max_len] for t in texts]
        L = max(len(i) for i in ids)
        tok = np.full((len(ids), L), self.vocab.pad, dtype=np.int64)
        for k, i in enumerate(ids):
            tok[k, : len(i)] = i
        return x.to(self.device), torch.from_numpy(tok).to(self.device), texts

    # ------------------------------------------------------------------ forward / loss
    def _encode(self, x: torch.Tensor) -> torch.Tensor:
        if not self.cfg.recompute:
            return self.model.encode_reference(x)
        from torch.utils.checkpoint import checkpoint

        return checkpoint(self.model.encode_reference, x, use_reentrant=False)

    def loss(self, x: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
        use_amp = self.cfg.amp and self.device.type == "cuda"
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_amp):
            mem = self._encode(x)
            logits = self.model.decode_logits_reference(mem, tokens[:, :-1])
        return F.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), tokens[:, 1:].reshape(-1),
                               ignore_index=self.vocab.pad)

    def train_step(self, x: torch.Tensor, tokens: torch.Tensor) -> float:
        cfg = self.cfg
        self.ddp.zero_grad()
        loss = self.loss(x, tokens)
        loss.backward()
        t0 = time.perf_counter()
        self.ddp.finish_backward()
        self.comm_time += time.perf_counter() - t0
        w = cfg.warmup_steps
        lr = cfg.lr * min(1.0, (self.step + 1) / w) if w > 0 else cfg.lr
        if cfg.decay_steps > 0:
            lr *= 0.05 + 0.95 * 0.5 * (1.0 + np.cos(np.pi * min(1.0, self.step / cfg.decay_steps)))
        self.opt.lr = float(lr)
        if cfg.dp_mode == "rs_ag" and self.ddp.world > 1:
            self.opt.step(all_reduce_norm=lambda t: dist.all_reduce(t))
            self.ddp.all_gather_params()
        else:
            self.opt.step()
        self.step += 1
        return float(loss.detach().item())

    def fit(self, steps: Optional[int] = None, verbose: bool = False) -> List[Dict[str, float]]:
        t0 = time.perf_counter()
        for _ in range(steps or self.cfg.steps):
            x, tok, _ = self.batch(self.step)
            loss = self.train_step(x, tok)
            self.history.append({"step": self.step, "loss": loss, "time": time.perf_counter() - t0})
            if verbose and self.env.is_main and self.step % self.cfg.log_every == 0:
                print(f"[trocr-train] step {self.step} loss {loss:.4f}", flush=True)
        return self.history

    # ------------------------------------------------------------------ evaluation: greedy decoding
    @torch.no_grad()
    def evaluate(self, n_batches: int = 1, step0: int = 10_000_000, through_kernels: bool = False) -> Dict[str, float]:
        """Greedy decode fresh batches (with a fixed vocabulary: the same strings in new renderings).  With
        ``through_kernels`` the inference path of the model (HIP kernels, KV cache, graph-captured steps) decodes
        instead of the PyTorch reference."""
        self.model.eval()
        n_ok = n_tot = ed = nch = 0
        max_len = min(self.model.cfg.max_len, self.cfg.max_chars + 4)
        for k in range(n_batches):
            x, _, texts = self.batch(step0 + k)
            if through_kernels:
                xi = torch.zeros((x.shape[0], x.shape[2], x.shape[3], 8), dtype=torch.bfloat16, device=x.device)
                xi[..., :3] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
                toks, _ = self.model.generate(xi, max_len=max_len)
            else:
                toks = self.model.generate_reference(x, max_len)
            preds = self.model.tokens_to_text(toks, self.vocab)
            for p, t in zip(preds, texts):
                n_ok += int(p == t)
                n_tot += 1
                ed += _edit_distance(p, t)
                nch += max(len(t), 1)
        self.model.train()
        if through_kernels:
            self.model._pk = None          # packed inference weights are stale after the next optimizer step
        return {"exact": n_ok / max(n_tot, 1), "cer": ed / max(nch, 1)}
