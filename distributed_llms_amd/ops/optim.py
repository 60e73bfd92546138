"""Fused AdamW over flat parameter buckets.

One ``dlamd_adamw`` launch updates a whole flat bucket (parallel/ddp.py) - or, in
``rs_ag`` data-parallel mode, just this rank's 1/N shard of it - instead of one
small launch per tensor.  Gradient clipping is folded in: the global-norm scale
is a device scalar read by the kernel, so clipping costs one ``dlamd_sumsq``
reduction and no extra pass over the gradients.  CPU tensors use the same
maths in PyTorch (the oracle in tests).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import _loader


def adamw_reference_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
                     b1: float, b2: float, eps: float, wd: float, gscale: float = 1.0) -> None:
    g = g * gscale
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.mul_(1 - lr * wd).addcdiv_(m, denom, value=-lr / bc1)


def adamw_step_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
                b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8, wd: float = 0.01,
                gscale: Optional[torch.Tensor] = None, shadow: Optional[torch.Tensor] = None) -> None:
    """In-place AdamW on flat fp32 tensors.  ``gscale``: optional 1-element device tensor multiplied into g."""
    if not p.is_cuda:
        adamw_reference_(p, g, m, v, step, lr, b1, b2, eps, wd, 1.0 if gscale is None else float(gscale))
        if shadow is not None:
            shadow.copy_(p)
        return
    assert p.dtype == g.dtype == m.dtype == v.dtype == torch.float32
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    n = p.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n
    if shadow is not None:
        assert shadow.dtype == torch.bfloat16 and shadow.numel() == n and shadow.is_contiguous()
    rc = _loader.lib().dlamd_adamw(
        _loader.ptr(p), _loader.ptr(g), _loader.ptr(m), _loader.ptr(v), _loader.ptr(shadow), n,
        lr, b1, b2, eps, wd, 1 - b1 ** step, 1 - b2 ** step, _loader.ptr(gscale), _loader.stream_ptr(),
    )
    _loader.check(rc, "dlamd_adamw")


def sumsq_into_(out: torch.Tensor, g: torch.Tensor) -> None:
    """out[0] += sum(g^2)."""
    if not g.is_cuda:
        out += g.float().pow(2).sum()
        return
    assert g.dtype == torch.float32 and g.is_contiguous() and out.dtype == torch.float32
    rc = _loader.lib().dlamd_sumsq(_loader.ptr(g), g.numel(), _loader.ptr(out), _loader.stream_ptr())
    _loader.check(rc, "dlamd_sumsq")


class FlatAdamW:
    """AdamW over a list of (flat_param, flat_grad) pairs, optionally restricted to a [lo, hi) shard of each."""

    def __init__(self, flats: Sequence[Tuple[torch.Tensor, torch.Tensor]], lr: float = 1e-3,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01,
                 max_grad_norm: Optional[float] = None, shards: Optional[Sequence[Tuple[int, int]]] = None):
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        self.flats = list(flats)
        self.shards = list(shards) if shards is not None else [(0, p.numel()) for p, _ in self.flats]
        self.m: List[torch.Tensor] = []
        self.v: List[torch.Tensor] = []
        for (p, _), (lo, hi) in zip(self.flats, self.shards):
            self.m.append(torch.zeros(hi - lo, dtype=torch.float32, device=p.device))
            self.v.append(torch.zeros(hi - lo, dtype=torch.float32, device=p.device))
        dev = self.flats[0][0].device
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._scale = torch.ones(1, dtype=torch.float32, device=dev)
        self.last_grad_norm: Optional[torch.Tensor] = None

    def step(self, all_reduce_norm=None) -> None:
        """``all_reduce_norm``: callable summing the squared-norm tensor across ranks (sharded mode)."""
        self.step_count += 1
        gscale = None
        if self.max_grad_norm is not None:
            self._norm.zero_()
            for (p, g), (lo, hi) in zip(self.flats, self.shards):
                sumsq_into_(self._norm, g[lo:hi])
            if all_reduce_norm is not None:
                all_reduce_norm(self._norm)
            norm = self._norm.sqrt()
            self.last_grad_norm = norm
            torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0, out=self._scale)
            gscale = self._scale
        for k, ((p, g), (lo, hi)) in enumerate(zip(self.flats, self.shards)):
            adamw_step_(p[lo:hi], g[lo:hi], self.m[k], self.v[k], self.step_count, self.lr, self.betas[0],
                        self.betas[1], self.eps, self.wd, gscale)

    def state_dict(self) -> dict:
        return {"step": self.step_count, "lr": self.lr, "m": [t.detach().cpu() for t in self.m],
                "v": [t.detach().cpu() for t in self.v], "shards": list(self.shards)}

    def load_state_dict(self, sd: dict) -> None:
        self.step_count = int(sd["step"])
        self.lr = float(sd.get("lr", self.lr))
        for k in range(len(self.m)):
            self.m[k].copy_(sd["m"][k])
            self.v[k].copy_(sd["v"][k])
