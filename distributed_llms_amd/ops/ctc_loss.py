"""CTC loss with a fused forward + gradient HIP kernel.

``ctc_loss(logits, targets, input_lens, target_lens)`` takes *raw logits*
``[N, T, V]`` (log-softmax is fused into the kernel) and returns per-sample
negative log-likelihoods.  On device tensors the gradient w.r.t. the logits is
produced in the same launch pair and replayed in ``backward``; on CPU tensors
the op falls back to ``torch.nn.functional.ctc_loss`` (also the test oracle).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F

from . import _loader


def ctc_loss_reference(logits: torch.Tensor, targets: torch.Tensor, input_lens: torch.Tensor,
                       target_lens: torch.Tensor, blank: int = 0, zero_infinity: bool = True) -> torch.Tensor:
    """Per-sample NLL via PyTorch (fp32).  logits: [N, T, V]; targets: concatenated labels."""
    lp = F.log_softmax(logits.float(), dim=-1).transpose(0, 1)  # [T, N, V]
    return F.ctc_loss(lp, targets.long().cpu(), input_lens.long().cpu(), target_lens.long().cpu(), blank=blank,
                      reduction="none", zero_infinity=zero_infinity)


def _ctc_gpu(logits: torch.Tensor, targets: torch.Tensor, input_lens: torch.Tensor, target_lens: torch.Tensor,
             blank: int, zero_infinity: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    assert logits.is_cuda and logits.dim() == 3 and logits.is_contiguous()
    assert logits.dtype in (torch.float32, torch.bfloat16)
    N, T, V = logits.shape
    dev = logits.device
    tl_host = target_lens.to("cpu", torch.int64)
    assert int(tl_host.sum()) == targets.numel(), "targets must be the concatenation of all label sequences"
    Smax = 2 * int(tl_host.max().item() if N > 0 else 0) + 1
    tgt = targets.to(dev, torch.int32).contiguous()
    tlen = tl_host.to(torch.int32)
    toff = (torch.cumsum(tlen, 0) - tlen).to(torch.int32)
    tlen_d, toff_d = tlen.to(dev), toff.to(dev)
    ilen_d = input_lens.to(dev, torch.int32).contiguous()
    lse = torch.empty((N * T,), dtype=torch.float32, device=dev)
    alpha = torch.empty((N * T * Smax,), dtype=torch.float32, device=dev)
    loss = torch.empty((N,), dtype=torch.float32, device=dev)
    grad = torch.empty((N, T, V), dtype=torch.float32, device=dev)
    rc = _loader.lib().dlamd_ctc_loss(
        _loader.ptr(logits), 1 if logits.dtype == torch.float32 else 0, _loader.ptr(tgt), _loader.ptr(toff_d),
        _loader.ptr(tlen_d), _loader.ptr(ilen_d), _loader.ptr(lse), _loader.ptr(alpha), _loader.ptr(loss),
        _loader.ptr(grad), N, T, V, Smax, blank, 1 if zero_infinity else 0, _loader.stream_ptr(),
    )
    _loader.check(rc, "dlamd_ctc_loss")
    return loss, grad


class _CTCLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, input_lens, target_lens, blank, zero_infinity):
        loss, grad = _ctc_gpu(logits.detach().contiguous(), targets, input_lens, target_lens, blank, zero_infinity)
        ctx.save_for_backward(grad)
        ctx.in_dtype = logits.dtype
        return loss

    @staticmethod
    def backward(ctx, go):
        (grad,) = ctx.saved_tensors
        g = grad * go.float().view(-1, 1, 1)
        return g.to(ctx.in_dtype), None, None, None, None, None


def ctc_loss(logits: torch.Tensor, targets: torch.Tensor, input_lens: torch.Tensor, target_lens: torch.Tensor,
             blank: int = 0, zero_infinity: bool = True) -> torch.Tensor:
    """Per-sample CTC negative log-likelihood, differentiable w.r.t. ``logits`` ([N, T, V] raw scores)."""
    if not logits.is_cuda:
        return ctc_loss_reference(logits, targets, input_lens, target_lens, blank, zero_infinity)
    return _CTCLossFn.apply(logits, targets, input_lens, target_lens, blank, zero_infinity)
