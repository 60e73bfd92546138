"""Linear layer for TRAINING on the in-repo kernels: forward, data gradient and weight gradient as one
``torch.autograd.Function``.

* forward ``y = x w^T + b``      -> ``ops.conv.linear_bf16`` (the persistent MFMA GEMM for large layers);
* data gradient ``dx = dy w``    -> the same kernel with the transposed weight as its "weight" operand;
* weight gradient ``dw = dy^T x`` -> ``weight_grad``: ``csrc/kernels/conv_wgrad.hip`` as a 1 x 1 convolution (its
  reduction runs over the pixels, i.e. over every row of the batch, with both operands as they lie in memory);
* bias gradient = a column sum.

Layers the kernels do not take (feature counts that are not multiples of 64 after padding, CPU tensors) fall back to
PyTorch.  The reference has no training code; there is nothing to cite.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import conv as C
from .conv_train import conv_wgrad


def _pad_to(t: torch.Tensor, dim: int, mult: int) -> torch.Tensor:
    n = t.shape[dim]
    p = (-n) % mult
    if p == 0:
        return t
    pad = [0, 0] * (t.dim() - 1 - dim) + [0, p]
    return F.pad(t, pad)


def weight_grad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """dW [N, K] fp32 = dy^T x for x [M, K], dy [M, N] (bf16, any strides) on the split-K weight-gradient kernel: a
    1 x 1 "convolution" over one image row of M pixels (the kernel's work units are runs of 64 consecutive pixels of a
    row); both channel counts are padded to multiples of 64 with zeros.

    Measured against the alternative of transposing both operands and running the GEMM kernels (whose reduction index
    must be the contiguous one): the two transposes of a 32 768 x 2048 gradient cost more than the GEMM saves - 16.8 k
    samples/s against 17.4 k for the recogniser's training step (the library GEMM it replaces: 20.3 k)."""
    M, K = x.shape
    N = dy.shape[1]
    xp = _pad_to(x.to(torch.bfloat16), 1, 64).contiguous()
    dyp = _pad_to(dy.to(torch.bfloat16), 1, 64).contiguous()
    dwp = conv_wgrad(xp.view(1, 1, M, xp.shape[1]), dyp.view(1, 1, M, dyp.shape[1]), 1, 1, (1, 1), (0, 0))
    return dwp.view(dyp.shape[1], xp.shape[1])[:N, :K]


class _LinearBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        # x: [M, K] bf16 contiguous; weight: [N, K] (the nn.Linear parameter, any float dtype); bias: [N] or None
        w = weight.detach().to(torch.bfloat16).contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        ctx.save_for_backward(x, w)
        ctx.meta = (weight.dtype, None if bias is None else bias.dtype)
        N = w.shape[0]
        if N % 8:
            # the kernel stores 4 output columns at a time (a 258-class classifier): zero rows fill the last group
            y = C.linear_bf16(x, _pad_to(w, 0, 8), None if b is None else _pad_to(b, 0, 8))
            return y[:, :N].contiguous()
        return C.linear_bf16(x, w, b)

    @staticmethod
    def backward(ctx, dy: torch.Tensor):
        x, w = ctx.saved_tensors
        wdtype, bdtype = ctx.meta
        dy = dy.to(torch.bfloat16).contiguous()
        M, K = x.shape
        N = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # the GEMM's reduction (here over N) runs in 8-element vectors: zero columns fill a ragged N
            dx = C.linear_bf16(_pad_to(dy, 1, 8), _pad_to(w.t(), 1, 8).contiguous())
        if ctx.needs_input_grad[1]:
            dw = weight_grad(x, dy).to(wdtype)
        if bdtype is not None and ctx.needs_input_grad[2]:
            db = dy.sum(0, dtype=torch.float32).to(bdtype)
        return dx, dw, db


def linear_train(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable ``x @ weight^T + bias`` in bf16.  x: [..., K]; returns [..., N] bf16."""
    lead = x.shape[:-1]
    K = x.shape[-1]
    if not x.is_cuda or K % 8 != 0:
        return F.linear(x.to(weight.dtype), weight, bias).to(torch.bfloat16 if x.is_cuda else x.dtype)
    y = _LinearBF16.apply(x.reshape(-1, K).to(torch.bfloat16).contiguous(), weight, bias)
    return y.reshape(*lead, weight.shape[0])
