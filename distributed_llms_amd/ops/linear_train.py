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
    """dW [N, K] fp32 = dy^T x for x [M, K], dy [M, N] (bf16, any strides) on the split-K weight-gradient kernel.

    That kernel gives a workgroup dW[64 output channels][up to 9 taps][64 input channels] and walks over runs of 64
    pixels of an image row; per MFMA K-step a wave reads 4 dY fragments plus ONE input fragment per tap.  A plain 1 x 1
    "convolution" over one row of M pixels would use one tap: 5 LDS fragment reads per 4 MFMAs and dy fetched once per
    64 input channels.  So the K input features are laid out as kh x kw TAPS of K / (kh kw) channels instead:

        X'[r][kw p + s][c] = x[p][((r kw + s) C' + c)]      (an image of kh rows and kw M pixels, C' = K / (kh kw))

    convolved with a kh x kw filter at stride (1, kw) and no padding has exactly one output row of M pixels, and its
    weight gradient dW'[co][r][s][c] is dW[co][(r kw + s) C' + c] - up to six taps per workgroup (kh <= 3, kw <= 2 is
    what the kernel's patch buffer holds at stride 2): 10 fragment reads per 24 MFMAs and a sixth of the dy traffic.
    Channel counts are padded to multiples of 64 with zeros.

    Measured against transposing both operands and running the GEMM kernels (whose reduction index must be the
    contiguous one): the two transposes of a 32 768 x 2048 gradient cost more than the GEMM saves."""
    M, K = x.shape
    N = dy.shape[1]
    xp = _pad_to(x.to(torch.bfloat16), 1, 64)
    dyp = _pad_to(dy.to(torch.bfloat16), 1, 64).contiguous()
    blocks = xp.shape[1] // 64
    kh, kw = next(((a, b) for a, b in ((3, 2), (2, 2), (3, 1), (2, 1)) if blocks % (a * b) == 0), (1, 1))
    cp = xp.shape[1] // (kh * kw)
    ximg = xp.reshape(M, kh, kw, cp).permute(1, 0, 2, 3).reshape(1, kh, M * kw, cp).contiguous()
    dwp = conv_wgrad(ximg, dyp.view(1, 1, M, dyp.shape[1]), kh, kw, (1, kw), (0, 0))        # [N64, kh, kw, C']
    return dwp.reshape(dyp.shape[1], xp.shape[1])[:N, :K]


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
