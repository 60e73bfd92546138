"""Batch normalisation (+ ReLU) for training on NHWC bf16 activations (``csrc/kernels/bn_train.hip``).

Two memory-bound passes each way (per-channel reduction, elementwise apply); the activation's mask is recomputed from
the input in the backward pass, so only the input, the batch mean and the inverse standard deviation are kept.
The reference has no training code; nothing to cite.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import _loader
from .conv import ACT


MAX_WGS = 512          # rows of the partial-sum scratch (bn_train.hip::BN_MAX_WGS)


def bn_serves(c: int) -> bool:
    return 8 <= c <= 512 and c % 8 == 0 and 256 % (c // 8) == 0


class _BNActNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, run_mean, run_var, eps, momentum, act):
        # z: [..., C] bf16 contiguous; gamma / beta fp32 [C]; running statistics updated in place
        C = z.shape[-1]
        P = z.numel() // C
        y = torch.empty_like(z)
        sums = torch.empty((2, C), dtype=torch.float32, device=z.device)
        partial = torch.empty((MAX_WGS, 2, C), dtype=torch.float32, device=z.device)
        mean = torch.empty((C,), dtype=torch.float32, device=z.device)
        invstd = torch.empty((C,), dtype=torch.float32, device=z.device)
        rc = _loader.lib().dlamd_bn_train_fwd(_loader.ptr(z), _loader.ptr(gamma), _loader.ptr(beta), _loader.ptr(y),
                                              _loader.ptr(sums), _loader.ptr(partial), _loader.ptr(mean), _loader.ptr(invstd), _loader.ptr(run_mean),
                                              _loader.ptr(run_var), P, C, float(eps), float(momentum), ACT[act],
                                              _loader.stream_ptr())
        _loader.check(rc, "dlamd_bn_train_fwd")
        ctx.save_for_backward(z, gamma, beta, mean, invstd)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        z, gamma, beta, mean, invstd = ctx.saved_tensors
        C = z.shape[-1]
        P = z.numel() // C
        dy = dy.contiguous()
        dz = torch.empty_like(z)
        sums = torch.empty((2, C), dtype=torch.float32, device=z.device)
        partial = torch.empty((MAX_WGS, 2, C), dtype=torch.float32, device=z.device)
        rc = _loader.lib().dlamd_bn_train_bwd(_loader.ptr(dy), _loader.ptr(z), _loader.ptr(mean), _loader.ptr(invstd),
                                              _loader.ptr(gamma), _loader.ptr(beta), _loader.ptr(sums), _loader.ptr(partial), _loader.ptr(dz), P, C,
                                              ACT[ctx.act], _loader.stream_ptr())
        _loader.check(rc, "dlamd_bn_train_bwd")
        return dz, sums[1].to(gamma.dtype), sums[0].to(beta.dtype), None, None, None, None, None


def bn_act_train(z: torch.Tensor, bn: torch.nn.BatchNorm2d, act: str = "relu") -> torch.Tensor:
    """``act(batch_norm(z))`` with batch statistics for a channels-last activation z: [N, H, W, C] bf16.  Updates the
    module's running statistics like ``nn.BatchNorm2d`` in training mode."""
    C = z.shape[-1]
    if (not z.is_cuda or z.dtype != torch.bfloat16 or not bn_serves(C) or act not in ("none", "relu") or not bn.training
            or bn.weight is None or bn.momentum is None or not bn.track_running_stats):
        y = bn(z.permute(0, 3, 1, 2))
        y = F.relu(y) if act == "relu" else y
        assert act in ("none", "relu")
        return y.permute(0, 2, 3, 1).contiguous()
    assert bn.weight.dtype == torch.float32 and bn.running_mean.dtype == torch.float32
    y = _BNActNHWC.apply(z.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, act)
    bn.num_batches_tracked += 1
    return y
