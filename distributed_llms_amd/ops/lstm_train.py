"""Differentiable bidirectional LSTM layer on the persistent HIP kernels (csrc/kernels/lstm_train.hip).

Training-side twin of ``ops/lstm.py``: the input projection and its gradients run on the in-repo GEMM and
weight-gradient kernels (``ops/linear_train.py``); the two recurrences - forward over the steps, backward over them
in reverse - run in the hand-written persistent kernels, one launch each instead of one GEMM + one cell kernel per
time step (the eager path needs ~10 000 launches per CRNN training step); the W_hh gradient is one GEMM per
direction over all steps.

Semantics are those of ``nn.LSTM(bidirectional=True, batch_first=True)`` fed a packed sequence and padded
back: steps at ``t >= len`` output zeros and receive no gradient; the reverse direction starts at ``len-1``.
The reference has no training code; nothing to cite there.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import _loader
from .lstm import lstm_gate_permutation

_PERM = {}


def _perm(H: int, device) -> torch.Tensor:
    key = (H, str(device))
    if key not in _PERM:
        _PERM[key] = lstm_gate_permutation(H).to(device)
    return _PERM[key]


def _frag(w: torch.Tensor) -> torch.Tensor:
    """[ndir, R, K] -> MFMA A-operand fragment order [ndir, R/16, K/32, 64, 8] bf16 (lane = 16*(k/8 % 4) + row)."""
    nd, R, K = w.shape
    return (w.to(torch.bfloat16).view(nd, R // 16, 16, K // 32, 4, 8).permute(0, 1, 3, 4, 2, 5)
            .reshape(nd, R // 16, K // 32, 64, 8).contiguous())


class _BiLSTMRecurrence(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xproj: torch.Tensor, w_hh: torch.Tensor, lens: Optional[torch.Tensor]):
        # xproj [N, T, ndir*4H] bf16 (permuted gate columns); w_hh [ndir, 4H, H] (permuted rows)
        assert xproj.is_cuda and xproj.dtype == torch.bfloat16 and xproj.is_contiguous()
        N, T, C = xproj.shape
        ndir, G, H = w_hh.shape
        assert C == ndir * G and G == 4 * H and H in (64, 128, 256), (xproj.shape, w_hh.shape)
        if lens is not None:
            assert lens.is_cuda and lens.dtype == torch.int32 and lens.numel() == N
        dev = xproj.device
        wfrag = _frag(w_hh.detach())
        y = torch.empty((N, T, ndir * H), dtype=torch.bfloat16, device=dev)
        gact = torch.empty((N, T, ndir * G), dtype=torch.bfloat16, device=dev)
        cbuf = torch.empty((N, T, ndir * H), dtype=torch.float32, device=dev)
        rc = _loader.lib().dlamd_lstm_train_fwd(_loader.ptr(xproj), _loader.ptr(wfrag), _loader.ptr(lens), _loader.ptr(y),
                                                _loader.ptr(gact), _loader.ptr(cbuf), N, T, H, ndir, _loader.stream_ptr())
        _loader.check(rc, "dlamd_lstm_train_fwd")
        ctx.save_for_backward(y, gact, cbuf, w_hh, lens if lens is not None else torch.empty(0, device=dev))
        ctx.has_lens = lens is not None
        return y

    @staticmethod
    def backward(ctx, dy: torch.Tensor):
        y, gact, cbuf, w_hh, lens = ctx.saved_tensors
        lens = lens if ctx.has_lens else None
        N, T, _ = y.shape
        ndir, G, H = w_hh.shape
        wtfrag = _frag(w_hh.detach().transpose(1, 2))          # rows = hidden units, K = gate columns
        dg = torch.zeros_like(gact)
        dyf = dy.float().contiguous()
        rc = _loader.lib().dlamd_lstm_train_bwd(_loader.ptr(dyf), _loader.ptr(gact), _loader.ptr(cbuf), _loader.ptr(wtfrag),
                                                _loader.ptr(lens), _loader.ptr(dg), N, T, H, ndir, _loader.stream_ptr())
        _loader.check(rc, "dlamd_lstm_train_bwd")
        dw = None
        if ctx.needs_input_grad[1]:
            # dW_hh[d] = sum over (n, t) of dG_t^T h_prev, h_prev = output of the previously processed step
            z = torch.zeros((N, 1, H), dtype=y.dtype, device=y.device)
            hp = [torch.cat([z, y[:, :-1, :H]], 1)]
            if ndir == 2:
                hp.append(torch.cat([y[:, 1:, H:], z], 1))      # y is zero at t >= len, so t = len-1 sees h_prev = 0
            from .linear_train import weight_grad        # the split-K weight-gradient kernel: no library GEMM

            dw = torch.stack([weight_grad(hp[d].reshape(N * T, H), dg[:, :, d * G:(d + 1) * G].reshape(N * T, G))
                              for d in range(ndir)], 0).to(w_hh.dtype)
        return dg, dw, None


def bilstm_train(x: torch.Tensor, m: torch.nn.LSTM, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [N, T, I] -> [N, T, 2H] bf16 through ``m``'s parameters (one layer, bidirectional, batch_first)."""
    assert m.num_layers == 1 and m.bidirectional and m.batch_first and m.bias
    H = m.hidden_size
    perm = _perm(H, x.device)
    w_ih = torch.cat([m.weight_ih_l0[perm], m.weight_ih_l0_reverse[perm]], 0)
    b = torch.cat([(m.bias_ih_l0 + m.bias_hh_l0)[perm], (m.bias_ih_l0_reverse + m.bias_hh_l0_reverse)[perm]], 0)
    w_hh = torch.stack([m.weight_hh_l0[perm], m.weight_hh_l0_reverse[perm]], 0)
    # the input projection and both of its gradients on the in-repo GEMM / weight-gradient kernels (ops/linear_train.py)
    from .linear_train import linear_train

    xproj = linear_train(x, w_ih, b).contiguous()
    if lens is not None:
        lens = lens.to(x.device, torch.int32)
    return _BiLSTMRecurrence.apply(xproj, w_hh, lens)
