"""Building blocks shared by the detector and the recognizers.

Every block has two execution paths over ONE set of parameters:

* ``forward(x)``  - NCHW fp32/bf16 PyTorch path with autograd (training,
  CPU proxy baseline).
* ``infer(x)``    - NHWC bf16 inference path: BatchNorm folded into the conv,
  weights pre-packed once as ``[Cout, kh, kw, Cin]`` bf16, executed by the MFMA
  implicit-GEMM kernel with fused bias / residual / activation epilogue
  (``ops.conv.conv2d_nhwc``; on CPU tensors the same call evaluates the fp32
  oracle, so ``infer`` is testable without a GPU).
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops import conv as C

IntPair = Union[int, Tuple[int, int]]


def _pair(v: IntPair) -> Tuple[int, int]:
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class ConvBNAct(nn.Module):
    """Conv2d (+BatchNorm) (+activation).  ``cin_pad``/``cout_pad`` describe the
    packed (kernel-side) channel counts: Cin must be a multiple of 8 and Cout of 4."""

    def __init__(self, cin: int, cout: int, k: IntPair = 3, stride: IntPair = 1, padding: Optional[IntPair] = None,
                 act: str = "relu", bn: bool = True, bias: bool = False):
        super().__init__()
        kh, kw = _pair(k)
        if padding is None:
            padding = (kh // 2, kw // 2)
        self.cin, self.cout = cin, cout
        self.k, self.stride, self.padding = (kh, kw), _pair(stride), _pair(padding)
        self.act = act
        self.conv = nn.Conv2d(cin, cout, (kh, kw), self.stride, self.padding, bias=bias or not bn)
        self.bn = nn.BatchNorm2d(cout) if bn else None
        nn.init.kaiming_normal_(self.conv.weight, mode="fan_out", nonlinearity="relu")
        if self.conv.bias is not None:
            nn.init.zeros_(self.conv.bias)
        self._packed = None

    # ---- training / proxy path ------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.conv(x)
        if self.bn is not None:
            y = self.bn(y)
        return C._act_ref(y, self.act)

    # ---- inference path -----------------------------------------------------------
    def fold(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """BN-folded (weight [Cout,kh,kw,Cin_pad] bf16, bias [Cout_pad] f32)."""
        w = self.conv.weight.detach().float()
        b = self.conv.bias.detach().float() if self.conv.bias is not None else torch.zeros(self.cout, device=w.device)
        if self.bn is not None:
            inv = torch.rsqrt(self.bn.running_var.float() + self.bn.eps) * self.bn.weight.detach().float()
            w = w * inv[:, None, None, None]
            b = (b - self.bn.running_mean.float()) * inv + self.bn.bias.detach().float()
        cin_p, cout_p = _round_up(self.cin, 8), _round_up(self.cout, 4)
        wp = torch.zeros(cout_p, self.k[0], self.k[1], cin_p, device=w.device)
        wp[: self.cout, :, :, : self.cin] = w.permute(0, 2, 3, 1)
        bp = torch.zeros(cout_p, device=w.device)
        bp[: self.cout] = b
        return wp.to(torch.bfloat16).contiguous(), bp.contiguous()

    def prepare(self) -> None:
        self._packed = self.fold()

    def train(self, mode: bool = True):
        if mode:
            self._packed = None
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def infer(self, x: torch.Tensor, **kw) -> torch.Tensor:
        if self._packed is None:
            self.prepare()
        w, b = self._packed
        return C.conv2d_nhwc(x, w, b, stride=self.stride, padding=self.padding, act=kw.pop("act", self.act), **kw)


class Deconv2x2(nn.Module):
    """ConvTranspose2d(k=2, s=2) (+BN) (+act) = GEMM + pixel-shuffle store."""

    def __init__(self, cin: int, cout: int, act: str = "relu", bn: bool = True):
        super().__init__()
        self.cin, self.cout, self.act = cin, cout, act
        self.deconv = nn.ConvTranspose2d(cin, cout, 2, 2, bias=not bn)
        self.bn = nn.BatchNorm2d(cout) if bn else None
        nn.init.kaiming_normal_(self.deconv.weight, mode="fan_out", nonlinearity="relu")
        if self.deconv.bias is not None:
            nn.init.zeros_(self.deconv.bias)
        self._packed = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.deconv(x)
        if self.bn is not None:
            y = self.bn(y)
        return C._act_ref(y, self.act)

    def fold(self) -> Tuple[torch.Tensor, torch.Tensor]:
        w = self.deconv.weight.detach().float()  # [Cin, Cout, 2, 2]
        b = self.deconv.bias.detach().float() if self.deconv.bias is not None else torch.zeros(self.cout, device=w.device)
        if self.bn is not None:
            inv = torch.rsqrt(self.bn.running_var.float() + self.bn.eps) * self.bn.weight.detach().float()
            w = w * inv[None, :, None, None]
            b = (b - self.bn.running_mean.float()) * inv + self.bn.bias.detach().float()
        # GEMM rows ordered (dy, dx, c')
        wg = w.permute(2, 3, 1, 0).reshape(4 * self.cout, 1, 1, self.cin)
        return wg.to(torch.bfloat16).contiguous(), b.repeat(4).contiguous()

    def prepare(self) -> None:
        self._packed = self.fold()

    def train(self, mode: bool = True):
        if mode:
            self._packed = None
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def infer(self, x: torch.Tensor, prob_thr: Optional[float] = None):
        if self._packed is None:
            self.prepare()
        w, b = self._packed
        if self.cout == 1:
            return C.conv2d_nhwc(x, w, b, act=self.act, out_mode=C.OUT_PROB, thr=0.3 if prob_thr is None else prob_thr)
        return C.conv2d_nhwc(x, w, b, act=self.act, out_mode=C.OUT_PIXEL_SHUFFLE)


class BasicBlock(nn.Module):
    """ResNet basic block; the residual add + ReLU is fused into the second conv's epilogue."""

    def __init__(self, cin: int, cout: int, stride: int = 1):
        super().__init__()
        self.c1 = ConvBNAct(cin, cout, 3, stride, act="relu")
        self.c2 = ConvBNAct(cout, cout, 3, 1, act="none")
        self.short = None
        if stride != 1 or cin != cout:
            self.short = ConvBNAct(cin, cout, 1, stride, padding=0, act="none")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        idt = x if self.short is None else self.short(x)
        return F.relu(self.c2(self.c1(x)) + idt)

    def infer(self, x: torch.Tensor) -> torch.Tensor:
        idt = x if self.short is None else self.short.infer(x)
        y = self.c1.infer(x)
        return self.c2.infer(y, act="relu", res=idt, res_mode=C.RES_SAME)


def prepare_all(module: nn.Module) -> None:
    """Fold + pack every conv of a model once (call after ``.to(device).eval()``)."""
    for m in module.modules():
        if isinstance(m, (ConvBNAct, Deconv2x2)):
            m.prepare()
