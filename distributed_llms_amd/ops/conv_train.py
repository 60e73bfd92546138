"""Convolution for TRAINING on the in-repo kernels: forward, data gradient and weight gradient of an NHWC bf16
convolution as one ``torch.autograd.Function``.

* forward  = ``ops.conv.conv2d_nhwc`` (the inference kernels; no bias, no activation);
* data gradient = the same forward kernels with the operands re-indexed: the filter turned by 180 degrees with its
  channel axes swapped, applied with stride 1 to the output gradient - for a strided convolution the gradient is
  first spread onto the input grid (zeros between the samples) inside a buffer that already holds the padding;
* weight gradient = ``csrc/kernels/conv_wgrad.hip`` (split-K MFMA kernel; both operands read from LDS through the
  transposed read, fp32 atomics).

The reference has no training code; there is nothing to cite.  Layers outside the kernels' shape contract (Cin or Cout
not a multiple of 64, kernels larger than 3 x 3, horizontal stride above 2) and CPU tensors take PyTorch's own
convolution, so the Function is safe to use for every layer of a model.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F

from . import _loader
from . import conv as C


def _stride2_tap_table() -> torch.Tensor:
    """Source tap (r * 3 + s, 9 = none) of the 2 x 2 gradient filter entry (parity group g = py * 2 + px, da, db)."""
    idx = []
    for py in range(2):
        for px in range(2):
            for da in range(2):
                for db in range(2):
                    r, sx = py - 2 * da + 1, px - 2 * db + 1
                    idx.append(r * 3 + sx if 0 <= r < 3 and 0 <= sx < 3 else 9)
    return torch.tensor(idx, dtype=torch.long)


_S2_TAPS = _stride2_tap_table()
# vertical stride 2, horizontal stride 1: (row parity py, da, column tap s of the column-flipped filter) -> r * 3 + s
_S21_TAPS = torch.tensor([(py - 2 * da + 1) * 3 + sx if 0 <= py - 2 * da + 1 < 3 else 9
                          for py in range(2) for da in range(2) for sx in range(3)], dtype=torch.long)


def wgrad_serves(cin: int, cout: int, kh: int, kw: int, sh: int, sw: int) -> bool:
    return cin % 64 == 0 and cout % 64 == 0 and 1 <= kh <= 3 and 1 <= kw <= 3 and 1 <= sw <= 2 and 1 <= sh <= 4


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, kh: int, kw: int, stride: Tuple[int, int], padding: Tuple[int, int]
               ) -> torch.Tensor:
    """x: [N,H,W,Cin] bf16, dy: [N,Ho,Wo,Cout] bf16 -> dW [Cout,kh,kw,Cin] fp32."""
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    (sh, sw), (ph, pw) = stride, padding
    if not x.is_cuda or not wgrad_serves(Cin, Cout, kh, kw, sh, sw):
        xf = x.float().permute(0, 3, 1, 2)
        dyf = dy.float().permute(0, 3, 1, 2)
        dw = torch.nn.grad.conv2d_weight(xf, (Cout, Cin, kh, kw), dyf, stride=stride, padding=padding)
        return dw.permute(0, 2, 3, 1).contiguous()
    assert x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and x.is_contiguous() and dy.is_contiguous()
    assert tuple(dy.shape[:3]) == (N, (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1)
    dw = torch.zeros((Cout, kh, kw, Cin), dtype=torch.float32, device=x.device)
    rc = _loader.lib().dlamd_conv_wgrad(_loader.ptr(x), _loader.ptr(dy), _loader.ptr(dw), N, H, W, Cin, Cout, kh, kw,
                                        sh, sw, ph, pw, _loader.stream_ptr())
    _loader.check(rc, "dlamd_conv_wgrad")
    return dw


def conv_dgrad(dy: torch.Tensor, w: torch.Tensor, in_hw: Tuple[int, int], stride: Tuple[int, int],
               padding: Tuple[int, int]) -> torch.Tensor:
    """dy: [N,Ho,Wo,Cout] bf16, w: [Cout,kh,kw,Cin] bf16 -> dX [N,H,W,Cin] bf16 through the forward kernels."""
    N, Ho, Wo, Cout = dy.shape
    _, kh, kw, Cin = w.shape
    (sh, sw), (ph, pw), (H, W) = stride, padding, in_hw
    # filter for the gradient: [Cin, kh, kw, Cout], taps turned by 180 degrees
    wt = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
    if sh == 1 and sw == 1 and kh - 1 - ph >= 0 and kw - 1 - pw >= 0 and H == Ho + kh - 1 - 2 * ph and W == Wo + kw - 1 - 2 * pw:
        return C.conv2d_nhwc(dy, wt, None, stride=1, padding=(kh - 1 - ph, kw - 1 - pw))
    if (sh, sw) == (2, 2) and (kh, kw) == (3, 3) and (ph, pw) == (1, 1) and Cin % 4 == 0:
        # stride 2: the four parities of the input position each see a 2 x 2 window of the output gradient (rows a, a + 1
        # and columns b, b + 1 for input pixel (2 a + py, 2 b + px); tap r = py - 2 da + 1, s = px - 2 db + 1).  Stacked
        # as four channel groups they are ONE 2 x 2 convolution whose pixel-shuffle store writes dX in place: 16 tap
        # blocks (9 of them non-zero) instead of the 36 of the zero-spread form, and no spread buffer.
        w9 = torch.cat([w.reshape(Cout, 9, Cin), w.new_zeros((Cout, 1, Cin))], dim=1)        # tap 9 = the zero tap
        w4 = w9[:, _S2_TAPS.to(w.device)].reshape(Cout, 4, 2, 2, Cin).permute(1, 4, 2, 3, 0).reshape(4 * Cin, 2, 2, Cout)
        w4 = w4.contiguous()
        dyp = F.pad(dy, (0, 0, 0, 1, 0, 1))
        dx = C.conv2d_nhwc(dyp, w4, None, stride=1, padding=0, out_mode=C.OUT_PIXEL_SHUFFLE)
        if dx.shape[1] != H or dx.shape[2] != W:
            dx = dx[:, :H, :W].contiguous()
        return dx
    if (sh, sw) == (2, 1) and (kh, kw) == (3, 3) and (ph, pw) == (1, 1) and H == 2 * Ho:
        # vertical stride 2: the two row parities as two channel groups of one 2 x 3 convolution (rows a, a + 1 of the
        # gradient for input row 2 a + py: tap r = py - 2 da + 1; the three column taps turned round as for stride 1),
        # then one copy that interleaves the rows - 12 tap blocks instead of 18, no spread buffer
        w9 = torch.cat([w.flip(2).reshape(Cout, 9, Cin), w.new_zeros((Cout, 1, Cin))], dim=1)
        w2 = w9[:, _S21_TAPS.to(w.device)].reshape(Cout, 2, 2, 3, Cin).permute(1, 4, 2, 3, 0).reshape(2 * Cin, 2, 3, Cout)
        dyp = F.pad(dy, (0, 0, 0, 0, 0, 1))
        out = C.conv2d_nhwc(dyp, w2.contiguous(), None, stride=1, padding=(0, 1))            # [N, Ho, W, 2 Cin]
        return out.view(N, Ho, W, 2, Cin).permute(0, 1, 3, 2, 4).reshape(N, H, W, Cin)
    # general case: gradient samples placed on the padded input grid, then a VALID stride-1 convolution
    buf = torch.zeros((N, H + kh - 1, W + kw - 1, Cout), dtype=dy.dtype, device=dy.device)
    oy0, ox0 = kh - 1 - ph, kw - 1 - pw
    assert oy0 >= 0 and ox0 >= 0, "padding larger than the kernel is not supported"
    buf[:, oy0 : oy0 + (Ho - 1) * sh + 1 : sh, ox0 : ox0 + (Wo - 1) * sw + 1 : sw] = dy
    return C.conv2d_nhwc(buf, wt, None, stride=1, padding=0)


class _ConvNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        # x: [N,H,W,Cin] bf16; weight: [Cout,Cin,kh,kw] (the nn.Conv2d parameter, any float dtype)
        w = weight.detach().permute(0, 2, 3, 1).to(torch.bfloat16).contiguous()
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.padding, ctx.wdtype = stride, padding, weight.dtype
        return C.conv2d_nhwc(x, w, None, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = conv_dgrad(dy, w, (x.shape[1], x.shape[2]), ctx.stride, ctx.padding)
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad(x, dy, w.shape[1], w.shape[2], ctx.stride, ctx.padding).permute(0, 3, 1, 2).to(ctx.wdtype)
        return dx, dw, None, None


class _ConvFirstLayer(torch.autograd.Function):
    """First convolution of a model: image in (no gradient wanted for it), few input channels.  Forward = the
    inference kernels on the image padded to 8 channels; weight gradient = im2col of the padded image ([N, Ho Wo, kh kw 8]:
    small, 16-byte pixels) times the output gradient on the split-K weight-gradient kernel."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        # x: [N, Cin, H, W] (any float dtype), weight: [Cout, Cin, kh, kw]
        cout, cin, kh, kw = weight.shape
        x8 = F.pad(x.detach().to(torch.bfloat16).permute(0, 2, 3, 1), (0, 8 - cin)).contiguous()     # NHWC, 8 channels
        w8 = F.pad(weight.detach().permute(0, 2, 3, 1), (0, 8 - cin)).to(torch.bfloat16).contiguous()
        ctx.save_for_backward(x8)
        ctx.meta = (stride, padding, (cout, cin, kh, kw), weight.dtype)
        return C.conv2d_nhwc(x8, w8, None, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, dy):
        (x8,) = ctx.saved_tensors
        stride, padding, (cout, cin, kh, kw), wdtype = ctx.meta
        # im2col as kh * kw strided slices of the padded NHWC image (16-byte pixels; F.unfold launches one kernel per
        # sample): cols[n, p, tap * 8 + c]
        (sh, sw), (ph, pw) = stride, padding
        N, Ho, Wo = dy.shape[0], dy.shape[1], dy.shape[2]
        xp = F.pad(x8, (0, 0, pw, pw, ph, ph))
        taps = [xp[:, r : r + sh * (Ho - 1) + 1 : sh, c : c + sw * (Wo - 1) + 1 : sw] for r in range(kh) for c in range(kw)]
        cols = torch.cat(taps, dim=3).reshape(N * Ho * Wo, kh * kw * 8)
        g = dy.reshape(N * Ho * Wo, cout)
        if dy.is_cuda:
            from .linear_train import weight_grad        # the split-K weight-gradient kernel: no library GEMM

            dw = weight_grad(cols, g).t()                                        # [kh kw 8, Cout]
        else:
            dw = cols.float().t() @ g.float()
        dw = dw.reshape(kh, kw, 8, cout)[:, :, :cin].permute(3, 2, 0, 1)         # [Cout, Cin, kh, kw]
        return None, dw.contiguous().to(wdtype), None, None


def conv2d_first_layer_train(x: torch.Tensor, weight: torch.Tensor, stride=(1, 1), padding=(0, 0)) -> torch.Tensor:
    """x: [N, Cin <= 8, H, W] image (NCHW, no gradient) -> [N, Ho, Wo, Cout] bf16 NHWC, differentiable in ``weight``."""
    stride = (stride, stride) if isinstance(stride, int) else tuple(stride)
    padding = (padding, padding) if isinstance(padding, int) else tuple(padding)
    assert x.shape[1] <= 8 and not x.requires_grad
    return _ConvFirstLayer.apply(x, weight, stride, padding)


def conv2d_nhwc_train(x: torch.Tensor, weight: torch.Tensor, stride=(1, 1), padding=(0, 0)) -> torch.Tensor:
    """Differentiable NHWC bf16 convolution.  x: [N,H,W,Cin] bf16 (Cin a multiple of 8), weight: [Cout,Cin,kh,kw]."""
    stride = (stride, stride) if isinstance(stride, int) else tuple(stride)
    padding = (padding, padding) if isinstance(padding, int) else tuple(padding)
    return _ConvNHWC.apply(x.contiguous(), weight, stride, padding)
