"""Page raster -> tensor and line-crop ops.

``resample_regions`` turns a list of rectangular regions of u8 source images
(whole pages for detection, text-line boxes for recognition) into one padded
NHWC bf16 batch ``[N, Hd, Wd, 8]`` (3 real channels + 5 zero channels so every
pixel is one 16-byte vector for the first MFMA convolution).

GPU path: ``dlamd_resample`` (csrc/kernels/resample.hip).  CPU path / oracle:
``resample_regions_reference`` builds the same fractional-box / bilinear
weights as dense matrices in fp32.

In the reference this work is PyMuPDF ``get_pixmap`` + OpenCV resize/normalise
inside magic-pdf (services/ocr_service/ocr_service.py:215).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _loader

ITEM_DTYPE = np.dtype(
    [
        ("src", "<u8"), ("Hs", "<i4"), ("Ws", "<i4"), ("C", "<i4"), ("rot", "<i4"),
        ("x0", "<f4"), ("y0", "<f4"), ("x1", "<f4"), ("y1", "<f4"),
        ("out_h", "<i4"), ("out_w", "<i4"), ("pitch", "<i4"), ("angle", "<f4"),
    ],
    align=False,
)
assert ITEM_DTYPE.itemsize == 56

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
HALF_MEAN = (0.5, 0.5, 0.5)
HALF_STD = (0.5, 0.5, 0.5)


@dataclass
class Region:
    """One resample job.  ``image`` is an HWC (or HW) uint8 tensor."""

    image: torch.Tensor
    box: Tuple[float, float, float, float]  # x0, y0, x1, y1 in source pixels
    out_h: int
    out_w: int
    rot: int = 0
    angle: float = 0.0      # radians: the region is ``box`` turned by this angle about its centre (slanted text line)


def _axis_weights(lo0: float, scale: float, n_out: int, n_src: int, reverse: bool = False) -> torch.Tensor:
    """Dense [n_out, n_src] resampling matrix identical to the kernel's maths."""
    Wm = torch.zeros(n_out, n_src, dtype=torch.float64)
    for o in range(n_out):
        u = (n_out - 1 - o) if reverse else o
        if scale > 1.0:
            lo = np.float32(lo0) + np.float32(u) * np.float32(scale)
            hi = lo + np.float32(scale)
            a = max(0, int(math.floor(lo)))
            b = min(n_src, int(math.ceil(hi)))
            if b <= a:
                a = min(max(a, 0), n_src - 1)
                b = a + 1
            ws = []
            for s in range(a, b):
                w = min(float(hi), s + 1.0) - max(float(lo), float(s))
                ws.append(max(w, 1e-6))
            tot = sum(ws)
            for s, w in zip(range(a, b), ws):
                Wm[o, s] = w / tot
        else:
            sx = float(np.float32(lo0) + (np.float32(u) + np.float32(0.5)) * np.float32(scale) - np.float32(0.5))
            sx = min(max(sx, 0.0), n_src - 1.0)
            a = int(math.floor(sx))
            b = min(a + 1, n_src - 1)
            fx = sx - a
            Wm[o, a] += 1.0 - fx
            Wm[o, b] += fx
    return Wm


def resample_regions_reference(
    regions: Sequence[Region], Hd: int, Wd: int, mean=IMAGENET_MEAN, std=IMAGENET_STD
) -> torch.Tensor:
    out = torch.zeros(len(regions), Hd, Wd, 8, dtype=torch.float32)
    for n, r in enumerate(regions):
        img = r.image
        if img.dim() == 2:
            img = img[:, :, None]
        img = img.cpu().double()
        Hs, Ws, C = img.shape
        x0, y0, x1, y1 = r.box
        rw, rh = x1 - x0, y1 - y0
        if r.angle != 0.0:
            res = _affine_crop_reference(img, r)
        elif r.rot == 0:
            su, sv = rw / r.out_w, rh / r.out_h
            # the kernel uses the box filter on both axes as soon as either axis shrinks
            if su > 1.0 or sv > 1.0:
                Wx = _axis_weights_box(x0, su, r.out_w, Ws)
                Wy = _axis_weights_box(y0, sv, r.out_h, Hs)
            else:
                Wx = _axis_weights(x0, su, r.out_w, Ws)
                Wy = _axis_weights(y0, sv, r.out_h, Hs)
            res = torch.einsum("oy,yxc,px->opc", Wy, img, Wx)  # [out_h, out_w, C]
        else:
            su, sv = rw / r.out_h, rh / r.out_w
            if su > 1.0 or sv > 1.0:
                Wu = _axis_weights_box(x0, su, r.out_h, Ws)
                Wv = _axis_weights_box(y0, sv, r.out_w, Hs, reverse=True)
            else:
                Wu = _axis_weights(x0, su, r.out_h, Ws)
                Wv = _axis_weights(y0, sv, r.out_w, Hs, reverse=True)
            # out[oy, ox] = sum_{y,x} Wv[ox, y] * Wu[oy, x] * img[y, x]
            res = torch.einsum("py,yxc,ox->opc", Wv, img, Wu)
        if C == 1:
            res = res.expand(-1, -1, 3)
        m = torch.tensor(mean, dtype=torch.float64)
        s = torch.tensor(std, dtype=torch.float64)
        res = (res / 255.0 - m) / s
        out[n, : r.out_h, : r.out_w, :3] = res.float()
    return out


def _affine_crop_reference(img: torch.Tensor, r: Region) -> torch.Tensor:
    """The kernel's affine path: up to 4 x 4 bilinear taps per output pixel, positions laid out in the rectangle's
    frame and turned by ``angle`` about its centre.  img: [Hs, Ws, C] float64 -> [out_h, out_w, C]."""
    Hs, Ws, C = img.shape
    x0, y0, x1, y1 = (np.float32(v) for v in r.box)
    rw, rh = x1 - x0, y1 - y0
    if r.rot == 0:
        su, sv = rw / np.float32(r.out_w), rh / np.float32(r.out_h)
        u = np.arange(r.out_w, dtype=np.float32)[None, :].repeat(r.out_h, 0)
        v = np.arange(r.out_h, dtype=np.float32)[:, None].repeat(r.out_w, 1)
    else:
        su, sv = rw / np.float32(r.out_h), rh / np.float32(r.out_w)
        u = np.arange(r.out_h, dtype=np.float32)[:, None].repeat(r.out_w, 1)
        v = (r.out_w - 1 - np.arange(r.out_w, dtype=np.float32))[None, :].repeat(r.out_h, 0)
    ca, sa = np.float32(math.cos(r.angle)), np.float32(math.sin(r.angle))
    cx, cy = np.float32(0.5) * (x0 + x1), np.float32(0.5) * (y0 + y1)
    nx, ny = min(4, max(1, int(math.ceil(su)))), min(4, max(1, int(math.ceil(sv))))
    a = img.numpy()
    acc = np.zeros((r.out_h, r.out_w, C), dtype=np.float64)
    for j in range(ny):
        ly = (v + np.float32((j + 0.5) / ny)) * sv - np.float32(0.5) * rh
        for i in range(nx):
            lx = (u + np.float32((i + 0.5) / nx)) * su - np.float32(0.5) * rw
            px = np.clip(cx + lx * ca - ly * sa - np.float32(0.5), 0, Ws - 1)
            py = np.clip(cy + lx * sa + ly * ca - np.float32(0.5), 0, Hs - 1)
            xa, ya = np.floor(px).astype(np.int64), np.floor(py).astype(np.int64)
            xb, yb = np.minimum(xa + 1, Ws - 1), np.minimum(ya + 1, Hs - 1)
            fx, fy = (px - xa)[..., None].astype(np.float64), (py - ya)[..., None].astype(np.float64)
            acc += (a[ya, xa] * (1 - fx) + a[ya, xb] * fx) * (1 - fy) + (a[yb, xa] * (1 - fx) + a[yb, xb] * fx) * fy
    return torch.from_numpy(acc / (nx * ny))


def _axis_weights_box(lo0: float, scale: float, n_out: int, n_src: int, reverse: bool = False) -> torch.Tensor:
    """Box-filter weights regardless of scale (kernel uses box on both axes if either shrinks)."""
    Wm = torch.zeros(n_out, n_src, dtype=torch.float64)
    for o in range(n_out):
        u = (n_out - 1 - o) if reverse else o
        lo = np.float32(lo0) + np.float32(u) * np.float32(scale)
        hi = lo + np.float32(scale)
        a = max(0, int(math.floor(lo)))
        b = min(n_src, int(math.ceil(hi)))
        if b <= a:
            a = min(max(a, 0), n_src - 1)
            b = a + 1
        ws = [max(min(float(hi), s + 1.0) - max(float(lo), float(s)), 1e-6) for s in range(a, b)]
        tot = sum(ws)
        for s, w in zip(range(a, b), ws):
            Wm[o, s] = w / tot
    return Wm


def pack_items(regions: Sequence[Region]) -> np.ndarray:
    items = np.zeros(len(regions), dtype=ITEM_DTYPE)
    for n, r in enumerate(regions):
        img = r.image
        assert img.dtype == torch.uint8
        Hs, Ws = img.shape[0], img.shape[1]
        C = 1 if img.dim() == 2 else img.shape[2]
        assert C in (1, 3)
        # rows may be pitched (e.g. a [:H, :W] view of a JPEG plane padded to whole MCUs); pixels must be packed
        assert img.stride(1) == C and (img.dim() == 2 or img.stride(2) == 1), "pixels must be densely packed in a row"
        items[n] = (img.data_ptr(), Hs, Ws, C, r.rot, r.box[0], r.box[1], r.box[2], r.box[3], r.out_h, r.out_w,
                    img.stride(0), r.angle)
    return items


def resample_items_gpu(
    items_dev: torch.Tensor, n: int, Hd: int, Wd: int, mean=IMAGENET_MEAN, std=IMAGENET_STD,
    out: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """Launch the kernel on an already-uploaded item table (uint8 tensor of n*56 bytes)."""
    if out is None:
        out = torch.empty((n, Hd, Wd, 8), dtype=torch.bfloat16, device=items_dev.device)
    rc = _loader.lib().dlamd_resample(
        _loader.ptr(items_dev), n, _loader.ptr(out), Hd, Wd,
        float(mean[0]), float(mean[1]), float(mean[2]),
        1.0 / std[0], 1.0 / std[1], 1.0 / std[2], _loader.stream_ptr(),
    )
    _loader.check(rc, "dlamd_resample")
    return out


def stem_serves(w: torch.Tensor, stride, padding) -> bool:
    """Whether ``resample_stem_items_gpu`` can stand in for resample + this first convolution."""
    return (w.dim() == 4 and tuple(w.shape[1:]) == (3, 3, 8) and w.shape[0] in (32, 64) and tuple(stride) == (2, 2)
            and tuple(padding) == (1, 1))


def resample_stem_items_gpu(
    items_dev: torch.Tensor, n: int, Hd: int, Wd: int, mean, std, w: torch.Tensor, bias: Optional[torch.Tensor],
    act: str = "relu",
) -> torch.Tensor:
    """Resample fused with the first convolution (3 x 3, stride 2, pad 1; ``w``: [Cout, 3, 3, 8] bf16 with BN folded,
    Cout 32 or 64): -> [n, (Hd - 1) // 2 + 1, (Wd - 1) // 2 + 1, Cout] bf16 without ever writing the [n, Hd, Wd, 8]
    tensor.  Same arithmetic as ``resample_items_gpu`` followed by ``conv2d_nhwc`` (the resampled value is rounded to
    bf16 before it is multiplied)."""
    from .conv import ACT

    assert w.dtype == torch.bfloat16 and w.is_contiguous() and stem_serves(w, (2, 2), (1, 1))
    cout = w.shape[0]
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() >= cout and bias.is_contiguous())
    out = torch.empty((n, (Hd - 1) // 2 + 1, (Wd - 1) // 2 + 1, cout), dtype=torch.bfloat16, device=items_dev.device)
    rc = _loader.lib().dlamd_resample_stem(
        _loader.ptr(items_dev), n, Hd, Wd, float(mean[0]), float(mean[1]), float(mean[2]),
        1.0 / std[0], 1.0 / std[1], 1.0 / std[2], _loader.ptr(w), _loader.ptr(bias), _loader.ptr(out), cout, ACT[act],
        _loader.stream_ptr())
    _loader.check(rc, "dlamd_resample_stem")
    return out


def resample_regions(
    regions: Sequence[Region], Hd: int, Wd: int, mean=IMAGENET_MEAN, std=IMAGENET_STD,
    out: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """[N, Hd, Wd, 8] bf16 batch from u8 regions (GPU kernel or CPU oracle)."""
    if len(regions) == 0:
        dev = out.device if out is not None else "cpu"
        return torch.zeros((0, Hd, Wd, 8), dtype=torch.bfloat16, device=dev)
    dev = regions[0].image.device
    if dev.type != "cuda":
        res = resample_regions_reference(regions, Hd, Wd, mean, std).to(torch.bfloat16)
        if out is not None:
            out.copy_(res)
            return out
        return res
    items = pack_items(regions)
    items_dev = torch.from_numpy(items.view(np.uint8).copy()).to(dev, non_blocking=False)
    return resample_items_gpu(items_dev, len(regions), Hd, Wd, mean, std, out)
