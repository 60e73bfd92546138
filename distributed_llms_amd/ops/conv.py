"""NHWC bf16 convolution / GEMM ops backed by the MFMA implicit-GEMM kernel.

``conv2d_nhwc`` is the single entry point for every GEMM-shaped op of the OCR
stack (DBNet convs, FPN fusion, transposed convs, CRNN convs, LSTM input
projections, ViT / decoder linears).  On a CUDA(HIP) tensor it launches
``dlamd_conv2d`` (csrc/kernels/conv_igemm.hip); on a CPU tensor it evaluates
``conv2d_nhwc_reference`` (plain PyTorch fp32), which is also the numerics
oracle for the GPU tests.

Reference parity: the reference runs these layers through magic-pdf's CPU
torch models (services/ocr_service/ocr_service.py:215-220,
download_models_hf.py:36-44); nothing of that code exists in-tree, so the op
contract here is new.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
import torch.nn.functional as F

from . import _loader

ACT = {"none": 0, "relu": 1, "hardswish": 2, "gelu": 3, "sigmoid": 4, "tanh": 5}
OUT_PLAIN, OUT_PIXEL_SHUFFLE, OUT_REPLICATE, OUT_PROB = 0, 1, 2, 3
RES_NONE, RES_SAME, RES_UP2 = 0, 1, 2

IntPair = Union[int, Tuple[int, int]]

USE_WEIGHT_FRAGMENTS = True
LIBRARY_GEMM_MIN_ROWS = 0         # > 0: plain GEMMs with at least this many rows go to hipBLASLt (comparison runs only)
# The persistent 256 x 256 LDS-DMA kernel (csrc/kernels/gemm_bf16.hip) takes a layer when its tile grid fills the
# chip and the 256-wide column tile wastes little: every plain linear layer of the recognizers and the wide 3x3
# convolutions.  cfg 8256 forces it, any other explicit cfg keeps conv_igemm.
GEMM8_AUTO = True
GEMM8_MIN_TILES = 192
CFG_GEMM8 = 8256


def gemm8_wanted(M: int, N: int, K: int, cfg: int) -> bool:
    if cfg == CFG_GEMM8:
        return True
    if cfg != 0 or not GEMM8_AUTO:
        return False
    tiles_n = (N + 255) // 256
    waste = tiles_n * 256 / N
    return waste <= 1.34 and ((M + 255) // 256) * tiles_n >= GEMM8_MIN_TILES


def weight_fragments(w: torch.Tensor) -> torch.Tensor:
    """[Cout, kh, kw, Cin] bf16 -> [Npad/16, Kpad/32, 64, 8]: lane l of a wave holds rows n = 16*tile + (l & 15),
    k = 32*step + 8*(l >> 4) .. +8 - exactly the A operand of ``mfma_f32_16x16x32_bf16``, so the kernel reads
    weights straight from L2 into registers as contiguous 1 KiB wave loads (no LDS staging for weights).
    N and K are padded to multiples of 128 with zeros (no bounds checks in the kernel)."""
    # cached on the tensor OBJECT (module weights persist; a data_ptr key could be recycled by the allocator)
    cached = getattr(w, "_dlamd_frag", None)
    if cached is not None and cached[0] == w._version:
        return cached[1]
    Cout = w.shape[0]
    K = w.numel() // Cout
    Np, Kp = (Cout + 127) // 128 * 128, (K + 127) // 128 * 128
    w2 = torch.zeros((Np, Kp), dtype=torch.bfloat16, device=w.device)
    w2[:Cout, :K] = w.reshape(Cout, K)
    f = w2.view(Np // 16, 16, Kp // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(Np // 16, Kp // 32, 64, 8)
    try:
        w._dlamd_frag = (w._version, f)
    except Exception:  # noqa: BLE001 - e.g. tensors that refuse attributes; just rebuild next time
        pass
    return f


# bench/conv_layer_breakdown.py sets this to a list: every launch appends (shape key, start event, end event)
PROFILE_SINK = None


def _pair(v: IntPair) -> Tuple[int, int]:
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


def _act_ref(v: torch.Tensor, act: str) -> torch.Tensor:
    if act == "relu":
        return F.relu(v)
    if act == "hardswish":
        return F.hardswish(v)
    if act == "gelu":
        return F.gelu(v)
    if act == "sigmoid":
        return torch.sigmoid(v)
    if act == "tanh":
        return torch.tanh(v)
    return v


def out_hw(H: int, W: int, kh: int, kw: int, sh: int, sw: int, ph: int, pw: int) -> Tuple[int, int]:
    return (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1


def conv2d_nhwc_reference(
    x: torch.Tensor,
    w: torch.Tensor,
    bias: Optional[torch.Tensor] = None,
    *,
    stride: IntPair = 1,
    padding: IntPair = 0,
    act: str = "none",
    res: Optional[torch.Tensor] = None,
    res_mode: int = RES_NONE,
    out_mode: int = OUT_PLAIN,
    out_f: int = 1,
    thr: float = 0.3,
    out_dtype: torch.dtype = torch.bfloat16,
):
    """fp32 PyTorch oracle.  x: [B,H,W,Cin], w: [Cout,kh,kw,Cin]."""
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    xf = x.float().permute(0, 3, 1, 2)
    wf = w.float().permute(0, 3, 1, 2)
    y = F.conv2d(xf, wf, None if bias is None else bias.float(), stride=(sh, sw), padding=(ph, pw))
    if res is not None and res_mode != RES_NONE:
        r = res.float().permute(0, 3, 1, 2)
        if res_mode == RES_UP2:
            r = F.interpolate(r, scale_factor=2, mode="nearest")[:, :, : y.shape[2], : y.shape[3]]
        y = y + r[:, : y.shape[1]]
    y = _act_ref(y, act)
    if out_mode == OUT_PROB:
        # Cout == 4 -> 2x2 pixels: channel (dy*2+dx)
        B, _, Ho, Wo = y.shape
        p = y.view(B, 2, 2, Ho, Wo).permute(0, 3, 1, 4, 2).reshape(B, 2 * Ho, 2 * Wo)
        return p.to(out_dtype), (p > thr).to(torch.uint8)
    if out_mode == OUT_PIXEL_SHUFFLE:
        B, C4, Ho, Wo = y.shape
        Cp = C4 // 4
        y = y.view(B, 2, 2, Cp, Ho, Wo).permute(0, 3, 4, 1, 5, 2).reshape(B, Cp, 2 * Ho, 2 * Wo)
    elif out_mode == OUT_REPLICATE and out_f > 1:
        y = F.interpolate(y, scale_factor=out_f, mode="nearest")
    return y.permute(0, 2, 3, 1).contiguous().to(out_dtype)


def conv2d_nhwc(
    x: torch.Tensor,
    w: torch.Tensor,
    bias: Optional[torch.Tensor] = None,
    *,
    stride: IntPair = 1,
    padding: IntPair = 0,
    act: str = "none",
    res: Optional[torch.Tensor] = None,
    res_mode: int = RES_NONE,
    out: Optional[torch.Tensor] = None,
    out_coff: int = 0,
    out_mode: int = OUT_PLAIN,
    out_f: int = 1,
    thr: float = 0.3,
    out_dtype: torch.dtype = torch.bfloat16,
    cfg: int = 0,
    wfrag: Optional[torch.Tensor] = None,
):
    """Fused conv (+bias +residual +activation) on NHWC bf16 tensors.

    x   : [B, H, W, Cin]   (last-dim stride 1; the pixel stride may exceed Cin,
                            i.e. x may be a channel slice of a wider buffer)
    w   : [Cout, kh, kw, Cin] bf16
    out : optional pre-allocated destination [B, Ho', Wo', Ctot]; the result is
          written at channel offset ``out_coff`` (fused concat).
    Returns the output tensor (or ``(prob, mask)`` for ``OUT_PROB``).
    """
    if not x.is_cuda:
        y = conv2d_nhwc_reference(
            x, w, bias, stride=stride, padding=padding, act=act, res=res, res_mode=res_mode,
            out_mode=out_mode, out_f=out_f, thr=thr, out_dtype=out_dtype,
        )
        if out is not None and out_mode != OUT_PROB:
            out[..., out_coff : out_coff + y.shape[-1]] = y
            return out
        return y

    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16, "conv2d_nhwc expects bf16"
    assert x.dim() == 4 and w.dim() == 4 and x.stride(3) == 1
    B, H, W, Cin = x.shape
    Cout, kh, kw, Cin_w = w.shape
    assert Cin_w == Cin, f"weight Cin {Cin_w} != input Cin {Cin}"
    assert w.is_contiguous()
    x_cstride = x.stride(2)
    assert x.stride(1) == W * x_cstride and x.stride(0) == H * W * x_cstride, "x must be pixel-contiguous"
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    Ho, Wo = out_hw(H, W, kh, kw, sh, sw, ph, pw)

    mask = None
    if out_mode == OUT_PROB:
        assert Cout == 4
        y = torch.empty((B, 2 * Ho, 2 * Wo), dtype=torch.bfloat16, device=x.device) if out is None else out
        mask = torch.empty((B, 2 * Ho, 2 * Wo), dtype=torch.uint8, device=x.device)
        y_cstride, y_coff = 4, 0
    else:
        if out_mode == OUT_PIXEL_SHUFFLE:
            oshape = (B, 2 * Ho, 2 * Wo, Cout // 4)
        elif out_mode == OUT_REPLICATE:
            oshape = (B, out_f * Ho, out_f * Wo, Cout)
        else:
            oshape = (B, Ho, Wo, Cout)
        if out is None:
            y = torch.empty(oshape, dtype=out_dtype, device=x.device)
            y_coff = 0
        else:
            y = out
            assert tuple(y.shape[:3]) == oshape[:3] and y.is_contiguous() and y.dtype == out_dtype
            assert out_coff + oshape[3] <= y.shape[3]
            y_coff = out_coff
        y_cstride = y.shape[3]
    res_cstride = 0
    if res is not None and res_mode != RES_NONE:
        assert res.dtype == torch.bfloat16 and res.is_contiguous()
        res_cstride = res.shape[3]
        if res_mode == RES_SAME:
            assert tuple(res.shape[:3]) == (B, Ho, Wo)
        else:
            assert tuple(res.shape[:3]) == (B, (Ho + 1) // 2, (Wo + 1) // 2)
        assert res.shape[3] >= Cout
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= Cout and bias.is_contiguous()

    act_id = ACT[act]
    rm = res_mode if res is not None else RES_NONE
    if gemm8_wanted(B * Ho * Wo, Cout, kh * kw * Cin, cfg) and _loader.lib().dlamd_gemm8_serves(
            B * Ho * Wo, Cout, kh * kw * Cin, Cin, x_cstride, kh, kw, act_id, rm, out_mode, y_cstride, y_coff):
        prof = PROFILE_SINK
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = _loader.lib().dlamd_gemm8(
            _loader.ptr(x), _loader.ptr(w), _loader.ptr(bias), _loader.ptr(res if rm != RES_NONE else None), _loader.ptr(y),
            B, H, W, Cin, x_cstride, Cout, kh, kw, sh, sw, ph, pw, act_id, rm, res_cstride, y_cstride, y_coff,
            1 if out_dtype == torch.float32 else 0, _loader.stream_ptr())
        if rc != -12:          # -12: an operand of 2 GiB or more (32-bit descriptor offsets) -> conv_igemm below
            _loader.check(rc, "dlamd_gemm8")
            if prof is not None:
                e1.record()
                prof.append(((B * Ho * Wo, Cout, kh * kw * Cin, f"gemm8 {kh}x{kw} s{sh}{sw} {act} rm{rm}"), e0, e1))
            return y
    elif cfg == CFG_GEMM8:
        raise _loader.KernelLibraryError("cfg 8256: the shape is outside the contract of dlamd_gemm8")

    frag = None
    if USE_WEIGHT_FRAGMENTS:
        frag = wfrag if wfrag is not None else weight_fragments(w)
        assert frag.is_contiguous() and frag.dtype == torch.bfloat16 and frag.dim() == 4 and frag.shape[2:] == (64, 8)
    prof = PROFILE_SINK
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _loader.lib().dlamd_conv2d(
        _loader.ptr(x), _loader.ptr(w), _loader.ptr(bias), _loader.ptr(res if res_mode != RES_NONE else None),
        _loader.ptr(y), _loader.ptr(mask),
        B, H, W, Cin, x_cstride, Cout, kh, kw, sh, sw, ph, pw,
        ACT[act], res_mode, res_cstride, out_mode, out_f, y_cstride, y_coff,
        1 if out_dtype == torch.float32 else 0, float(thr), cfg,
        _loader.ptr(frag), frag.shape[0] if frag is not None else 0, frag.shape[1] if frag is not None else 0,
        _loader.stream_ptr(),
    )
    _loader.check(rc, "dlamd_conv2d")
    if prof is not None:
        e1.record()
        prof.append(((B * Ho * Wo, Cout, kh * kw * Cin, f"{kh}x{kw} s{sh}{sw} {act} om{out_mode} rm{res_mode}"), e0, e1))
    if out_mode == OUT_PROB:
        return y, mask
    return y


# comparison runs only (LIBRARY_GEMM_MIN_ROWS > 0): plain bias-only GEMMs at least this wide go to the library too
LIBRARY_GEMM_MIN_COLS = 16384


def linear_bf16(
    x: torch.Tensor,
    w: torch.Tensor,
    bias: Optional[torch.Tensor] = None,
    *,
    act: str = "none",
    res: Optional[torch.Tensor] = None,
    out_dtype: torch.dtype = torch.bfloat16,
    cfg: int = 0,
) -> torch.Tensor:
    """y = act(x @ w^T + bias (+ res)): the persistent GEMM kernel for large layers, the 1x1 path of conv_igemm
    for small-M / narrow ones.

    x: [..., K] bf16, w: [N, K] bf16, bias: [N] f32, res: [..., N] bf16.
    """
    lead = x.shape[:-1]
    K = x.shape[-1]
    N = w.shape[0]
    # Library GEMMs (hipBLASLt through torch) are a comparison path, off by default (LIBRARY_GEMM_MIN_ROWS = 0): every
    # linear layer of inference runs on the in-repo kernels below.
    if (LIBRARY_GEMM_MIN_ROWS > 0 and x.is_cuda and out_dtype == torch.bfloat16 and w.is_contiguous()
            and (x.numel() // K >= LIBRARY_GEMM_MIN_ROWS
                 or (N >= LIBRARY_GEMM_MIN_COLS and x.numel() // K >= 64 and act == "none" and res is None))
            and ((act == "none") or (act == "gelu" and res is None and bias is not None))):
        b16 = None
        if bias is not None:
            cached = getattr(bias, "_dlamd_bf16", None)
            if cached is None or cached[0] != bias._version:
                cached = (bias._version, bias.detach().to(torch.bfloat16))
                try:
                    bias._dlamd_bf16 = cached
                except Exception:  # noqa: BLE001
                    pass
            b16 = cached[1][:N]
        x2d = x.reshape(-1, K)
        if act == "gelu":
            # the library's bias + GELU epilogue (tanh form: differs from erf by < 3e-4, below bf16 resolution):
            # ViT fc1 0.89 ms against 1.60 ms for the fused in-repo kernel
            return torch._addmm_activation(b16, x2d, w.t(), use_gelu=True).reshape(*lead, N)
        if res is not None:
            # residual as the GEMM's C operand (beta = 1), bias in place afterwards: ViT fc2 0.81 ms against 1.07
            y = torch.addmm(res.reshape(-1, N), x2d, w.t())
            return (y.add_(b16) if b16 is not None else y).reshape(*lead, N)
        return F.linear(x2d, w, b16).reshape(*lead, N)
    x2 = x.reshape(1, 1, -1, K)
    r2 = None if res is None else res.reshape(1, 1, -1, N)
    # the MFMA fragment image of the weights is cached on the weight tensor object: it must be taken from ``w`` itself,
    # not from the 4-d view made below (a fresh object on every call, whose cache dies with it - every small-M linear of
    # a decode step then rebuilt its fragments: a fill, a copy and a permute per layer, 1.25 of 4.2 ms per step)
    frag = None
    if x.is_cuda and USE_WEIGHT_FRAGMENTS and not gemm8_wanted(x2.shape[2], N, K, cfg):
        frag = weight_fragments(w)
    y = conv2d_nhwc(
        x2, w.view(N, 1, 1, K), bias, act=act, res=r2, res_mode=RES_SAME if res is not None else RES_NONE,
        out_dtype=out_dtype, cfg=cfg, wfrag=frag,
    )
    return y.reshape(*lead, N)


def db_head_a2_fragments(w3: torch.Tensor, cmid: int = 64) -> torch.Tensor:
    """MFMA A fragments [8, 64, 8] of the fused DB head's second GEMM (csrc/kernels/db_head.hip).

    ``w3``: the last deconv as GEMM rows [4 (r = dy, dx), cmid].  A2[(q, r), (g, c')] = [q == g] * w3[r, c'] over the
    256 rows (g, c') of the first deconv; its K order is permuted so that K-step s, lane group fq, element j is row
    ``(2s + j // 4) * 16 + 4 fq + j % 4`` - the accumulator values a lane holds after the first GEMM."""
    assert cmid == 64 and tuple(w3.shape[-1:]) == (cmid,) and w3.numel() == 4 * cmid
    w3 = w3.reshape(4, cmid).float()
    a2 = torch.zeros(16, 4 * cmid, device=w3.device)
    for q in range(4):
        a2[4 * q : 4 * q + 4, q * cmid : (q + 1) * cmid] = w3
    lane = torch.arange(64, device=w3.device)
    row, fq = lane % 16, lane // 16
    s = torch.arange(8, device=w3.device)[:, None, None]
    j = torch.arange(8, device=w3.device)[None, None, :]
    n = (2 * s + j // 4) * 16 + 4 * fq[None, :, None] + j % 4                 # [8, 64, 8]
    return a2[row[None, :, None].expand_as(n), n].to(torch.bfloat16).contiguous()


def db_head_fused(h: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, a2frag: torch.Tensor, b3: torch.Tensor,
                  thr: float = 0.3) -> Tuple[torch.Tensor, torch.Tensor]:
    """Deconv(64->64, ReLU) + deconv(64->1) + sigmoid + threshold in one kernel.  h [B,Hq,Wq,64] bf16 ->
    (prob [B,4Hq,4Wq] bf16, mask uint8); same arithmetic as the two ``conv2d_nhwc`` launches it replaces."""
    assert h.is_cuda and h.dtype == torch.bfloat16 and h.is_contiguous() and h.dim() == 4 and h.shape[3] == 64
    B, Hq, Wq, _ = h.shape
    assert Wq % 2 == 0, "fused DB head needs an even quarter-resolution width"
    assert w2.dtype == torch.bfloat16 and w2.is_contiguous() and w2.numel() == 256 * 64
    assert b2.dtype == torch.float32 and b2.numel() == 256 and b3.dtype == torch.float32 and b3.numel() == 4
    assert a2frag.dtype == torch.bfloat16 and a2frag.is_contiguous() and tuple(a2frag.shape) == (8, 64, 8)
    prob = torch.empty((B, 4 * Hq, 4 * Wq), dtype=torch.bfloat16, device=h.device)
    mask = torch.empty((B, 4 * Hq, 4 * Wq), dtype=torch.uint8, device=h.device)
    rc = _loader.lib().dlamd_db_head_fused(_loader.ptr(h), _loader.ptr(w2), _loader.ptr(b2), _loader.ptr(a2frag),
                                           _loader.ptr(b3), _loader.ptr(prob), _loader.ptr(mask), B, Hq, Wq, float(thr),
                                           _loader.stream_ptr())
    _loader.check(rc, "dlamd_db_head_fused")
    return prob, mask


def block_head_split(logits: torch.Tensor, channels: int = 2):
    """Block-head logits [B, H, W, 8] bf16 -> per channel (figures, formulas) the probability map [B, H, W] bf16 and the bit
    mask [B, H, W] u8 (logit > 0) the connected-component pass reads: one kernel instead of slicing, sigmoid, compare
    and two casts per channel.  -> [(prob, mask)] * channels."""
    assert logits.dtype == torch.bfloat16 and logits.is_contiguous() and logits.shape[-1] == 8 and channels in (1, 2)
    shape = tuple(logits.shape[:-1])
    if not logits.is_cuda:
        out = []
        for c in range(channels):
            l = logits[..., c].float()
            out.append((torch.sigmoid(l).to(torch.bfloat16).contiguous(), (l > 0).to(torch.uint8).contiguous()))
        return out
    out = [(torch.empty(shape, dtype=torch.bfloat16, device=logits.device), torch.empty(shape, dtype=torch.uint8, device=logits.device))
           for _ in range(channels)]
    p1, m1 = out[1] if channels == 2 else (None, None)
    rc = _loader.lib().dlamd_block_head_split(_loader.ptr(logits), _loader.ptr(out[0][0]), _loader.ptr(out[0][1]),
                                              _loader.ptr(p1), _loader.ptr(m1), logits.numel() // 8, _loader.stream_ptr())
    _loader.check(rc, "dlamd_block_head_split")
    return out
