"""Persistent LDS-DMA GEMM / implicit-GEMM conv kernel (csrc/kernels/gemm_bf16.hip) against a plain PyTorch fp32
reference of the same op: tile edges in M and N, every epilogue, several tiles per workgroup, 3x3 / strided / 1x1
gathers, concat-offset stores."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from distributed_llms_amd.ops import conv as C


def _check(y, ref, rel=0.02, abs_=0.06):
    err = (y.float() - ref.float()).abs()
    tol = rel * ref.float().abs() + abs_
    assert not bool((err > tol).any()), f"max err {float(err.max()):.4f}, {int((err > tol).sum())} elements off"
    assert float(err.max()) < 8 * abs_ + 8 * rel * float(ref.abs().max())


GEMM_CASES = [
    # M, K, N, act, res, f32, grid
    (256, 128, 256, "none", False, False, 0),
    (1000, 192, 264, "none", False, False, 0),
    (1000, 192, 264, "relu", False, False, 2),
    (777, 512, 520, "gelu", False, False, 3),
    (3000, 256, 768, "none", True, False, 4),
    (515, 128, 1024, "relu", True, False, 0),
    (2049, 320, 256, "none", False, True, 5),
    (64, 1024, 2304, "none", False, False, 0),
]


@pytest.mark.parametrize("M,K,N,act,use_res,f32,grid", GEMM_CASES)
def test_gemm8_matches_reference(M, K, N, act, use_res, f32, grid, monkeypatch):
    if grid:
        monkeypatch.setenv("DLAMD_GEMM8_GRID", str(grid))
    else:
        monkeypatch.delenv("DLAMD_GEMM8_GRID", raising=False)
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) if use_res else None
    y = C.linear_bf16(x, w, b, act=act, res=res, out_dtype=torch.float32 if f32 else torch.bfloat16, cfg=C.CFG_GEMM8)
    ref = x.float() @ w.float().t() + b
    if use_res:
        ref = ref + res.float()
    ref = F.relu(ref) if act == "relu" else F.gelu(ref) if act == "gelu" else ref
    assert y.shape == (M, N) and y.dtype == (torch.float32 if f32 else torch.bfloat16)
    _check(y, ref)


def test_gemm8_exact_integers_asymmetric():
    """A = shifted identity blocks, W small integers: every product is exact in bf16, so any swapped row / column /
    k-chunk shows up as a wrong integer."""
    M, K, N = 512, 256, 512
    x = torch.zeros(M, K, device="cuda")
    x[torch.arange(M), (torch.arange(M) * 3 + 1) % K] = 1.0
    x[torch.arange(M), (torch.arange(M) * 5 + 7) % K] += 2.0
    w = ((torch.arange(N, device="cuda")[:, None] * 7 + torch.arange(K, device="cuda")[None, :] * 3) % 13 - 6).float()
    y = C.linear_bf16(x.to(torch.bfloat16), w.to(torch.bfloat16), None, cfg=C.CFG_GEMM8)
    ref = x @ w.t()
    assert torch.equal(y.float(), ref)


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, act, res, coff/ctot
    (2, 37, 41, 64, 256, 3, 1, 1, "relu", False, (0, 256)),
    (1, 50, 30, 128, 256, 3, 2, 1, "relu", False, (0, 256)),
    (3, 24, 24, 64, 512, 3, 1, 1, "none", True, (0, 512)),
    (2, 33, 20, 192, 256, 1, 1, 0, "relu", False, (64, 384)),
    (1, 40, 40, 64, 264, 3, 1, 1, "none", False, (0, 264)),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,pad,act,use_res,cat", CONV_CASES)
def test_gemm8_conv_matches_reference(B, H, W, Cin, Cout, k, s, pad, act, use_res, cat, monkeypatch):
    monkeypatch.setenv("DLAMD_GEMM8_GRID", "3")
    g = torch.Generator(device="cuda").manual_seed(H * 31 + Cout)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, k, k, Cin, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(Cout, device="cuda", generator=g)
    Ho, Wo = C.out_hw(H, W, k, k, s, s, pad, pad)
    res = torch.randn(B, Ho, Wo, Cout, device="cuda", generator=g).to(torch.bfloat16) if use_res else None
    coff, ctot = cat
    out = torch.full((B, Ho, Wo, ctot), 7.0, device="cuda", dtype=torch.bfloat16)
    y = C.conv2d_nhwc(x, w, b, stride=s, padding=pad, act=act, res=res, res_mode=C.RES_SAME if use_res else C.RES_NONE,
                      out=out, out_coff=coff, cfg=C.CFG_GEMM8)
    ref = C.conv2d_nhwc_reference(x, w, b, stride=s, padding=pad, act=act, res=res,
                                  res_mode=C.RES_SAME if use_res else C.RES_NONE, out_dtype=torch.float32)
    _check(y[..., coff : coff + Cout], ref)
    if coff:
        assert bool((y[..., :coff] == 7.0).all())
    if coff + Cout < ctot:
        assert bool((y[..., coff + Cout :] == 7.0).all())
