#!/usr/bin/env python3
"""Persistent 256x256 LDS-DMA GEMM (csrc/kernels/gemm_bf16.hip) against the 64x128 conv_igemm path and hipBLASLt on the
linear layers of the recognizers and the wide convolutions: correctness first (fp32 reference on a slice), then
interleaved timing rounds in one process on random data."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F

from distributed_llms_amd.ops import conv as C


def timed(fn, iters=8):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    quick = "--quick" in sys.argv
    shapes = [("crnn lstm proj", 256000, 512, 2048, "none", False), ("crnn lstm proj (small batch)", 20000, 512, 2048, "none", False),
              ("vit qkv", 147712, 768, 2304, "none", False), ("vit fc1", 147712, 768, 3072, "gelu", False),
              ("vit fc2", 147712, 3072, 768, "none", True), ("vit proj", 147712, 768, 768, "none", True),
              ("square 4096", 4096, 4096, 4096, "none", False), ("square 8192", 8192, 8192, 8192, "none", False)]
    if quick:
        shapes = shapes[:3]
    for name, M, K, N, act, use_res in shapes:
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        bb = b.to(torch.bfloat16)
        res = torch.randn(M, N, device="cuda").to(torch.bfloat16) if use_res else None
        y8 = C.linear_bf16(x, w, b, act=act, res=res, cfg=C.CFG_GEMM8)
        rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M), torch.randint(0, M, (400,))]).cuda()
        ref = x[rows].float() @ w.float().t() + b
        if use_res:
            ref = ref + res[rows].float()
        if act == "gelu":
            ref = F.gelu(ref)
        err = (y8[rows].float() - ref).abs()
        tol = 0.02 * ref.abs() + 0.05
        bad = int((err > tol).sum())
        old_auto = C.GEMM8_AUTO
        C.GEMM8_AUTO = False
        f8 = lambda: C.linear_bf16(x, w, b, act=act, res=res, cfg=C.CFG_GEMM8)
        fo = lambda: C.linear_bf16(x, w, b, act=act, res=res)
        if act == "gelu":
            fl = lambda: torch._addmm_activation(bb, x, w.t(), use_gelu=True)
        elif use_res:
            fl = lambda: torch.addmm(res, x, w.t()).add_(bb)
        else:
            fl = lambda: F.linear(x, w, bb)
        for f in (f8, fo, fl):
            f(); f()
        t8, to, tl = [], [], []
        for _ in range(5):
            t8.append(timed(f8)); to.append(timed(fo)); tl.append(timed(fl))
        C.GEMM8_AUTO = old_auto
        fl_ = 2.0 * M * K * N
        med = lambda v: sorted(v)[len(v) // 2]
        print(json.dumps({"layer": name, "M": M, "K": K, "N": N, "act": act, "res": use_res, "bad": bad,
                          "max_err": round(float(err.max()), 4),
                          "gemm8_ms": round(med(t8), 4), "igemm_ms": round(med(to), 4), "lib_ms": round(med(tl), 4),
                          "gemm8_tflops": round(fl_ / med(t8) / 1e9, 1), "igemm_tflops": round(fl_ / med(to) / 1e9, 1),
                          "lib_tflops": round(fl_ / med(tl) / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
