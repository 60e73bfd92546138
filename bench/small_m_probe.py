"""Decoder-step linears (M = batch 256): tile / K-depth choices of the in-repo GEMM, isolated."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_llms_amd.ops import conv as C

dev = "cuda"
for M, K, N in [(256, 1024, 1024), (256, 1024, 3072), (256, 1024, 4096), (256, 4096, 1024), (256, 1024, 50304)]:
    x = torch.randn(1, 1, M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, 1, 1, K, device=dev) * 0.03).to(torch.bfloat16)
    b = torch.zeros(N, device=dev)
    frag = C.weight_fragments(w)
    row = [f"M{M} K{K} N{N}"]
    ref = None
    for cfg in (0, 2032, 4032, 1064, 1128):
        try:
            for _ in range(3):
                y = C.conv2d_nhwc(x, w, b, cfg=cfg, wfrag=frag)
            torch.cuda.synchronize()
            if ref is None:
                ref = y.float()
            err = float((y.float() - ref).abs().max())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                y = C.conv2d_nhwc(x, w, b, cfg=cfg, wfrag=frag)
            e1.record()
            torch.cuda.synchronize()
            row.append(f"cfg {cfg}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us (err {err:.3f})")
        except Exception as e:
            row.append(f"cfg {cfg}: n/a {str(e)[:40]}")
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    x2, w2 = x.view(M, K), w.view(N, K)
    for _ in range(3):
        torch.nn.functional.linear(x2, w2)
    t0.record()
    for _ in range(50):
        torch.nn.functional.linear(x2, w2)
    t1.record()
    torch.cuda.synchronize()
    row.append(f"hipBLASLt: {t0.elapsed_time(t1) / 50 * 1e3:.1f} us")
    print(" | ".join(row), flush=True)
