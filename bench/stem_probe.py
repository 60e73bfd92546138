"""Isolated timing of the first-layer kernels (conv_stem.hip, cfg 9008) against the generic implicit-GEMM tiles."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_llms_amd.ops import conv as C

dev = "cuda"
cases = [("rec c1 8->64 s2", 1, 48, 330000, 8, 2, 64), ("det stem1 8->32 s2", 25, 1344, 960, 8, 2, 32),
         ("det stem2 32->32 s1", 25, 672, 480, 32, 1, 32), ("det stem3 32->64 s2", 25, 672, 480, 32, 2, 64)]
for name, B, H, W, Cin, st, Cout in cases:
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.zeros(Cout, device=dev)
    row = [name]
    for cfg in (9008, 32, 64, 1064, 1128):
        try:
            for _ in range(3):
                y = C.conv2d_nhwc(x, w, b, stride=st, padding=1, act="relu", cfg=cfg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                y = C.conv2d_nhwc(x, w, b, stride=st, padding=1, act="relu", cfg=cfg)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            gb = (x.numel() + y.numel()) * 2 / 1e9
            row.append(f"cfg {cfg}: {ms:.3f} ms ({gb / ms * 1e3:.0f} GB/s)")
        except Exception as e:
            row.append(f"cfg {cfg}: n/a")
    print(" | ".join(row), flush=True)
