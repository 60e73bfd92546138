#!/usr/bin/env python3
"""Probe: time of one conv launch as a function of K (Cin) at fixed M and Cout, per tile config.
time = overhead + slope * (K / 64): separates the main-loop rate from per-workgroup prologue / epilogue cost."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from distributed_llms_amd.ops import conv as C


def timed(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    dev = "cuda"
    N, H, W, Cout = 256, 12, 160, int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for cfg in (1128, 8128):
        row = {}
        for Cin in (64, 128, 256, 512):
            x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
            w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02).to(torch.bfloat16)
            b = torch.zeros(Cout, device=dev)
            row[Cin * 9 // 64] = round(timed(lambda: C.conv2d_nhwc(x, w, b, stride=1, padding=1, act="relu", cfg=cfg)), 4)
        ks = sorted(row)
        slope = (row[ks[-1]] - row[ks[0]]) / (ks[-1] - ks[0])
        print(json.dumps({"cfg": cfg, "M": N * H * W, "Cout": Cout, "ms_by_ktiles": row, "ms_per_ktile": round(slope, 5),
                          "overhead_ms": round(row[ks[0]] - slope * ks[0], 4)}), flush=True)


if __name__ == "__main__":
    main()
