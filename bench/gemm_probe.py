#!/usr/bin/env python3
"""Plain-GEMM probe: this repository's conv_igemm 1x1 path vs the library GEMM (hipBLASLt through torch) on the
linear layers of the recognizers.  Decides where `linear_bf16` should call the library (plain GEMM + bias)."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F

from distributed_llms_amd.ops.conv import linear_bf16


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    shapes = [("crnn lstm proj", 256000, 512, 2048), ("crnn lstm proj (small batch)", 20000, 512, 2048), ("crnn fc", 256000, 512, 96),
              ("vit qkv", 147712, 768, 2304), ("vit fc1", 147712, 768, 3072), ("vit fc2", 147712, 3072, 768),
              ("dec qkv (256 crops)", 256, 1024, 3072), ("dec lm head", 256, 1024, 50272)]
    for name, M, K, N in shapes:
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        bb = b.to(torch.bfloat16)
        ours = timed(lambda: linear_bf16(x, w, b))
        lib = timed(lambda: F.linear(x, w, bb))
        fl = 2.0 * M * K * N
        print(json.dumps({"layer": name, "M": M, "K": K, "N": N, "ours_ms": round(ours, 4), "lib_ms": round(lib, 4),
                          "ours_tflops": round(fl / ours / 1e9, 1), "lib_tflops": round(fl / lib / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
