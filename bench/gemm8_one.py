#!/usr/bin/env python3
"""One shape of the persistent GEMM kernel, a few launches: the target of rocprofv3 counter runs."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from distributed_llms_amd.ops import conv as C

M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8192, 8192, 8192)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
b = torch.randn(N, device="cuda")
for _ in range(iters):
    y = C.linear_bf16(x, w, b, cfg=C.CFG_GEMM8)
torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
