#!/usr/bin/env python3
"""In-kernel section timing of the persistent GEMM (diagnostic build with s_memtime stamps): where an interval's
cycles go for a wave of each group.  Shares are meaningful, the stamped build's run time is not."""
import ctypes
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from distributed_llms_amd.ops import _loader

M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8192, 8192, 8192)
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
tiles = ((M + 255) // 256) * ((N + 255) // 256)
grid = min(tiles, 256)
st = torch.zeros(grid, 8, 8, dtype=torch.int64, device="cuda")
lib = _loader.lib()
lib.dlamd_gemm8_stamp.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
for _ in range(3):
    rc = lib.dlamd_gemm8_stamp(_loader.ptr(x), _loader.ptr(w), _loader.ptr(y), M, N, K, _loader.ptr(st), _loader.stream_ptr())
    assert rc == 0, rc
torch.cuda.synchronize()
ktiles = (tiles + grid - 1) // grid * (K // 64)
s = st.double() / ktiles
names = ["I1 epilogue + ds_read issue", "I1 dma issue", "I1 vmcnt(8)", "I1 lgkmcnt(0)", "I1 barrier 1", "I1 mfma", "I1 barrier 2", "I2 all"]
for g, ws in (("G0 (waves 0-3)", slice(0, 4)), ("G1 (waves 4-7)", slice(4, 8))):
    m = s[:, ws].mean(dim=(0, 1))
    print(g, {n: round(float(v)) for n, v in zip(names, m)}, "sum per K-tile", round(float(m.sum())))
