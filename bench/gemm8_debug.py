#!/usr/bin/env python3
"""Error-pattern dump for the persistent GEMM kernel on exact integer data (debug aid)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from distributed_llms_amd.ops import conv as C

def run(M, K, N, kmask=None, bias=False):
    x = torch.randint(-3, 4, (M, K), device="cuda").float()
    if kmask is not None:
        x = x * kmask.cuda().float()[None, :]
    w = torch.randint(-3, 4, (N, K), device="cuda").float()
    b = torch.randint(-5, 6, (N,), device="cuda").float() if bias else None
    y = C.linear_bf16(x.to(torch.bfloat16), w.to(torch.bfloat16), b, cfg=C.CFG_GEMM8, out_dtype=torch.float32)
    ref = x @ w.t() + (b if bias else 0)
    ok = (y == ref)
    print(f"M{M} K{K} N{N} bias={bias} kmask={'y' if kmask is not None else 'n'}: correct {float(ok.float().mean()):.4f}", flush=True)
    if not bool(ok.all()):
        print("  by n%32 :", [round(float(ok[:, i::32].float().mean()), 2) for i in range(32)])
        print("  by n//32:", [round(float(ok[:, i*32:(i+1)*32].float().mean()), 2) for i in range(N // 32)])
        print("  by m%16 :", [round(float(ok[i::16].float().mean()), 2) for i in range(16)])
        print("  by m//16:", [round(float(ok[i*16:(i+1)*16].float().mean()), 2) for i in range(min(M // 16, 16))])
        bad = (~ok).nonzero()[:6]
        for m, n in bad.tolist():
            print(f"   y[{m},{n}] = {float(y[m, n])} ref {float(ref[m, n])}")
        # is y a permutation of ref columns within the row?
        r0 = ref[0].tolist(); y0 = y[0].tolist()
        print("  row0 y[:40]  ", [int(v) for v in y0[:40]])
        print("  row0 ref[:40]", [int(v) for v in r0[:40]])

torch.manual_seed(0)
run(256, 128, 256)
k0 = torch.zeros(128); k0[:64] = 1
run(256, 128, 256, kmask=k0)
k1 = torch.zeros(128); k1[64:] = 1
run(256, 128, 256, kmask=k1)
kc = torch.zeros(128); kc[:8] = 1
run(256, 128, 256, kmask=kc)
run(256, 128, 256, bias=True)
