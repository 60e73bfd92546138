#!/usr/bin/env python3
"""Probe: the Cout = 64 3x3 layers of the detector (layer1, FPN smooth convs, head) per kernel configuration."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from distributed_llms_amd.ops import conv as C


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
    dev = "cuda"
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    shapes = [("layer1 64->64 @1/4", 336, 240, 64, 64), ("head1 256->64 @1/4", 336, 240, 256, 64),
              ("p3 256->64 @1/8", 168, 120, 256, 64), ("layer2 128->128 @1/8", 168, 120, 128, 128), ("layer3 256->256 @1/16", 84, 60, 256, 256),
              ("crnn c3 128->256 12 rows", 12, 5000, 128, 256), ("crnn c2-like 64->128 s1 24 rows", 24, 5000, 64, 128)]
    for name, H, W, Cin, Cout in shapes:
        Bx = B if W < 1000 else 8
        x = torch.randn(Bx, H, W, Cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.zeros(Cout, device=dev)
        fl = 2.0 * (B if W < 1000 else 8) * H * W * Cout * 9 * Cin
        row = {"layer": name, "gflop": round(fl / 1e9, 1)}
        ref = None
        for cfg in (0, 3064, 3264, 3464, 3428, 3128, 1128, 8256):
            try:
                y = C.conv2d_nhwc(x, w, b, stride=1, padding=1, act="relu", cfg=cfg)
                if ref is None:
                    ref = y.float()
                else:
                    assert (y.float() - ref).abs().max().item() < 0.1, cfg
                ms = timed(lambda: C.conv2d_nhwc(x, w, b, stride=1, padding=1, act="relu", cfg=cfg))
                row[str(cfg)] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}
            except Exception as e:  # noqa: BLE001
                row[str(cfg)] = str(e)[:60]
        print(json.dumps(row), flush=True)
    # stride-2 3x3 layers that run on the 64 x 128 direct-weight tile by default
    for name, n, H, W, Cin, Cout in (("crnn c2 64->128 s2 24 rows", 300, 24, 960, 64, 128), ("layer2.0.c1 64->128 s2 @1/4", B, 336, 240, 64, 128),
                                     ("layer3.0.c1 128->256 s2 @1/8", B, 168, 120, 128, 256)):
        x = torch.randn(n, H, W, Cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.zeros(Cout, device=dev)
        fl = 2.0 * n * (H // 2) * (W // 2) * Cout * 9 * Cin
        row = {"layer": name, "gflop": round(fl / 1e9, 1)}
        ref = None
        for cfg in (0, 3064, 3264, 3464, 3428, 3128, 1128, 8256):
            try:
                y = C.conv2d_nhwc(x, w, b, stride=2, padding=1, act="relu", cfg=cfg)
                if ref is None:
                    ref = y.float()
                else:
                    assert (y.float() - ref).abs().max().item() < 0.1, cfg
                ms = timed(lambda: C.conv2d_nhwc(x, w, b, stride=2, padding=1, act="relu", cfg=cfg))
                row[str(cfg)] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}
            except Exception as e:  # noqa: BLE001
                row[str(cfg)] = str(e)[:60]
        print(json.dumps(row), flush=True)
    # FPN laterals (1x1 into 256 channels, + nearest-2x residual): streaming kernel (9100) against the generic tiles
    for name, H, W, Cin in (("in2 64->256 @1/4 +up2", 336, 240, 64), ("in3 128->256 @1/8 +up2", 168, 120, 128)):
        x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(256, 1, 1, Cin, device=dev) * 0.05).to(torch.bfloat16)
        b = torch.zeros(256, device=dev)
        r = torch.randn(B, H // 2, W // 2, 256, device=dev).to(torch.bfloat16)
        out = torch.empty(B, H, W, 256, device=dev, dtype=torch.bfloat16)
        gb = (x.numel() + out.numel() + r.numel()) * 2 / 1e9
        row = {"layer": name, "gbytes": round(gb, 3)}
        ref = None
        for cfg in (9100, 3900, 1128):
            try:
                y = C.conv2d_nhwc(x, w, b, res=r, res_mode=C.RES_UP2, out=out, cfg=cfg)
                if ref is None:
                    ref = y.float().clone()
                else:
                    assert (y.float() - ref).abs().max().item() < 0.1, cfg
                ms = timed(lambda: C.conv2d_nhwc(x, w, b, res=r, res_mode=C.RES_UP2, out=out, cfg=cfg))
                row[str(cfg)] = {"ms": round(ms, 4), "gbps": round(gb / ms * 1e3, 1)}
            except Exception as e:  # noqa: BLE001
                row[str(cfg)] = str(e)[:60]
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
