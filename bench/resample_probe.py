#!/usr/bin/env python3
"""Probe: the two shapes of the resample kernel in the serving path - 25 whole pages shrunk to detector scale, and
~1000 text-line crops resized to height 48 - against their HBM floors."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from distributed_llms_amd.ops import resample as rs


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
    dev = torch.device("cuda")
    B = 25
    pages = [torch.randint(0, 255, (3508, 2480), dtype=torch.uint8, device=dev) for _ in range(B)]
    regs = [rs.Region(p, (0.0, 0.0, 2480.0, 3508.0), 1344, 950) for p in pages]
    items = torch.from_numpy(rs.pack_items(regs).view(np.uint8).copy()).to(dev)
    ms = timed(lambda: rs.resample_items_gpu(items, B, 1344, 960))
    floor = (B * 3508 * 2480 + B * 1344 * 960 * 16) / 5e12 * 1e3
    print(json.dumps({"shape": "25 pages 3508x2480 -> 1344x950 (bucket 960)", "ms": round(ms, 4), "hbm_floor_ms": round(floor, 4)}), flush=True)
    # the fused form the engine runs: resample + first convolution (3 -> 32 for the detector, 3 -> 64 for the recogniser)
    w32 = (torch.randn(32, 3, 3, 8, device=dev) * 0.1).to(torch.bfloat16)
    w64 = (torch.randn(64, 3, 3, 8, device=dev) * 0.1).to(torch.bfloat16)
    b32, b64 = torch.zeros(32, device=dev), torch.zeros(64, device=dev)
    ms = timed(lambda: rs.resample_stem_items_gpu(items, B, 1344, 960, rs.IMAGENET_MEAN, rs.IMAGENET_STD, w32, b32))
    floor = (B * 3508 * 2480 + B * 672 * 480 * 64) / 5e12 * 1e3
    print(json.dumps({"shape": "fused stem, 25 pages -> 672x480x32", "ms": round(ms, 4), "hbm_floor_ms": round(floor, 4)}), flush=True)
    rng = np.random.default_rng(0)
    for wb, n in ((1536, 600), (2048, 375), (512, 100)):
        regs = []
        for k in range(n):
            x0, y0 = float(rng.uniform(200, 400)), float(rng.uniform(200, 3300))
            w = float(rng.uniform(0.6, 0.98)) * wb * 60.0 / 48.0
            ang = float(rng.uniform(-0.05, 0.05)) if len(sys.argv) > 1 else 0.0
            regs.append(rs.Region(pages[k % B], (x0, y0, min(x0 + w, 2470.0), y0 + 60.0), 48, int(min(w * 48 / 60, wb)), 0, ang))
        items = torch.from_numpy(rs.pack_items(regs).view(np.uint8).copy()).to(dev)
        ms = timed(lambda: rs.resample_items_gpu(items, n, 48, wb, rs.HALF_MEAN, rs.HALF_STD))
        floor = (n * 48 * wb * 16 + n * 60 * wb * 1.25) / 5e12 * 1e3
        print(json.dumps({"shape": f"{n} crops -> 48x{wb}" + (" slanted" if len(sys.argv) > 1 else ""), "ms": round(ms, 4),
                          "hbm_floor_ms": round(floor, 4)}), flush=True)
        ms = timed(lambda: rs.resample_stem_items_gpu(items, n, 48, wb, rs.HALF_MEAN, rs.HALF_STD, w64, b64))
        floor = (n * 24 * (wb // 2) * 128 + n * 60 * wb * 1.25) / 5e12 * 1e3
        print(json.dumps({"shape": f"fused stem, {n} crops -> 24x{wb // 2}x64", "ms": round(ms, 4), "hbm_floor_ms": round(floor, 4)}), flush=True)


if __name__ == "__main__":
    main()
