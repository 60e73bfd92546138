"""Model-level microbenchmarks on one MI355X: DBNet and CRNN inference at production shapes.

    python bench/kernel_bench.py [--det-batch 16] [--rec-batch 1024] [--rec-width 640] [--iters 10]

Prints ms / step, achieved TFLOP/s (dense conv+GEMM FLOPs only) and per-layer times.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from distributed_llms_amd.models.crnn import CRNN  # noqa: E402
from distributed_llms_amd.models.dbnet import DBNet  # noqa: E402
from distributed_llms_amd.ops import conv as C  # noqa: E402


def timed(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def layer_table(cases, iters):
    rows = []
    for name, x, w, b, kw in cases:
        ms = timed(lambda: C.conv2d_nhwc(x, w, b, **kw), iters)
        B, H, W_, Cin = x.shape
        Cout, kh, kw_, _ = w.shape
        sh, sw = C._pair(kw.get("stride", 1))
        ph, pw = C._pair(kw.get("padding", 0))
        Ho, Wo = C.out_hw(H, W_, kh, kw_, sh, sw, ph, pw)
        fl = 2.0 * B * Ho * Wo * Cout * kh * kw_ * Cin
        by = 2.0 * (x.numel() + B * Ho * Wo * Cout)
        rows.append({"layer": name, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1), "gbps": round(by / ms / 1e6, 0),
                     "M": B * Ho * Wo, "N": Cout, "K": kh * kw_ * Cin})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--det-batch", type=int, default=16)
    ap.add_argument("--det-h", type=int, default=1344)
    ap.add_argument("--det-w", type=int, default=960)
    ap.add_argument("--rec-batch", type=int, default=1024)
    ap.add_argument("--rec-width", type=int, default=640)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--layers", action="store_true")
    ap.add_argument("--sweep", action="store_true", help="time every tile config on every layer (DBNet + CRNN shapes)")
    args = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    out = {}

    det = DBNet().to(dev).prepare()
    x = torch.randn(args.det_batch, args.det_h, args.det_w, 8, device=dev).to(torch.bfloat16)
    x[..., 3:] = 0
    ms = timed(lambda: det.infer(x), args.iters)
    fl = det.flops_per_page(args.det_h, args.det_w) * args.det_batch
    out["dbnet"] = {"batch": args.det_batch, "ms": round(ms, 3), "ms_per_page": round(ms / args.det_batch, 4),
                    "tflops": round(fl / ms / 1e9, 1), "gflop_per_page": round(fl / args.det_batch / 1e9, 1)}

    rec = CRNN().to(dev).prepare()
    xr = torch.randn(args.rec_batch, 48, args.rec_width, 8, device=dev).to(torch.bfloat16)
    xr[..., 3:] = 0
    lens = torch.full((args.rec_batch,), args.rec_width // 4, dtype=torch.int32, device=dev)
    ms = timed(lambda: rec.infer(xr, lens), args.iters)
    fl = rec.flops_per_crop(args.rec_width) * args.rec_batch
    out["crnn"] = {"batch": args.rec_batch, "width": args.rec_width, "ms": round(ms, 3),
                   "us_per_crop": round(ms / args.rec_batch * 1e3, 2), "tflops": round(fl / ms / 1e9, 1)}
    print(json.dumps(out))

    if args.layers:
        B, H, W = args.det_batch, args.det_h, args.det_w

        def mk(cin, cout, k, h, w, **kw):
            xx = torch.randn(B, h, w, cin, device=dev).to(torch.bfloat16)
            ww = (torch.randn(cout, k, k, cin, device=dev) * 0.05).to(torch.bfloat16)
            bb = torch.zeros(cout, device=dev)
            return xx, ww, bb, dict(padding=k // 2, act="relu", **kw)

        cases = [
            ("stem1 8->32 s2 @1/1", *mk(8, 32, 3, H, W, stride=2)),
            ("stem2 32->32 @1/2", *mk(32, 32, 3, H // 2, W // 2)),
            ("stem3 32->64 s2 @1/2", *mk(32, 64, 3, H // 2, W // 2, stride=2)),
            ("l1 64->64 @1/4", *mk(64, 64, 3, H // 4, W // 4)),
            ("l2 128->128 @1/8", *mk(128, 128, 3, H // 8, W // 8)),
            ("l3 256->256 @1/16", *mk(256, 256, 3, H // 16, W // 16)),
            ("l4 512->512 @1/32", *mk(512, 512, 3, H // 32, W // 32)),
            ("in2 1x1 64->256 @1/4", *mk(64, 256, 1, H // 4, W // 4)),
            ("p2 256->64 @1/4", *mk(256, 64, 3, H // 4, W // 4)),
            ("head2 1x1 64->256 @1/4", *mk(64, 256, 1, H // 4, W // 4)),
        ]
        for r in layer_table(cases, args.iters):
            print(json.dumps(r))

    if args.sweep:
        B, H, W = args.det_batch, args.det_h, args.det_w

        def mk2(bb, cin, cout, kh, kw, h, w, stride=1, pad=None):
            xx = torch.randn(bb, h, w, cin, device=dev).to(torch.bfloat16)
            ww = (torch.randn(cout, kh, kw, cin, device=dev) * 0.05).to(torch.bfloat16)
            b_ = torch.zeros(cout, device=dev)
            return xx, ww, b_, dict(padding=(kh // 2, kw // 2) if pad is None else pad, act="relu", stride=stride)

        NB, RW = 100, 2048
        T = RW // 4
        cases = [
            ("det stem1 8->32 s2", *mk2(B, 8, 32, 3, 3, H, W, 2)),
            ("det stem2 32->32", *mk2(B, 32, 32, 3, 3, H // 2, W // 2)),
            ("det stem3 32->64 s2", *mk2(B, 32, 64, 3, 3, H // 2, W // 2, 2)),
            ("det l1 64->64", *mk2(B, 64, 64, 3, 3, H // 4, W // 4)),
            ("det l2a 64->128 s2", *mk2(B, 64, 128, 3, 3, H // 4, W // 4, 2)),
            ("det l2 128->128", *mk2(B, 128, 128, 3, 3, H // 8, W // 8)),
            ("det l3 256->256", *mk2(B, 256, 256, 3, 3, H // 16, W // 16)),
            ("det l4 512->512", *mk2(B, 512, 512, 3, 3, H // 32, W // 32)),
            ("det in2 1x1 64->256", *mk2(B, 64, 256, 1, 1, H // 4, W // 4)),
            ("det in5 1x1 512->256", *mk2(B, 512, 256, 1, 1, H // 32, W // 32)),
            ("det p2 256->64", *mk2(B, 256, 64, 3, 3, H // 4, W // 4)),
            ("det head1 256->64", *mk2(B, 256, 64, 3, 3, H // 4, W // 4)),
            ("det head2 1x1 64->256", *mk2(B, 64, 256, 1, 1, H // 4, W // 4)),
            ("rec c1 8->64 s2", *mk2(NB, 8, 64, 3, 3, 48, RW, 2)),
            ("rec c2 64->128 s2", *mk2(NB, 64, 128, 3, 3, 24, RW // 2, 2)),
            ("rec c3 128->256", *mk2(NB, 128, 256, 3, 3, 12, T)),
            ("rec c4 256->256 s(2,1)", *mk2(NB, 256, 256, 3, 3, 12, T, (2, 1))),
            ("rec c5 256->512 s(2,1)", *mk2(NB, 256, 512, 3, 3, 6, T, (2, 1))),
            ("rec c6 512->512 (3,1)", *mk2(NB, 512, 512, 3, 1, 3, T, 1, (0, 0))),
            ("rec proj 512->2048", *mk2(1, 512, 2048, 1, 1, 1, 10 * NB * T // 2)),
            ("rec proj2 512->2048 small", *mk2(1, 512, 2048, 1, 1, 1, NB * T)),
            ("rec fc 512->96", *mk2(1, 512, 96, 1, 1, 1, 10 * NB * T // 2)),
        ]
        for name, x_, w_, b_, kw in cases:
            res = {}
            for cfg in (0, 32, 64, 128, 1064, 1128, 2032, 3064, 3128, 4128, 8128):
                if cfg in (32,) and w_.shape[0] > 64:
                    continue
                patch_ok = (w_.shape[1] == 3 and w_.shape[2] == 3 and C._pair(kw['stride'])[1] == 1 and x_.shape[3] % 64 == 0)
                if 3000 <= cfg < 7000 and not patch_ok:
                    continue
                if cfg >= 7000 and x_.shape[3] % 64 != 0:
                    continue
                res[cfg] = round(timed(lambda: C.conv2d_nhwc(x_, w_, b_, cfg=cfg, **kw), max(3, args.iters // 2), 2), 4)
            Bx, Hx, Wx, Cin = x_.shape
            Cout, kh, kw_, _ = w_.shape
            sh, sw = C._pair(kw["stride"])
            ph, pw = C._pair(kw["padding"])
            Ho, Wo = C.out_hw(Hx, Wx, kh, kw_, sh, sw, ph, pw)
            fl = 2.0 * Bx * Ho * Wo * Cout * kh * kw_ * Cin
            best = min((v, k) for k, v in res.items() if k != 0)
            print(json.dumps({"layer": name, "M": Bx * Ho * Wo, "N": Cout, "K": kh * kw_ * Cin, "ms_by_cfg": res,
                              "auto_tflops": round(fl / res[0] / 1e9, 1), "best_cfg": best[1],
                              "best_tflops": round(fl / best[0] / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
