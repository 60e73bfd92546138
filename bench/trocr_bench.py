#!/usr/bin/env python3
"""ViT-encoder + Transformer-decoder recognizer (TrOCR-base-sized), bf16, one MI355X.

Reports line crops/sec for encode + greedy decode of ``--tokens`` tokens per crop (hipGraph-replayed steps),
the encoder-only rate, and the equivalent pages/sec at ``--lines-per-page`` lines."""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from distributed_llms_amd.models.trocr import TrOCR, TrOCRConfig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--tokens", type=int, default=32)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--lines-per-page", type=float, default=40.0)
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)
    cfg = TrOCRConfig()
    m = TrOCR(cfg).cuda().prepare()
    x = torch.randn(args.batch, cfg.image_size, cfg.image_size, 8, device="cuda").to(torch.bfloat16)
    x[..., 3:] = 0
    m.generate(x, args.tokens, use_graph=not args.no_graph)   # warm-up + graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        m.encode(x)
    torch.cuda.synchronize()
    t_enc = (time.perf_counter() - t0) / args.iters
    t0 = time.perf_counter()
    for _ in range(args.iters):
        m.generate(x, args.tokens, use_graph=not args.no_graph)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / args.iters
    fl = m.flops_per_crop(args.tokens) * args.batch
    print(json.dumps({
        "metric": "TrOCR-base-sized recognizer line crops/sec (encode + greedy decode)",
        "value": round(args.batch / t_all, 1), "unit": "crops/s",
        "pages_per_sec_equiv": round(args.batch / t_all / args.lines_per_page, 2),
        "encode_crops_per_sec": round(args.batch / t_enc, 1), "encode_ms": round(t_enc * 1e3, 2),
        "decode_ms_per_token": round((t_all - t_enc) * 1e3 / args.tokens, 3), "tflops": round(fl / t_all / 1e12, 1),
        "params_M": round(m.num_params() / 1e6, 1), "batch": args.batch, "decode_tokens": args.tokens,
        "hip_graph": not args.no_graph, "dtype": "bf16",
    }), flush=True)


if __name__ == "__main__":
    main()
