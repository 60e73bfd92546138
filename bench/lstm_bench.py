#!/usr/bin/env python3
"""Recurrent-kernel microbenchmark: us per time step for the persistent BiLSTM variants.

    python bench/lstm_bench.py [N] [T]        # every variant: 0 automatic (v4, 6/8 of W_hh resident), 1 generic 4-wave kernel, 5 v4 with 7/8 resident
"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from distributed_llms_amd.ops import lstm


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 970
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    H, I = 256, 512
    g = torch.Generator().manual_seed(0)
    w_ih = [torch.randn(4 * H, I, generator=g) * 0.05 for _ in range(2)]
    w_hh = [torch.randn(4 * H, H, generator=g) * 0.05 for _ in range(2)]
    b = [torch.zeros(4 * H) for _ in range(4)]
    p = lstm.pack_lstm_weights(w_ih, w_hh, b[:2], b[2:]).to("cuda")
    x = (torch.randn(N, T, I, generator=g) * 0.5).to(torch.bfloat16).cuda()
    lens = torch.full((N,), T, dtype=torch.int32, device="cuda")
    xproj = lstm.linear_bf16(x, p.w_ih, p.bias)
    y = torch.empty((N, T, 2 * H), dtype=torch.bfloat16, device="cuda")
    from distributed_llms_amd.ops import _loader

    def run():
        rc = _loader.lib().dlamd_lstm_bidir(_loader.ptr(xproj), _loader.ptr(p.w_frag), _loader.ptr(lens), None, None,
                                            None, _loader.ptr(y), N, T, H, 2, _loader.stream_ptr())
        assert rc == 0

    for variant in (0, 1, 5):
        lstm.set_kernel_variant(variant)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"variant={variant} N={N} T={T}: {ms:.3f} ms/launch, {ms * 1e3 / T:.2f} us/step", flush=True)
    lstm.set_kernel_variant(0)


if __name__ == "__main__":
    main()
