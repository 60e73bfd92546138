"""Row-wise int32 prefix sums (``csrc/kernels/scan.hip``): every row is spread over ``len / 1024`` workgroups.

The JPEG decoder needs two per batch of pages (block counts per 128-byte subsequence, DC differences per block);
``torch.cumsum`` walks each row with few workgroups and took 0.27 ms per call on 36 rows x 136 k elements.
"""
from __future__ import annotations

import torch

from . import _loader

CHUNK = 1024


def row_cumsum_i32(x: torch.Tensor, out: torch.Tensor, exclusive: bool = False) -> torch.Tensor:
    """``out[r, i] = sum(x[r, : i + 1])`` (or ``sum(x[r, :i])`` when ``exclusive``).  x, out: [rows, len] int32,
    contiguous, distinct tensors."""
    assert x.dtype == torch.int32 and out.dtype == torch.int32 and x.dim() == 2 and x.shape == out.shape
    assert x.is_contiguous() and out.is_contiguous()
    if not x.is_cuda:
        torch.cumsum(x, dim=1, dtype=torch.int32, out=out)
        if exclusive:
            out.sub_(x)
        return out
    rows, n = x.shape
    if rows == 0 or n == 0:
        return out
    assert x.data_ptr() != out.data_ptr()
    scratch = torch.empty((rows * ((n + CHUNK - 1) // CHUNK),), dtype=torch.int32, device=x.device)
    rc = _loader.lib().dlamd_row_cumsum_i32(_loader.ptr(x), _loader.ptr(out), _loader.ptr(scratch), scratch.numel(), rows, n,
                                            1 if exclusive else 0, _loader.stream_ptr())
    _loader.check(rc, "dlamd_row_cumsum_i32")
    return out
