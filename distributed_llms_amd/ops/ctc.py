"""CTC greedy decode + row argmax.

GPU: ``dlamd_row_argmax`` + ``dlamd_ctc_collapse`` (csrc/kernels/ctc_decode.hip).
CPU / oracle: pure PyTorch.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import _loader


def row_argmax_reference(logits: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    lf = logits.float()
    p = torch.softmax(lf, dim=-1)
    prob, idx = p.max(dim=-1)
    idx = lf.argmax(dim=-1)
    return idx.to(torch.int32), prob


def row_argmax(logits: torch.Tensor, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
               ) -> Tuple[torch.Tensor, torch.Tensor]:
    """argmax and max softmax probability over the last dim. logits: [..., V] bf16/f32.  ``out`` = (int32 indices,
    float32 probabilities) of the leading shape: written in place (static buffers of a captured decode step)."""
    if not logits.is_cuda:
        idx, prob = row_argmax_reference(logits)
        if out is not None:
            out[0].copy_(idx)
            out[1].copy_(prob)
            return out
        return idx, prob
    assert logits.dtype in (torch.bfloat16, torch.float32)
    assert logits.stride(-1) == 1
    V = logits.shape[-1]
    lead = logits.shape[:-1]
    l2 = logits.reshape(-1, V)
    R = l2.shape[0]
    if out is not None:
        idx, prob = out
        assert idx.dtype == torch.int32 and prob.dtype == torch.float32 and idx.numel() == R and prob.numel() == R
        assert idx.is_contiguous() and prob.is_contiguous()
    else:
        idx = torch.empty((R,), dtype=torch.int32, device=logits.device)
        prob = torch.empty((R,), dtype=torch.float32, device=logits.device)
    rc = _loader.lib().dlamd_row_argmax(
        _loader.ptr(l2), 1 if logits.dtype == torch.float32 else 0, _loader.ptr(idx), _loader.ptr(prob),
        R, V, l2.stride(0), _loader.stream_ptr(),
    )
    _loader.check(rc, "dlamd_row_argmax")
    return idx.reshape(lead), prob.reshape(lead)


def ctc_greedy_reference(
    logits: torch.Tensor, lens: Optional[torch.Tensor] = None, blank: int = 0
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    idx, prob = row_argmax_reference(logits)
    N, T = idx.shape
    out = torch.full((N, T), -1, dtype=torch.int32)
    out_len = torch.zeros((N,), dtype=torch.int32)
    conf = torch.zeros((N,), dtype=torch.float32)
    idx_c, prob_c = idx.cpu(), prob.cpu()
    for n in range(N):
        L = T if lens is None else min(int(lens[n]), T)
        prev, k, ps = None, 0, 0.0
        for t in range(L):
            v = int(idx_c[n, t])
            if v != blank and v != prev:
                out[n, k] = v
                ps += float(prob_c[n, t])
                k += 1
            prev = v
        out_len[n] = k
        conf[n] = ps / k if k > 0 else 0.0
    return out, out_len, conf


def ctc_greedy_decode(
    logits: torch.Tensor, lens: Optional[torch.Tensor] = None, blank: int = 0
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """logits [N, T, V] -> (ids [N, T] padded with -1, lengths [N], mean confidence [N])."""
    if not logits.is_cuda:
        return ctc_greedy_reference(logits, lens, blank)
    N, T, V = logits.shape
    idx, prob = row_argmax(logits)
    out = torch.empty((N, T), dtype=torch.int32, device=logits.device)
    out_len = torch.empty((N,), dtype=torch.int32, device=logits.device)
    conf = torch.empty((N,), dtype=torch.float32, device=logits.device)
    if lens is not None:
        assert lens.dtype == torch.int32 and lens.is_cuda
    rc = _loader.lib().dlamd_ctc_collapse(
        _loader.ptr(idx), _loader.ptr(prob), _loader.ptr(lens), None, _loader.ptr(out), _loader.ptr(out_len),
        _loader.ptr(conf), N, T, blank, _loader.stream_ptr(),
    )
    _loader.check(rc, "dlamd_ctc_collapse")
    return out, out_len, conf


def ctc_greedy_decode_ragged(logits: torch.Tensor, lens: torch.Tensor, row_off: torch.Tensor, tmax: int,
                             blank: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """logits: [R, V] packed rows; sequence n occupies rows row_off[n] .. row_off[n] + lens[n].
    -> (ids [N, tmax] padded -1, lengths [N], confidence [N])."""
    N = lens.shape[0]
    if not logits.is_cuda:
        out = torch.full((N, tmax), -1, dtype=torch.int32)
        out_len = torch.zeros((N,), dtype=torch.int32)
        conf = torch.zeros((N,), dtype=torch.float32)
        for n in range(N):
            r0, L = int(row_off[n]), int(lens[n])
            o, ol, c = ctc_greedy_reference(logits[r0 : r0 + L].unsqueeze(0), None, blank)
            out[n, :L] = o[0]
            out_len[n], conf[n] = ol[0], c[0]
        return out, out_len, conf
    idx, prob = row_argmax(logits)
    out = torch.empty((N, tmax), dtype=torch.int32, device=logits.device)
    out_len = torch.empty((N,), dtype=torch.int32, device=logits.device)
    conf = torch.empty((N,), dtype=torch.float32, device=logits.device)
    assert lens.dtype == torch.int32 and row_off.dtype == torch.int64 and lens.is_cuda and row_off.is_cuda
    rc = _loader.lib().dlamd_ctc_collapse(
        _loader.ptr(idx), _loader.ptr(prob), _loader.ptr(lens), _loader.ptr(row_off), _loader.ptr(out),
        _loader.ptr(out_len), _loader.ptr(conf), N, tmax, blank, _loader.stream_ptr(),
    )
    _loader.check(rc, "dlamd_ctc_collapse")
    return out, out_len, conf


def ids_to_strings(ids: torch.Tensor, lens: torch.Tensor, charset: str) -> List[str]:
    """Vocabulary index 0 is the CTC blank; index i >= 1 is ``charset[i-1]``."""
    ids_c = ids.cpu().tolist()
    lens_c = lens.cpu().tolist()
    out = []
    n_chars = len(charset)
    for row, L in zip(ids_c, lens_c):
        out.append("".join(charset[v - 1] for v in row[:L] if 1 <= v <= n_chars))
    return out
