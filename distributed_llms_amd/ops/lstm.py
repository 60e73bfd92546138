"""(Bi)LSTM layer for the CRNN recognizer.

Split exactly the way the hardware wants it:
  * input projection for all time steps = one big MFMA GEMM (``linear_bf16``)
  * recurrence = persistent ``dlamd_lstm_bidir`` kernel (csrc/kernels/lstm.hip)

Weight preparation (``pack_lstm_weights``) converts PyTorch's ``nn.LSTM``
layout (rows = gate*H + unit, gates i,f,g,o) into the kernel's permuted gate
order and MFMA fragment order once, at model-load time.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch

from . import _loader
from .conv import linear_bf16


def lstm_gate_permutation(H: int) -> torch.Tensor:
    """perm[r] = source row (gate*H + unit) of permuted row r = tile*16 + fq*4 + gate."""
    r = torch.arange(4 * H)
    tile, rr = r // 16, r % 16
    fq, gate = rr // 4, rr % 4
    unit = (tile // 4) * 16 + fq * 4 + tile % 4
    return gate * H + unit


@dataclass
class PackedLSTM:
    H: int
    ndir: int
    w_ih: torch.Tensor     # [ndir*4H, I] bf16, permuted rows
    bias: torch.Tensor     # [ndir*4H] f32 (b_ih + b_hh), permuted
    w_frag: torch.Tensor   # [ndir, 4H/16, H/32, 64, 8] bf16 fragment order
    # originals (fp32) for the CPU / oracle path
    w_ih_ref: Sequence[torch.Tensor]
    w_hh_ref: Sequence[torch.Tensor]
    b_ref: Sequence[torch.Tensor]

    def to(self, device) -> "PackedLSTM":
        return PackedLSTM(
            self.H, self.ndir, self.w_ih.to(device), self.bias.to(device), self.w_frag.to(device),
            [w.to(device) for w in self.w_ih_ref], [w.to(device) for w in self.w_hh_ref],
            [b.to(device) for b in self.b_ref],
        )


def pack_lstm_weights(
    w_ih: Sequence[torch.Tensor], w_hh: Sequence[torch.Tensor], b_ih: Sequence[torch.Tensor],
    b_hh: Sequence[torch.Tensor],
) -> PackedLSTM:
    """One entry per direction (forward[, reverse]); tensors in nn.LSTM layout."""
    ndir = len(w_ih)
    H = w_hh[0].shape[1]
    assert H % 64 == 0, "kernel supports H in {64,128,256}"
    perm = lstm_gate_permutation(H)
    wi, bs, wf = [], [], []
    for d in range(ndir):
        wi.append(w_ih[d].float()[perm])
        bs.append((b_ih[d].float() + b_hh[d].float())[perm])
        wp = w_hh[d].float()[perm]  # [4H, H]
        f = wp.view(4 * H // 16, 16, H // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(4 * H // 16, H // 32, 64, 8)
        wf.append(f)
    return PackedLSTM(
        H=H, ndir=ndir,
        w_ih=torch.cat(wi, 0).to(torch.bfloat16).contiguous(),
        bias=torch.cat(bs, 0).float().contiguous(),
        w_frag=torch.stack(wf, 0).to(torch.bfloat16).contiguous(),
        w_ih_ref=[w.float() for w in w_ih], w_hh_ref=[w.float() for w in w_hh],
        b_ref=[(a.float() + b.float()) for a, b in zip(b_ih, b_hh)],
    )


def lstm_reference(x: torch.Tensor, p: PackedLSTM, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 oracle with packed-sequence semantics (reverse pass starts at len-1; padding -> 0)."""
    N, T, _ = x.shape
    H = p.H
    xf = x.float()
    y = torch.zeros(N, T, p.ndir * H, dtype=torch.float32, device=x.device)
    L = torch.full((N,), T, dtype=torch.long, device=x.device) if lens is None else lens.long().clamp(max=T)
    for d in range(p.ndir):
        wih, whh, b = p.w_ih_ref[d].to(x.device), p.w_hh_ref[d].to(x.device), p.b_ref[d].to(x.device)
        h = torch.zeros(N, H, device=x.device)
        c = torch.zeros(N, H, device=x.device)
        for step in range(T):
            act = step < L
            t = torch.where(act, torch.full_like(L, step) if d == 0 else (L - 1 - step), torch.zeros_like(L))
            xt = xf[torch.arange(N, device=x.device), t]
            g = xt @ wih.t() + h @ whh.t() + b
            i, f, gg, o = g.chunk(4, dim=1)
            c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h_new = torch.sigmoid(o) * torch.tanh(c_new)
            m = act[:, None]
            c = torch.where(m, c_new, c)
            h = torch.where(m, h_new, h)
            idx = torch.nonzero(act).flatten()
            y[idx, t[idx], d * H : (d + 1) * H] = h_new[idx]
    return y


def lstm_layer(x: torch.Tensor, p: PackedLSTM, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x: [N, T, I] bf16 -> [N, T, ndir*H] bf16."""
    if not x.is_cuda:
        return lstm_reference(x, p, lens).to(x.dtype)
    N, T, _ = x.shape
    xproj = linear_bf16(x.contiguous(), p.w_ih, p.bias)  # [N, T, ndir*4H], permuted gate columns
    y = torch.empty((N, T, p.ndir * p.H), dtype=torch.bfloat16, device=x.device)
    if lens is not None:
        assert lens.dtype == torch.int32 and lens.is_cuda
    rc = _loader.lib().dlamd_lstm_bidir(
        _loader.ptr(xproj), _loader.ptr(p.w_frag), _loader.ptr(lens), None, None, None, _loader.ptr(y), N, T, p.H,
        p.ndir, _loader.stream_ptr(),
    )
    _loader.check(rc, "dlamd_lstm_bidir")
    return y


@dataclass
class RaggedLayout:
    """Sequences of different allocated lengths packed back to back as rows of one [R, C] matrix.

    ``row_off[n]`` = first row of sequence n, ``tcap[n]`` = rows allocated to it, ``lens[n]`` = valid
    steps (<= tcap), ``perm`` = slot -> sequence order used to schedule workgroups (longest first, so
    the 16 sequences sharing a workgroup have similar lengths)."""

    row_off: torch.Tensor   # int64 [N]
    tcap: torch.Tensor      # int32 [N]
    lens: torch.Tensor      # int32 [N]
    perm: torch.Tensor      # int32 [N]
    rows: int
    tmax: int

    @property
    def n(self) -> int:
        return int(self.lens.shape[0])


def lstm_layer_ragged(x: torch.Tensor, p: PackedLSTM, lay: RaggedLayout) -> torch.Tensor:
    """x: [R, I] bf16 packed rows -> [R, ndir*H] bf16; ONE recurrent launch for every width bucket."""
    if not x.is_cuda:
        out = torch.zeros(x.shape[0], p.ndir * p.H, dtype=x.dtype)
        ro, tc, ln = lay.row_off.tolist(), lay.tcap.tolist(), lay.lens.tolist()
        for n in range(lay.n):
            seg = x[ro[n] : ro[n] + tc[n]].unsqueeze(0)
            y = lstm_reference(seg, p, torch.tensor([ln[n]]))
            out[ro[n] : ro[n] + tc[n]] = y[0].to(x.dtype)
        return out
    assert x.dim() == 2 and x.is_contiguous()
    xproj = linear_bf16(x, p.w_ih, p.bias)
    y = torch.empty((x.shape[0], p.ndir * p.H), dtype=torch.bfloat16, device=x.device)
    assert lay.row_off.dtype == torch.int64 and lay.tcap.dtype == torch.int32 and lay.lens.dtype == torch.int32
    assert lay.perm.dtype == torch.int32 and lay.row_off.is_cuda
    rc = _loader.lib().dlamd_lstm_bidir(
        _loader.ptr(xproj), _loader.ptr(p.w_frag), _loader.ptr(lay.lens), _loader.ptr(lay.row_off),
        _loader.ptr(lay.tcap), _loader.ptr(lay.perm), _loader.ptr(y), lay.n, lay.tmax, p.H, p.ndir,
        _loader.stream_ptr(),
    )
    _loader.check(rc, "dlamd_lstm_bidir")
    return y


def set_kernel_variant(variant: int) -> None:
    """Tests / benchmarks: force a recurrence kernel (0 automatic, 1 the generic 4-wave kernel, 5 v4 with 7 / 8 of W_hh resident)."""
    _loader.lib().dlamd_lstm_set_variant(int(variant))
