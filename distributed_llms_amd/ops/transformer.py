"""Transformer ops for the TrOCR-sized recognizer: LayerNorm, flash attention (MFMA),
single-query decode attention over a KV cache, embedding gather.  GPU: csrc/kernels/transformer.hip;
CPU tensors: plain PyTorch fp32 (also the oracle in tests)."""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

from . import _loader

HEAD_DIM = 64


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm(x (+ res)) over the last dim.  x/res bf16 [..., C]; gamma/beta fp32 [C]."""
    if not x.is_cuda:
        v = x.float() + (res.float() if res is not None else 0.0)
        return F.layer_norm(v, (x.shape[-1],), gamma.float(), beta.float(), eps).to(x.dtype)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    if res is not None:
        assert res.shape == x.shape and res.is_contiguous() and res.dtype == torch.bfloat16
    rc = _loader.lib().dlamd_layernorm(_loader.ptr(x), _loader.ptr(gamma), _loader.ptr(beta), _loader.ptr(res),
                                       _loader.ptr(y), rows, C, float(eps), 0, _loader.stream_ptr())
    _loader.check(rc, "dlamd_layernorm")
    return y


def attention_reference(q, k, v, H: int, scale: float, causal: bool = False) -> torch.Tensor:
    """q: [B, Sq, H*64], k/v: [B, Sk, H*64] -> [B, Sq, H*64] (fp32 maths)."""
    B, Sq, _ = q.shape
    Sk = k.shape[1]
    qf = q.float().view(B, Sq, H, HEAD_DIM).transpose(1, 2)
    kf = k.float().view(B, Sk, H, HEAD_DIM).transpose(1, 2)
    vf = v.float().view(B, Sk, H, HEAD_DIM).transpose(1, 2)
    s = (qf @ kf.transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(torch.ones(Sq, Sk, dtype=torch.bool, device=q.device).triu(1), float("-inf"))
    o = torch.softmax(s, -1) @ vf
    return o.transpose(1, 2).reshape(B, Sq, H * HEAD_DIM)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, H: int, scale: Optional[float] = None,
              causal: bool = False) -> torch.Tensor:
    """Multi-head attention over full sequences.  q/k/v may be strided views (e.g. slices of a fused QKV
    projection): only the last dim must be contiguous.  Returns a contiguous [B, Sq, H*64] bf16 tensor."""
    scale = 1.0 / math.sqrt(HEAD_DIM) if scale is None else scale
    if not q.is_cuda:
        return attention_reference(q, k, v, H, scale, causal).to(q.dtype)
    B, Sq, D = q.shape
    Sk = k.shape[1]
    assert D == H * HEAD_DIM and k.shape[2] == D and v.shape[2] == D
    assert q.dtype == k.dtype == v.dtype == torch.bfloat16
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    o = torch.empty((B, Sq, D), dtype=torch.bfloat16, device=q.device)
    rc = _loader.lib().dlamd_attention_fwd(
        _loader.ptr(q), _loader.ptr(k), _loader.ptr(v), _loader.ptr(o), B, H, Sq, Sk,
        q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), o.stride(0), o.stride(1),
        float(scale), 1 if causal else 0, _loader.stream_ptr())
    _loader.check(rc, "dlamd_attention_fwd")
    return o


def attention_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, H: int,
                     k_new: Optional[torch.Tensor] = None, v_new: Optional[torch.Tensor] = None,
                     t: Optional[torch.Tensor] = None, klen: int = 0, scale: Optional[float] = None) -> torch.Tensor:
    """One query per sequence.  q: [B, H*64].  Self-attention: pass k_new/v_new ([B, H*64], may be strided
    views) and the device step counter ``t`` (int32 [1]); the row is appended to the caches at position t and
    the query attends over [0, t].  Cross-attention: pass ``klen`` (keys already in the caches)."""
    scale = 1.0 / math.sqrt(HEAD_DIM) if scale is None else scale
    B, D = q.shape
    Tcap = k_cache.shape[1]
    if not q.is_cuda:
        if k_new is not None:
            tt = int(t.item())
            k_cache[:, tt] = k_new
            v_cache[:, tt] = v_new
            n = tt + 1
        else:
            n = klen
        return attention_reference(q.unsqueeze(1), k_cache[:, :n], v_cache[:, :n], H, scale)[:, 0].to(q.dtype)
    assert q.dtype == torch.bfloat16 and q.stride(1) == 1
    assert k_cache.stride(2) == 1 and v_cache.stride() == k_cache.stride() and k_cache.dtype == torch.bfloat16
    o = torch.empty((B, D), dtype=torch.bfloat16, device=q.device)
    n_bs = 0
    if k_new is not None:
        assert k_new.stride(1) == 1 and v_new.stride() == k_new.stride() and t is not None and t.dtype == torch.int32
        n_bs = k_new.stride(0)
    rc = _loader.lib().dlamd_attention_decode(
        _loader.ptr(q), q.stride(0), _loader.ptr(k_cache), _loader.ptr(v_cache), k_cache.stride(0), k_cache.stride(1),
        _loader.ptr(k_new), _loader.ptr(v_new), n_bs, _loader.ptr(t), int(klen), Tcap, _loader.ptr(o), o.stride(0),
        B, H, float(scale), _loader.stream_ptr())
    _loader.check(rc, "dlamd_attention_decode")
    return o


def embed_rows(ids: torch.Tensor, tok: torch.Tensor, pos: torch.Tensor, t: Optional[torch.Tensor] = None,
               t_const: int = 0, emb_scale: float = 1.0) -> torch.Tensor:
    """tok[ids] * emb_scale + pos[t].  ids int32 [B]; t: device int32 [1] step counter (or ``t_const``)."""
    if not ids.is_cuda:
        tt = int(t.item()) if t is not None else t_const
        tt = max(0, min(tt, pos.shape[0] - 1))
        return (tok[ids.long().clamp(0, tok.shape[0] - 1)].float() * emb_scale + pos[tt].float()).to(tok.dtype)
    assert ids.dtype == torch.int32 and tok.dtype == torch.bfloat16 and pos.dtype == torch.bfloat16
    assert tok.is_contiguous() and pos.is_contiguous() and tok.shape[1] == pos.shape[1]
    B, C = ids.shape[0], tok.shape[1]
    y = torch.empty((B, C), dtype=torch.bfloat16, device=ids.device)
    rc = _loader.lib().dlamd_embed_rows(_loader.ptr(ids), _loader.ptr(tok), _loader.ptr(pos), _loader.ptr(t),
                                        _loader.ptr(y), B, C, tok.shape[0], pos.shape[0], int(t_const),
                                        float(emb_scale), _loader.stream_ptr())
    _loader.check(rc, "dlamd_embed_rows")
    return y


def decode_bookkeep(tokens: torch.Tensor, probs: torch.Tensor, t: torch.Tensor, out: torch.Tensor, psum: torch.Tensor,
                    cnt: torch.Tensor, done: torch.Tensor, flag: torch.Tensor, eos: int) -> None:
    """Tail of one greedy-decode step, in place: ``out[t, b] = tokens[b]``, ``psum[b] += probs[b]``, ``cnt[b] += 1`` for
    every sequence that is not done; sequences that emitted ``eos`` become done; ``t += 1``; ``flag[0]`` is set to the
    step count the first time no sequence is live.  ``out``: int32 [max_len, B]; ``done``: uint8 [B]; ``t``, ``flag``:
    int32 [1] (``flag`` may live in pinned host memory: the host polls it without synchronising)."""
    if not tokens.is_cuda:
        tt = int(t.item())
        live = done == 0
        if tt < out.shape[0]:
            out[tt] = torch.where(live, tokens, out[tt])
            psum += probs * live
            cnt += live
        done |= (live & (tokens == eos)).to(done.dtype)
        t += 1
        if not bool((done == 0).any()) and int(flag.item()) == 0:
            flag.fill_(tt + 1)
        return
    assert tokens.dtype == torch.int32 and probs.dtype == torch.float32 and t.dtype == torch.int32
    assert out.dtype == torch.int32 and out.is_contiguous() and out.shape[1] == tokens.shape[0]
    assert psum.dtype == torch.float32 and cnt.dtype == torch.float32 and done.dtype == torch.uint8 and flag.dtype == torch.int32
    rc = _loader.lib().dlamd_decode_bookkeep(_loader.ptr(tokens), _loader.ptr(probs), _loader.ptr(t), _loader.ptr(out),
                                             _loader.ptr(psum), _loader.ptr(cnt), _loader.ptr(done), flag.data_ptr(),
                                             tokens.shape[0], out.shape[0], int(eos), _loader.stream_ptr())
    _loader.check(rc, "dlamd_decode_bookkeep")
