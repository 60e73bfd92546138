"""ViT-encoder + Transformer-decoder line recognizer, TrOCR-base-sized.

Sizes (defaults): 384 x 384 crop -> 24 x 24 patches of 16 x 16 + CLS = 577 tokens, encoder 12 x
(d = 768, 12 heads, MLP 3072); decoder 12 x (d = 1024, 16 heads, FFN 4096, cross-attention over the
577 encoder tokens), vocabulary 50 265, ~330 M parameters.  Both stacks are pre-LN; weights are
random-init (no checkpoint is reachable offline), so this matches TrOCR-base in *shape and cost*,
not in its exact normalisation placement.

MI355X inference path (``encode`` / ``generate``):
* patch embedding = the MFMA implicit-GEMM conv (16 x 16 stride 16, K = 2048);
* every linear = the same kernel's 1x1 path with bias / GELU / residual fused in the epilogue;
  Q, K, V are one fused [3d, d] GEMM, cross-attention K/V for all 577 memory tokens one [2d, d_enc] GEMM
  per layer, computed once per batch;
* encoder attention = flash-style MFMA kernel streaming K/V through LDS in 64-key blocks;
* one greedy decode step (embed -> 12 layers -> LM-head GEMM -> argmax -> token history / EOS marks / step
  counter) has fully static launch parameters: the step index lives in device memory and the decode-attention
  kernel appends to the KV cache itself, so the whole step (~140 launches) is captured ONCE per batch bucket in
  a hipGraph and replayed per generated token; the host loop around the replays launches nothing else and learns
  that every sequence has finished from a flag in pinned host memory, read without synchronising the stream.
  Decode is GEMV-shaped at small batch, so crops are batched (>= 256);
* the encoder runs in chunks of ``ENC_CHUNK`` crops: its activations then stay inside the 256 MB last-level cache
  (a 1024-crop pass measured 4490 crops/s in one piece against 6080 at 256).

``forward_reference`` is the plain PyTorch fp32 implementation over the same parameters (test oracle).
In the reference service this role is played by magic-pdf's UniMERNet / OCR recognizers on CPU
(services/ocr_service/download_models_hf.py:40-41).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops import ctc as ctc_ops
from ..ops import transformer as T
from ..ops.conv import conv2d_nhwc, linear_bf16


@dataclass
class TrOCRConfig:
    image_size: int = 384
    patch: int = 16
    enc_dim: int = 768
    enc_layers: int = 12
    enc_heads: int = 12
    enc_mlp: int = 3072
    dec_dim: int = 1024
    dec_layers: int = 12
    dec_heads: int = 16
    dec_ffn: int = 4096
    vocab: int = 50265
    max_len: int = 128
    bos: int = 0
    pad: int = 1
    eos: int = 2
    # decoder variant.  Defaults = this repository's own layout (pre-LN, final LN).  The published TrOCR decoder
    # (transformers' TrOCRDecoder) is post-LN with a LayerNorm on the embeddings, learned positions shifted by two and
    # no final LN: ``hf_trocr_config`` sets these, ``tools/download_models.convert_hf_trocr`` maps its tensors.
    post_ln: bool = False
    embed_ln: bool = False
    pos_offset: int = 0
    final_ln: bool = True
    embed_scale: float = 1.0
    act: str = "gelu"           # "gelu" | "relu" (decoder feed-forward)
    enc_ln_eps: float = 1e-5    # LayerNorm epsilon of the encoder (transformers' ViT: 1e-12)

    @property
    def tokens(self) -> int:
        return (self.image_size // self.patch) ** 2 + 1

    @property
    def vocab_padded(self) -> int:
        return (self.vocab + 15) // 16 * 16


# Formula recogniser (the reference's MFR model, services/ocr_service/download_models_hf.py:40, is a UniMERNet
# encoder-decoder): the same architecture at a size that trains in minutes.  Formula crops are letter-boxed into a
# 96 x 384 canvas = 6 x 24 patches, the same 144 patch tokens as a 192 x 192 square.
FORMULA_HW = (96, 384)


def hf_trocr_config(image_size: int = 384, patch: int = 16, enc_dim: int = 768, enc_layers: int = 12, enc_heads: int = 12,
                    enc_mlp: int = 3072, dec_dim: int = 1024, dec_layers: int = 12, dec_heads: int = 16, dec_ffn: int = 4096,
                    vocab: int = 50265, max_len: int = 512, act: str = "gelu", scale_embedding: bool = False,
                    layernorm_embedding: bool = True) -> TrOCRConfig:
    """The layout of the published TrOCR checkpoints (defaults: microsoft/trocr-base-*)."""
    return TrOCRConfig(image_size=image_size, patch=patch, enc_dim=enc_dim, enc_layers=enc_layers, enc_heads=enc_heads,
                       enc_mlp=enc_mlp, dec_dim=dec_dim, dec_layers=dec_layers, dec_heads=dec_heads, dec_ffn=dec_ffn,
                       vocab=vocab, max_len=max_len, bos=0, pad=1, eos=2, post_ln=True, embed_ln=layernorm_embedding,
                       pos_offset=2, final_ln=False, embed_scale=(dec_dim ** 0.5 if scale_embedding else 1.0), act=act,
                       enc_ln_eps=1e-12)


def formula_config() -> TrOCRConfig:
    return TrOCRConfig(image_size=192, enc_dim=256, enc_layers=4, enc_heads=4, enc_mlp=1024, dec_dim=256, dec_layers=4,
                       dec_heads=4, dec_ffn=1024, vocab=272, max_len=128)


class _EncLayer(nn.Module):
    def __init__(self, d: int, mlp: int, eps: float = 1e-5):
        super().__init__()
        self.ln1, self.ln2 = nn.LayerNorm(d, eps=eps), nn.LayerNorm(d, eps=eps)
        self.qkv, self.proj = nn.Linear(d, 3 * d), nn.Linear(d, d)
        self.fc1, self.fc2 = nn.Linear(d, mlp), nn.Linear(mlp, d)


class _DecLayer(nn.Module):
    def __init__(self, d: int, d_enc: int, ffn: int):
        super().__init__()
        self.ln1, self.ln2, self.ln3 = nn.LayerNorm(d), nn.LayerNorm(d), nn.LayerNorm(d)
        self.qkv, self.proj = nn.Linear(d, 3 * d), nn.Linear(d, d)
        self.xq, self.xkv, self.xproj = nn.Linear(d, d), nn.Linear(d_enc, 2 * d), nn.Linear(d, d)
        self.fc1, self.fc2 = nn.Linear(d, ffn), nn.Linear(ffn, d)


def _pk(lin: nn.Linear):
    return lin.weight.detach().to(torch.bfloat16).contiguous(), lin.bias.detach().float().contiguous()


def _ln(ln: nn.LayerNorm):
    return ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous()


class TrOCR(nn.Module):
    def __init__(self, cfg: Optional[TrOCRConfig] = None):
        super().__init__()
        self.cfg = cfg = cfg or TrOCRConfig()
        assert cfg.enc_dim == 64 * cfg.enc_heads and cfg.dec_dim == 64 * cfg.dec_heads, "head dim is 64"
        self.patch_embed = nn.Conv2d(3, cfg.enc_dim, cfg.patch, cfg.patch)
        self.cls = nn.Parameter(torch.zeros(1, 1, cfg.enc_dim))
        self.enc_pos = nn.Parameter(torch.randn(1, cfg.tokens, cfg.enc_dim) * 0.02)
        self.enc = nn.ModuleList([_EncLayer(cfg.enc_dim, cfg.enc_mlp, cfg.enc_ln_eps) for _ in range(cfg.enc_layers)])
        self.enc_ln = nn.LayerNorm(cfg.enc_dim, eps=cfg.enc_ln_eps)
        self.tok = nn.Embedding(cfg.vocab, cfg.dec_dim)
        self.pos = nn.Embedding(cfg.max_len + cfg.pos_offset, cfg.dec_dim)
        self.emb_ln = nn.LayerNorm(cfg.dec_dim) if cfg.embed_ln else None
        self.dec = nn.ModuleList([_DecLayer(cfg.dec_dim, cfg.enc_dim, cfg.dec_ffn) for _ in range(cfg.dec_layers)])
        self.dec_ln = nn.LayerNorm(cfg.dec_dim) if cfg.final_ln else None
        self.lm_head = nn.Linear(cfg.dec_dim, cfg.vocab, bias=False)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, std=0.02)
        self._pk: Optional[Dict] = None
        self._graphs: Dict[int, "_DecodeGraph"] = {}

    def _apply(self, fn, *a, **k):
        self._pk = None
        self._graphs = {}
        return super()._apply(fn, *a, **k)

    def num_params(self) -> int:
        return sum(p.numel() for p in self.parameters())

    # ------------------------------------------------------------------ PyTorch fp32 reference
    def encode_reference(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B, 3, S, S] normalised -> memory [B, tokens, enc_dim]."""
        cfg = self.cfg
        h = self.patch_embed(x).flatten(2).transpose(1, 2)
        h = torch.cat([self.cls.expand(h.shape[0], -1, -1), h], 1) + self.enc_pos
        for l in self.enc:
            y = l.ln1(h)
            q, k, v = l.qkv(y).chunk(3, -1)
            h = h + l.proj(T.attention_reference(q, k, v, cfg.enc_heads, 0.125))
            h = h + l.fc2(F.gelu(l.fc1(l.ln2(h))))
        return self.enc_ln(h)

    def decode_logits_reference(self, memory: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
        """Teacher-forced decoder: tokens [B, L] -> logits [B, L, vocab]."""
        cfg = self.cfg
        L = tokens.shape[1]
        act = F.gelu if cfg.act == "gelu" else F.relu
        x = self.tok(tokens) * cfg.embed_scale + self.pos(torch.arange(L, device=tokens.device) + cfg.pos_offset)[None]
        if self.emb_ln is not None:
            x = self.emb_ln(x)
        for l in self.dec:
            xk, xv = l.xkv(memory).chunk(2, -1)
            if cfg.post_ln:
                q, k, v = l.qkv(x).chunk(3, -1)
                x = l.ln1(x + l.proj(T.attention_reference(q, k, v, cfg.dec_heads, 0.125, causal=True)))
                x = l.ln2(x + l.xproj(T.attention_reference(l.xq(x), xk, xv, cfg.dec_heads, 0.125)))
                x = l.ln3(x + l.fc2(act(l.fc1(x))))
            else:
                q, k, v = l.qkv(l.ln1(x)).chunk(3, -1)
                x = x + l.proj(T.attention_reference(q, k, v, cfg.dec_heads, 0.125, causal=True))
                x = x + l.xproj(T.attention_reference(l.xq(l.ln2(x)), xk, xv, cfg.dec_heads, 0.125))
                x = x + l.fc2(act(l.fc1(l.ln3(x))))
        return self.lm_head(self.dec_ln(x) if self.dec_ln is not None else x)

    @torch.no_grad()
    def generate_reference(self, x: torch.Tensor, max_len: int) -> torch.Tensor:
        mem = self.encode_reference(x)
        toks = torch.full((x.shape[0], 1), self.cfg.bos, dtype=torch.long, device=x.device)
        for _ in range(max_len):
            nxt = self.decode_logits_reference(mem, toks)[:, -1].argmax(-1, keepdim=True)
            toks = torch.cat([toks, nxt], 1)
        return toks[:, 1:]

    # ------------------------------------------------------------------ MI355X inference path
    def prepare(self) -> "TrOCR":
        self.eval()
        cfg = self.cfg
        dev = self.lm_head.weight.device
        pk: Dict = {}
        w = self.patch_embed.weight.detach().float()                       # [D, 3, p, p]
        wp = torch.zeros(cfg.enc_dim, cfg.patch, cfg.patch, 8, device=dev)
        wp[..., :3] = w.permute(0, 2, 3, 1)
        pk["patch"] = (wp.to(torch.bfloat16).contiguous(), self.patch_embed.bias.detach().float().contiguous())
        pk["cls_pos"] = (self.cls.detach() + self.enc_pos.detach()[:, :1]).to(torch.bfloat16).contiguous()
        pk["pos"] = self.enc_pos.detach()[:, 1:].to(torch.bfloat16).contiguous()
        pk["enc"] = [dict(ln1=_ln(l.ln1), ln2=_ln(l.ln2), qkv=_pk(l.qkv), proj=_pk(l.proj), fc1=_pk(l.fc1),
                          fc2=_pk(l.fc2)) for l in self.enc]
        pk["enc_ln"] = _ln(self.enc_ln)
        pk["dec"] = [dict(ln1=_ln(l.ln1), ln2=_ln(l.ln2), ln3=_ln(l.ln3), qkv=_pk(l.qkv), proj=_pk(l.proj),
                          xq=_pk(l.xq), xkv=_pk(l.xkv), xproj=_pk(l.xproj), fc1=_pk(l.fc1), fc2=_pk(l.fc2))
                     for l in self.dec]
        pk["dec_ln"] = _ln(self.dec_ln) if self.dec_ln is not None else None
        pk["emb_ln"] = _ln(self.emb_ln) if self.emb_ln is not None else None
        pk["tok"] = (self.tok.weight.detach() * cfg.embed_scale).to(torch.bfloat16).contiguous()
        pk["dpos"] = self.pos.weight.detach()[cfg.pos_offset:].to(torch.bfloat16).contiguous()
        lw = torch.zeros(cfg.vocab_padded, cfg.dec_dim, device=dev)
        lw[: cfg.vocab] = self.lm_head.weight.detach().float()
        lb = torch.full((cfg.vocab_padded,), -1e4, device=dev)
        lb[: cfg.vocab] = 0.0
        pk["lm"] = (lw.to(torch.bfloat16).contiguous(), lb.contiguous())
        self._pk = pk
        self._graphs = {}
        return self

    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B, S, S, 8] NHWC bf16 (3 real channels) -> memory [B, tokens, enc_dim] bf16."""
        if self._pk is None:
            self.prepare()
        cfg, pk = self.cfg, self._pk
        B = x.shape[0]
        n = cfg.tokens
        h = torch.empty((B, n, cfg.enc_dim), dtype=torch.bfloat16, device=x.device)
        pe = conv2d_nhwc(x, pk["patch"][0], pk["patch"][1], stride=cfg.patch)      # [B, g, g, D]
        h[:, 1:] = pe.reshape(B, n - 1, cfg.enc_dim) + pk["pos"]
        h[:, :1] = pk["cls_pos"]
        eps = cfg.enc_ln_eps
        for l in pk["enc"]:
            y = T.layernorm(h, *l["ln1"], eps=eps)
            qkv = linear_bf16(y, *l["qkv"])
            d = cfg.enc_dim
            a = T.attention(qkv[..., :d], qkv[..., d : 2 * d], qkv[..., 2 * d :], cfg.enc_heads)
            h = linear_bf16(a, *l["proj"], res=h)
            y = T.layernorm(h, *l["ln2"], eps=eps)
            y = linear_bf16(y, *l["fc1"], act="gelu")
            h = linear_bf16(y, *l["fc2"], res=h)
        return T.layernorm(h, *pk["enc_ln"], eps=eps)

    def _decode_step(self, st: "_DecodeState") -> None:
        """One greedy step on the static state (tokens, t, caches) - the body that gets graph-captured."""
        cfg, pk = self.cfg, self._pk
        d = cfg.dec_dim
        x = T.embed_rows(st.tokens, pk["tok"], pk["dpos"], st.t)
        if pk["emb_ln"] is not None:
            x = T.layernorm(x, *pk["emb_ln"])
        post = cfg.post_ln
        for li, l in enumerate(pk["dec"]):
            y = x if post else T.layernorm(x, *l["ln1"])
            qkv = linear_bf16(y, *l["qkv"])
            a = T.attention_decode(qkv[:, :d], st.k_self[li], st.v_self[li], cfg.dec_heads,
                                   k_new=qkv[:, d : 2 * d], v_new=qkv[:, 2 * d :], t=st.t)
            x = linear_bf16(a, *l["proj"], res=x)
            if post:
                x = T.layernorm(x, *l["ln1"])
            y = x if post else T.layernorm(x, *l["ln2"])
            q = linear_bf16(y, *l["xq"])
            a = T.attention_decode(q, st.k_cross[li], st.v_cross[li], cfg.dec_heads, klen=cfg.tokens)
            x = linear_bf16(a, *l["xproj"], res=x)
            if post:
                x = T.layernorm(x, *l["ln2"])
            y = x if post else T.layernorm(x, *l["ln3"])
            y = linear_bf16(y, *l["fc1"], act=cfg.act)
            x = linear_bf16(y, *l["fc2"], res=x)
            if post:
                x = T.layernorm(x, *l["ln3"])
        if pk["dec_ln"] is not None:
            x = T.layernorm(x, *pk["dec_ln"])
        logits = linear_bf16(x, *pk["lm"])
        ctc_ops.row_argmax(logits, out=(st.tokens, st.probs))
        # history, probability sums, EOS marks, step counter and the all-done flag: one small kernel inside the step
        T.decode_bookkeep(st.tokens, st.probs, st.t, st.out, st.psum, st.cnt, st.done, st.flag, cfg.eos)

    ENC_CHUNK = 256

    @torch.no_grad()
    def generate(self, x: torch.Tensor, max_len: Optional[int] = None, use_graph: bool = True,
                 eos_check_every: int = 8) -> Tuple[torch.Tensor, torch.Tensor]:
        """x: [B, S, S, 8] bf16 -> (tokens [B, max_len] int32, mean token probability [B]).

        Decoding stops once every sequence of the batch has produced ``eos``: the step itself raises a flag in pinned
        host memory, and every ``eos_check_every`` steps the host waits for the step it launched that many steps ago
        (never for the newest one - the queue stays filled) and looks at the flag.  Positions after a sequence's
        ``eos`` are filled with ``pad`` and do not count towards its mean probability.  0 disables the early exit."""
        if self._pk is None:
            self.prepare()
        cfg, pk = self.cfg, self._pk
        max_len = min(max_len or cfg.max_len, cfg.max_len)
        B = x.shape[0]
        on_gpu = x.is_cuda
        Bb = B if not on_gpu else max(16, 1 << (B - 1).bit_length())
        g = self._graphs.get(Bb)
        if g is None:
            g = _DecodeGraph(self, Bb, capture=on_gpu and use_graph)
            self._graphs[Bb] = g
        st = g.state
        # encoder and the cross-attention K/V of every layer, once per batch, a chunk of crops at a time
        for c0 in range(0, B, self.ENC_CHUNK):
            mem = self.encode(x[c0 : c0 + self.ENC_CHUNK])
            for li, l in enumerate(pk["dec"]):
                kv = linear_bf16(mem, *l["xkv"])
                st.k_cross[li][c0 : c0 + mem.shape[0]].copy_(kv[..., : cfg.dec_dim])
                st.v_cross[li][c0 : c0 + mem.shape[0]].copy_(kv[..., cfg.dec_dim :])
        st.reset(B, cfg)
        events = []
        steps = 0
        for s in range(max_len):
            g.step()
            steps = s + 1
            if eos_check_every > 0 and on_gpu:
                ev = torch.cuda.Event()
                ev.record()
                events.append(ev)
                if steps % eos_check_every == 0 and steps < max_len:
                    if len(events) > eos_check_every:
                        events[-1 - eos_check_every].synchronize()       # an OLD step: the newer ones keep the GPU busy
                    if int(st.flag[0]) != 0:
                        break
            elif eos_check_every > 0 and int(st.flag[0]) != 0:
                break
        self.last_decode_steps = steps
        out = st.out[:max_len, :B].t().contiguous()
        return out, (st.psum[:B] / st.cnt[:B].clamp(min=1.0))

    def tokens_to_text(self, tokens: torch.Tensor, vocab=None) -> List[str]:
        """ids -> text through the native byte-level vocabulary (csrc/runtime/bpe_tokenizer.cpp).  Without a
        loaded vocab.json the built-in one is used: ids 0..3 are specials (bos, pad, eos, unk), 4..259 the raw bytes."""
        from ..runtime.tokenizer import ByteLevelVocab

        if vocab is None:
            vocab = getattr(self, "_vocab", None)
            if vocab is None:
                vocab = self._vocab = ByteLevelVocab()
        return vocab.decode_batch(tokens.detach().cpu().numpy(), stop_at_eos=True)

    def flops_per_crop(self, decode_tokens: int) -> float:
        cfg = self.cfg
        n, d, m = cfg.tokens, cfg.enc_dim, cfg.enc_mlp
        enc = cfg.enc_layers * (2.0 * n * d * (4 * d + 2 * m) + 4.0 * n * n * d) + 2.0 * (n - 1) * 3 * cfg.patch ** 2 * d
        dd, f = cfg.dec_dim, cfg.dec_ffn
        cross_kv = cfg.dec_layers * 2.0 * n * d * 2 * dd
        per_tok = cfg.dec_layers * (2.0 * dd * (4 * dd + 2 * dd + 2 * f) + 4.0 * n * dd) + 2.0 * dd * cfg.vocab
        return enc + cross_kv + decode_tokens * per_tok


class _DecodeState:
    def __init__(self, model: TrOCR, Bb: int, device):
        cfg = model.cfg
        z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=device)
        self.tokens = torch.zeros((Bb,), dtype=torch.int32, device=device)
        self.probs = torch.zeros((Bb,), dtype=torch.float32, device=device)
        self.t = torch.zeros((1,), dtype=torch.int32, device=device)
        self.k_self = [z(Bb, cfg.max_len, cfg.dec_dim) for _ in range(cfg.dec_layers)]
        self.v_self = [z(Bb, cfg.max_len, cfg.dec_dim) for _ in range(cfg.dec_layers)]
        self.k_cross = [z(Bb, cfg.tokens, cfg.dec_dim) for _ in range(cfg.dec_layers)]
        self.v_cross = [z(Bb, cfg.tokens, cfg.dec_dim) for _ in range(cfg.dec_layers)]
        # bookkeeping of the greedy loop, updated by the last kernel of every step (ops.transformer.decode_bookkeep)
        self.out = torch.full((cfg.max_len, Bb), cfg.pad, dtype=torch.int32, device=device)
        self.psum = torch.zeros((Bb,), dtype=torch.float32, device=device)
        self.cnt = torch.zeros((Bb,), dtype=torch.float32, device=device)
        self.done = torch.zeros((Bb,), dtype=torch.uint8, device=device)
        on_gpu = torch.device(device).type == "cuda"
        # "every sequence has emitted eos", written by the device, read by the host without a stream synchronisation
        self.flag = torch.zeros((1,), dtype=torch.int32, pin_memory=True) if on_gpu else torch.zeros((1,), dtype=torch.int32)

    def reset(self, B: int, cfg) -> None:
        """Start a new batch of ``B`` sequences (rows beyond B are padding of the batch bucket: done from the start)."""
        self.tokens.fill_(cfg.bos)
        self.t.zero_()
        self.out.fill_(cfg.pad)
        self.psum.zero_()
        self.cnt.zero_()
        self.done.zero_()
        self.done[B:] = 1
        if self.flag.is_pinned():
            torch.cuda.current_stream().synchronize()        # no earlier step may still raise the flag we clear now
        self.flag.zero_()


class _DecodeGraph:
    """Static decode state + (on GPU) the captured hipGraph of one greedy step for a batch bucket."""

    def __init__(self, model: TrOCR, Bb: int, capture: bool):
        self.model = model
        dev = model.lm_head.weight.device
        self.state = _DecodeState(model, Bb, dev)
        self.graph = None
        self.replays = 0
        if capture:
            cur = torch.cuda.current_stream(dev)
            s = torch.cuda.Stream(dev)
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                model._decode_step(self.state)          # warm-up: allocations, lazily set kernel attributes
            cur.wait_stream(s)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=s):
                model._decode_step(self.state)

    def step(self) -> None:
        if self.graph is not None:
            self.graph.replay()
            self.replays += 1
        else:
            self.model._decode_step(self.state)
