"""TrOCR-sized recognizer: fused inference path vs PyTorch fp32 reference (CPU oracle; GPU kernels on MI355X)."""
import numpy as np
import pytest
import torch

from distributed_llms_amd.models.trocr import TrOCR, TrOCRConfig
from distributed_llms_amd.ops import transformer as T

SMALL = TrOCRConfig(image_size=64, patch=16, enc_dim=128, enc_layers=2, enc_heads=2, enc_mlp=256, dec_dim=128,
                    dec_layers=2, dec_heads=2, dec_ffn=256, vocab=300, max_len=12)


def _nhwc8(x):
    B, C, H, W = x.shape
    out = torch.zeros(B, H, W, 8, dtype=x.dtype)
    out[..., :C] = x.permute(0, 2, 3, 1)
    return out


def _check_model(device):
    torch.manual_seed(0)
    m = TrOCR(SMALL).eval()
    # make the logits decisive so bf16 rounding cannot flip the argmax
    with torch.no_grad():
        m.lm_head.weight.mul_(8.0)
        m.tok.weight.mul_(4.0)
    x = torch.randn(3, 3, 64, 64)
    with torch.no_grad():
        mem_ref = m.encode_reference(x)
    m = m.to(device).prepare()
    xin = _nhwc8(x).to(torch.bfloat16).to(device)
    mem = m.encode(xin).float().cpu()
    assert mem.shape == (3, SMALL.tokens, SMALL.enc_dim)
    err = (mem - mem_ref).abs()
    assert err.mean().item() < 0.03 and err.max().item() < 0.3, (err.mean().item(), err.max().item())
    toks, conf = m.generate(xin, max_len=10)
    assert toks.shape == (3, 10) and toks.dtype == torch.int32 and torch.isfinite(conf).all()
    # teacher-force the reference on the produced prefix: every chosen token must be (near-)optimal for it
    mc = m.to("cpu").float()
    with torch.no_grad():
        prefix = torch.cat([torch.full((3, 1), SMALL.bos, dtype=torch.long), toks.cpu().long()[:, :-1]], 1)
        ref_logits = mc.decode_logits_reference(mem_ref, prefix)
    chosen = ref_logits.gather(-1, toks.cpu().long().unsqueeze(-1)).squeeze(-1)
    gap = ref_logits.max(-1).values - chosen
    assert (gap < 0.15 * ref_logits.std()).float().mean().item() > 0.9, gap
    assert isinstance(mc.tokens_to_text(toks)[0], str)
    return toks


def test_trocr_fused_path_matches_reference_cpu():
    _check_model("cpu")


@pytest.mark.gpu
def test_trocr_fused_path_matches_reference_gpu():
    toks = _check_model("cuda")
    # graph replay == eager stepping
    torch.manual_seed(0)
    m = TrOCR(SMALL).eval()
    with torch.no_grad():
        m.lm_head.weight.mul_(8.0)
        m.tok.weight.mul_(4.0)
    m = m.cuda().prepare()
    x = _nhwc8(torch.randn(3, 3, 64, 64)).to(torch.bfloat16).cuda()
    t_graph, _ = m.generate(x, max_len=10, use_graph=True)
    assert m._graphs and next(iter(m._graphs.values())).replays == 10
    m._graphs = {}
    t_eager, _ = m.generate(x, max_len=10, use_graph=False)
    assert torch.equal(t_graph, t_eager)


@pytest.mark.gpu
@pytest.mark.parametrize("Sq,Sk,causal", [(577, 577, False), (100, 100, True), (64, 130, False), (17, 17, False)])
def test_attention_kernel_matches_reference(Sq, Sk, causal):
    g = torch.Generator().manual_seed(Sq)
    B, H = 2, 3
    qkv = (torch.randn(B, max(Sq, Sk), 3 * H * 64, generator=g)).to(torch.bfloat16).cuda()
    d = H * 64
    q, k, v = qkv[:, :Sq, :d], qkv[:, :Sk, d : 2 * d], qkv[:, :Sk, 2 * d :]
    # asymmetric content: make V depend on position so a transposed / permuted PV product cannot pass
    v = (v.float() + torch.arange(Sk, device="cuda").view(1, Sk, 1) * 0.01).to(torch.bfloat16)
    got = T.attention(q, k, v, H, causal=causal).float().cpu()
    ref = T.attention_reference(q.cpu(), k.cpu(), v.cpu(), H, 0.125, causal)
    err = (got - ref).abs()
    assert err.max().item() < 0.05 and err.mean().item() < 4e-3, (err.max().item(), err.mean().item())


@pytest.mark.gpu
def test_attention_softmax_rescale_branch_is_exercised():
    """A key far down the sequence with a huge score forces the running max to jump in a late block."""
    B, H, S = 1, 1, 300
    q = torch.zeros(B, S, 64)
    k = torch.zeros(B, S, 64)
    q[..., 0] = 1.0
    k[0, 250, 0] = 60.0            # spike inside the 4th K block
    k[0, 10, 0] = 20.0
    v = torch.randn(B, S, 64)
    qb, kb, vb = (t.to(torch.bfloat16).cuda() for t in (q, k, v))
    got = T.attention(qb, kb, vb, H).float().cpu()
    ref = T.attention_reference(qb.cpu(), kb.cpu(), vb.cpu(), H, 0.125)
    assert (got - ref).abs().max().item() < 0.03


@pytest.mark.gpu
def test_layernorm_decode_attention_and_embedding_kernels():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 768, generator=g).to(torch.bfloat16)
    r = torch.randn(37, 768, generator=g).to(torch.bfloat16)
    gm, bt = torch.rand(768, generator=g) + 0.5, torch.randn(768, generator=g) * 0.1
    for res in (None, r):
        got = T.layernorm(x.cuda(), gm.cuda(), bt.cuda(), res=None if res is None else res.cuda()).float().cpu()
        ref = T.layernorm(x, gm, bt, res=res).float()
        assert (got - ref).abs().max().item() < 0.04
    B, H, Tcap = 5, 4, 16
    d = H * 64
    kc, vc = torch.zeros(B, Tcap, d, dtype=torch.bfloat16), torch.zeros(B, Tcap, d, dtype=torch.bfloat16)
    kg, vg = kc.clone().cuda(), vc.clone().cuda()
    t_c, t_g = torch.zeros(1, dtype=torch.int32), torch.zeros(1, dtype=torch.int32).cuda()
    for step in range(6):
        qkv = torch.randn(B, 3 * d, generator=g).to(torch.bfloat16)
        ref = T.attention_decode(qkv[:, :d], kc, vc, H, k_new=qkv[:, d : 2 * d], v_new=qkv[:, 2 * d :], t=t_c)
        qg = qkv.cuda()
        got = T.attention_decode(qg[:, :d], kg, vg, H, k_new=qg[:, d : 2 * d], v_new=qg[:, 2 * d :], t=t_g)
        assert (got.float().cpu() - ref.float()).abs().max().item() < 0.03
        t_c += 1
        t_g += 1
    assert torch.equal(kg.cpu(), kc)
    mem_k = torch.randn(B, 577, d, generator=g).to(torch.bfloat16)
    mem_v = torch.randn(B, 577, d, generator=g).to(torch.bfloat16)
    q = torch.randn(B, d, generator=g).to(torch.bfloat16)
    ref = T.attention_decode(q, mem_k, mem_v, H, klen=577)
    got = T.attention_decode(q.cuda(), mem_k.cuda(), mem_v.cuda(), H, klen=577)
    assert (got.float().cpu() - ref.float()).abs().max().item() < 0.03
    tok = torch.randn(50, 128, generator=g).to(torch.bfloat16)
    pos = torch.randn(9, 128, generator=g).to(torch.bfloat16)
    ids = torch.tensor([0, 49, 7, 7], dtype=torch.int32)
    t3 = torch.tensor([3], dtype=torch.int32)
    ref = T.embed_rows(ids, tok, pos, t3)
    got = T.embed_rows(ids.cuda(), tok.cuda(), pos.cuda(), t3.cuda())
    assert (got.float().cpu() - ref.float()).abs().max().item() < 0.02


def _engine_pages(n=3):
    import torch
    from distributed_llms_amd.pdf import synth
    from distributed_llms_amd.runtime.engine import PageImage

    pages = []
    for s in range(n):
        pg, tr = synth.render_page(s, size=(248, 350), font_px=10, margin=16)
        pages.append(PageImage(torch.from_numpy(pg), "d", s, layout_prior=tr.boxes))
    return pages


def test_engine_runs_with_the_transformer_recognizer_cpu():
    """OCR_RECOGNIZER=trocr: the engine crops to S x S, runs encoder + greedy decoder, decodes tokens
    with the native byte-level vocabulary, and is deterministic."""
    from distributed_llms_amd.runtime.workers import build_engine

    eng = build_engine("cpu", {"model": "tiny", "recognizer": "trocr",
                               "cfg": dict(det_limit_side=160, max_batch_pages=2, trocr_max_tokens=6, trocr_batch=8)})
    assert eng.rec_kind == "trocr"
    pages = _engine_pages()
    a = eng.process_pages(pages)
    b = eng.process_pages(pages)
    assert [len(r.lines) for r in a] == [len(p.layout_prior) for p in pages]
    assert [l.text for r in a for l in r.lines] == [l.text for r in b for l in r.lines]
    assert all(0.0 <= l.confidence <= 1.0 for r in a for l in r.lines)


@pytest.mark.gpu
def test_engine_transformer_recognizer_gpu_matches_eager_decode():
    """Same pages through the GPU engine: hipGraph-replayed decode steps, two pipeline slots sharing one decode
    state (ordered by an event), against the same engine without graphs."""
    import torch
    from distributed_llms_amd.models.dbnet import DBNet
    from distributed_llms_amd.runtime.engine import EngineConfig, OcrEngine

    torch.manual_seed(3)
    det = DBNet(widths=(16, 32, 32, 64), stem=8, inner=32)
    rec = TrOCR(SMALL)
    pages = _engine_pages(5)
    outs = []
    for graphs in (True, False):
        eng = OcrEngine("cuda:0", det, rec, EngineConfig(det_limit_side=160, max_batch_pages=2, trocr_max_tokens=8,
                                                         trocr_batch=16, use_hip_graphs=graphs))
        r1 = eng.process_pages(pages)
        r2 = eng.process_pages(pages)
        assert [l.text for r in r1 for l in r.lines] == [l.text for r in r2 for l in r.lines]
        outs.append([l.text for r in r1 for l in r.lines])
        assert [len(r.lines) for r in r1] == [len(p.layout_prior) for p in pages]
    assert outs[0] == outs[1]


def _tiny_trocr():
    from distributed_llms_amd.models.trocr import TrOCR, TrOCRConfig

    torch.manual_seed(0)
    return TrOCR(TrOCRConfig(image_size=64, enc_dim=128, enc_layers=2, enc_heads=2, enc_mlp=256, dec_dim=128, dec_layers=2,
                             dec_heads=2, dec_ffn=256, vocab=272, max_len=32))


def test_cross_entropy_trainer_overfits_64_synthetic_lines():
    """Teacher-forced cross-entropy training of the encoder-decoder (train/trocr_trainer.py): 64 fixed strings,
    fresh renderings every step; greedy decoding of new renderings must reproduce the strings."""
    from distributed_llms_amd.train.trocr_trainer import Seq2SeqConfig, Seq2SeqTrainer

    tr = Seq2SeqTrainer(_tiny_trocr(), Seq2SeqConfig(batch_size=32, width=256, lr=2e-3, vocab=64, max_chars=12,
                                                     warmup_steps=10, decay_steps=700), "cpu")
    tr.fit(700)
    first, last = tr.history[0]["loss"], float(np.mean([h["loss"] for h in tr.history[-20:]]))
    assert first > 3.0 and last < 0.03, (first, last)
    ev = tr.evaluate(n_batches=2)
    assert ev["exact"] >= 0.9 and ev["cer"] < 0.03, ev


def test_generate_stops_at_eos_and_pads_after_it():
    m = _tiny_trocr()
    # an untrained model never emits eos on its own: force it by making eos the most likely token after 3 steps
    with torch.no_grad():
        m.pos.weight[3:] += 4.0 * m.lm_head.weight[m.cfg.eos][None]
        m.lm_head.weight[m.cfg.eos] *= 3.0
    x = torch.randn(3, 64, 64, 8).to(torch.bfloat16)
    x[..., 3:] = 0
    toks, conf = m.generate(x, max_len=32, use_graph=False, eos_check_every=4)
    assert m.last_decode_steps < 32                              # left the loop early ...
    t = toks.numpy()
    for row in t:
        k = int(np.argmax(row == m.cfg.eos))
        assert row[k] == m.cfg.eos and (row[k + 1:] == m.cfg.pad).all()     # ... and nothing but pad follows eos
    full, conf_full = m.generate(x, max_len=32, use_graph=False, eos_check_every=0)
    assert np.array_equal(full.numpy(), t) and torch.allclose(conf, conf_full)
    assert all(s == "" or isinstance(s, str) for s in m.tokens_to_text(toks))


@pytest.mark.gpu
def test_trained_trocr_decodes_through_the_hip_inference_path():
    """Train the small encoder-decoder on the GPU, then decode with the inference path (HIP attention / layernorm /
    GEMM kernels, KV cache, graph-captured steps): same strings as the PyTorch reference decode."""
    from distributed_llms_amd.train.trocr_trainer import Seq2SeqConfig, Seq2SeqTrainer

    tr = Seq2SeqTrainer(_tiny_trocr(), Seq2SeqConfig(batch_size=32, width=256, lr=2e-3, vocab=32, max_chars=12,
                                                     warmup_steps=10, decay_steps=600), "cuda:0")
    tr.fit(600)
    assert float(np.mean([h["loss"] for h in tr.history[-20:]])) < 0.03, tr.history[-1]
    ref = tr.evaluate(n_batches=1)
    hip = tr.evaluate(n_batches=1, through_kernels=True)
    assert ref["exact"] >= 0.9 and hip["exact"] >= 0.85 and hip["cer"] < 0.05, (ref, hip)
