"""Model parity: fused NHWC inference path vs plain PyTorch NCHW path (same parameters)."""
import pytest
import torch

from distributed_llms_amd.models.crnn import CRNN
from distributed_llms_amd.models.dbnet import DBNet


def _randomize_bn(model):
    g = torch.Generator().manual_seed(0)
    if hasattr(model, "head3"):
        with torch.no_grad():   # undo the prior-probability init so the probability map has real contrast
            model.head3.deconv.weight.mul_(100.0)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)


def _nhwc8(x):
    B, C, H, W = x.shape
    out = torch.zeros(B, H, W, 8, dtype=x.dtype)
    out[..., :C] = x.permute(0, 2, 3, 1)
    return out


def test_dbnet_infer_matches_forward_cpu():
    torch.manual_seed(0)
    net = DBNet(widths=(16, 32, 32, 64), stem=8, inner=32).eval()
    _randomize_bn(net)
    x = torch.randn(2, 3, 64, 96)
    with torch.no_grad():
        ref = net(x)
    prob, mask = net.infer(_nhwc8(x).to(torch.bfloat16))
    assert prob.shape == (2, 64, 96) and mask.shape == (2, 64, 96) and mask.dtype == torch.uint8
    # bf16 activations between layers in the oracle path -> loose tolerance
    err = (prob.float() - ref).abs()
    assert err.mean().item() < 0.01 and err.max().item() < 0.25


def test_crnn_infer_matches_forward_cpu():
    torch.manual_seed(0)
    net = CRNN(hidden=64, widths=(16, 32, 32, 32, 64, 64)).eval()
    _randomize_bn(net)
    x = torch.randn(3, 3, 48, 64) * 0.5
    lens = torch.tensor([16, 9, 3], dtype=torch.int32)
    with torch.no_grad():
        ref = net(x, lens)
    logits = net.infer_logits(_nhwc8(x).to(torch.bfloat16), lens).float()
    assert logits.shape == (3, 16, net.vocab_padded)
    for n in range(3):
        L = int(lens[n])
        assert (logits[n, :L, : net.num_classes] - ref[n, :L]).abs().max().item() < 0.08
    ids, ln, conf = net.infer(_nhwc8(x).to(torch.bfloat16), lens)
    assert ids.shape == (3, 16) and (ln <= lens).all()
    strs = net.decode_strings(ids, ln)
    assert len(strs) == 3 and all(isinstance(s, str) for s in strs)
    assert net.encode_text("Hi!") == [ord("H") - 31, ord("i") - 31, ord("!") - 31]


@pytest.mark.gpu
def test_dbnet_gpu_matches_cpu_oracle():
    torch.manual_seed(0)
    net = DBNet().eval()
    _randomize_bn(net)
    x = (torch.randn(2, 3, 96, 160) * 0.7)
    x8 = _nhwc8(x).to(torch.bfloat16)
    ref_prob, _ = net.infer(x8)
    net = net.to("cuda").prepare()
    prob, mask = net.infer(x8.to("cuda"))
    torch.cuda.synchronize()
    err = (prob.float().cpu() - ref_prob.float()).abs()
    assert err.max().item() < 0.06, err.max().item()
    assert err.mean().item() < 0.01
    assert bool(((mask.cpu() > 0) == (prob.cpu().float() > net.bin_thresh)).all())


@pytest.mark.gpu
def test_crnn_gpu_matches_cpu_oracle():
    torch.manual_seed(0)
    net = CRNN().eval()
    _randomize_bn(net)
    x = (torch.randn(5, 3, 48, 160) * 0.5)
    lens = torch.tensor([40, 33, 17, 40, 1], dtype=torch.int32)
    x8 = _nhwc8(x).to(torch.bfloat16)
    ref = net.infer_logits(x8, lens).float()
    net = net.to("cuda").prepare()
    out = net.infer_logits(x8.to("cuda"), lens.to("cuda")).float().cpu()
    for n in range(5):
        L = int(lens[n])
        d = (out[n, :L] - ref[n, :L]).abs()
        assert d.max().item() < 0.15, (n, d.max().item())
    ids, ln, conf = net.infer(x8.to("cuda"), lens.to("cuda"))
    assert ids.shape == (5, 40) and bool((ln.cpu() <= lens).all())


def _ragged_case(device):
    from distributed_llms_amd.ops.lstm import RaggedLayout

    torch.manual_seed(0)
    net = CRNN(hidden=64, widths=(16, 32, 32, 32, 64, 64)).eval()
    _randomize_bn(net)
    net = net.to(device).prepare()
    xs = [torch.randn(3, 48, 64, 8) * 0.5, torch.randn(2, 48, 128, 8) * 0.5]
    for x in xs:
        x[..., 3:] = 0
    xs = [x.to(torch.bfloat16).to(device) for x in xs]
    lens = [torch.tensor([16, 9, 3], dtype=torch.int32), torch.tensor([32, 20], dtype=torch.int32)]
    tcap = torch.tensor([16, 16, 16, 32, 32], dtype=torch.int32)
    ln = torch.cat(lens)
    row_off = (torch.cumsum(tcap.long(), 0) - tcap.long())
    perm = torch.argsort(-ln.long(), stable=True).to(torch.int32)
    lay = RaggedLayout(row_off.to(device), tcap.to(device), ln.to(device), perm.to(device), int(tcap.sum()), 32)
    ids_r, len_r, conf_r = net.infer_ragged(xs, lay)
    k = 0
    for x, l in zip(xs, lens):
        ids, n, conf = net.infer(x, l.to(device))
        for j in range(x.shape[0]):
            L = int(n[j])
            assert int(len_r[k]) == L
            assert torch.equal(ids_r[k, :L].cpu(), ids[j, :L].cpu())
            assert abs(float(conf_r[k]) - float(conf[j])) < 2e-2
            k += 1


def test_crnn_ragged_equals_per_bucket_cpu():
    _ragged_case("cpu")


@pytest.mark.gpu
def test_crnn_ragged_equals_per_bucket_gpu():
    _ragged_case("cuda")


def test_balanced_chunks_cut_a_document_into_equal_batches():
    from distributed_llms_amd.runtime.engine import balanced_chunks

    assert balanced_chunks(100, 36) == [(0, 34), (34, 67), (67, 100)]
    assert balanced_chunks(100, 25) == [(0, 25), (25, 50), (50, 75), (75, 100)]
    assert balanced_chunks(5, 36) == [(0, 5)] and balanced_chunks(0, 36) == []
    assert balanced_chunks(37, 36) == [(0, 19), (19, 37)]
    for n in range(1, 200):
        ch = balanced_chunks(n, 7)
        sizes = [b - a for a, b in ch]
        assert ch[0][0] == 0 and ch[-1][1] == n and all(ch[i][1] == ch[i + 1][0] for i in range(len(ch) - 1))
        assert max(sizes) <= 7 and max(sizes) - min(sizes) <= 1 and len(ch) == -(-n // 7)


def test_crnn_character_set_follows_the_checkpoint_and_grows_in_place(tmp_path):
    """A checkpoint trained on printable ASCII (96 classifier rows) loads into a model built for the Western European
    set, which then reads ASCII only; growing it back keeps the old classes' rows."""
    from distributed_llms_amd.models.crnn import ASCII_CHARSET, DEFAULT_CHARSET, LATIN_EXTRA
    from distributed_llms_amd.utils.weights import load_model_weights, save_model_weights

    assert DEFAULT_CHARSET.startswith(ASCII_CHARSET) and len(set(DEFAULT_CHARSET)) == len(DEFAULT_CHARSET)
    kw = dict(hidden=64, widths=(16, 32, 32, 32, 64, 64))
    old = CRNN(charset=ASCII_CHARSET, **kw)
    save_model_weights(old, str(tmp_path / "ascii.safetensors"))
    net = CRNN(**kw)
    assert net.num_classes == len(DEFAULT_CHARSET) + 1
    load_model_weights(net, str(tmp_path / "ascii.safetensors"))
    assert net.charset == ASCII_CHARSET and net.fc.out_features == 96
    assert torch.equal(net.fc.weight, old.fc.weight)
    net.set_charset(DEFAULT_CHARSET)
    assert net.fc.out_features == len(DEFAULT_CHARSET) + 1 and torch.equal(net.fc.weight[:96], old.fc.weight)
    assert net.encode_text("Straße 5 €") == [DEFAULT_CHARSET.index(c) + 1 for c in "Straße 5 €"]
    x = torch.randn(2, 3, 48, 64)
    assert net.eval()(x).shape[-1] == len(DEFAULT_CHARSET) + 1
    # every extra character is something the page generator can write
    from distributed_llms_amd.pdf import synth

    vocab = set("".join(synth._WORDS_I18N))
    assert all(c in vocab for c in LATIN_EXTRA)
    pg, truth = synth.render_layout_page(7, size=(620, 877), fonts=["DejaVuSerif.ttf"], i18n=1.0)
    text = "".join(truth.texts)
    assert any(c in LATIN_EXTRA for c in text) and all(c in DEFAULT_CHARSET for c in text)
    # the Computer Modern files carry TeX encodings: only ASCII is drawn with them, and the truth says so
    assert synth._font_safe("Straße déjà 5 €", "cmr10.ttf") == "Strae dj 5"
    assert synth._font_safe("Straße déjà 5 €", "DejaVuSans.ttf") == "Straße déjà 5 €"
