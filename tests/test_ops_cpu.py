"""CPU checks of the op oracles / CPU dispatch paths (no GPU needed).

These pin the *references* the GPU kernel tests compare against to independent
PyTorch / SciPy formulations.
"""
import numpy as np
import torch
import torch.nn.functional as F

from distributed_llms_amd.ops import conv as C
from distributed_llms_amd.ops import ctc, dbpost, lstm, resample


def test_conv_reference_layouts():
    torch.manual_seed(0)
    x = torch.randn(2, 9, 11, 8)
    w = torch.randn(16, 3, 3, 8) * 0.1
    b = torch.randn(16)
    y = C.conv2d_nhwc(x, w, b, stride=2, padding=1, act="relu", out_dtype=torch.float32)
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b, stride=2, padding=1)).permute(0, 2, 3, 1)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)


def test_pixel_shuffle_equals_conv_transpose():
    torch.manual_seed(0)
    Cin, Cp = 8, 4
    x = torch.randn(2, 5, 7, Cin)
    wt = torch.randn(Cin, Cp, 2, 2) * 0.2          # ConvTranspose2d weight layout
    bias = torch.randn(Cp)
    ref = F.conv_transpose2d(x.permute(0, 3, 1, 2), wt, bias, stride=2).permute(0, 2, 3, 1)
    # GEMM weight rows ordered (dy, dx, c')
    w = wt.permute(2, 3, 1, 0).reshape(4 * Cp, 1, 1, Cin)
    y = C.conv2d_nhwc(x, w, bias.repeat(4), out_mode=C.OUT_PIXEL_SHUFFLE, out_dtype=torch.float32)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)


def test_prob_mode_equals_conv_transpose():
    torch.manual_seed(0)
    x = torch.randn(1, 4, 6, 8)
    wt = torch.randn(8, 1, 2, 2)
    ref = torch.sigmoid(F.conv_transpose2d(x.permute(0, 3, 1, 2), wt, None, stride=2))[:, 0]
    w = wt.permute(2, 3, 1, 0).reshape(4, 1, 1, 8)
    p, m = C.conv2d_nhwc(x, w, None, act="sigmoid", out_mode=C.OUT_PROB, thr=0.5, out_dtype=torch.float32)
    torch.testing.assert_close(p, ref, rtol=1e-5, atol=1e-5)
    assert torch.equal(m > 0, ref > 0.5)


def test_upsample_residual_and_replicate():
    torch.manual_seed(0)
    x = torch.randn(1, 6, 8, 8)
    w = torch.randn(8, 1, 1, 8)
    r = torch.randn(1, 3, 4, 8)
    y = C.conv2d_nhwc(x, w, None, res=r, res_mode=C.RES_UP2, out_dtype=torch.float32)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2)) + F.interpolate(r.permute(0, 3, 1, 2), scale_factor=2)
    torch.testing.assert_close(y, ref.permute(0, 2, 3, 1), rtol=1e-5, atol=1e-5)
    buf = torch.zeros(1, 12, 16, 24)
    C.conv2d_nhwc(x, w, None, out=buf, out_coff=8, out_mode=C.OUT_REPLICATE, out_f=2, out_dtype=torch.float32)
    ref2 = F.interpolate(F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2)), scale_factor=2).permute(0, 2, 3, 1)
    torch.testing.assert_close(buf[..., 8:16], ref2, rtol=1e-5, atol=1e-5)
    assert buf[..., :8].abs().sum() == 0


def test_resample_reference_integer_area():
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (40, 60), dtype=torch.uint8, generator=g)
    out = resample.resample_regions([resample.Region(img, (0, 0, 60, 40), 20, 30)], 24, 32,
                                    mean=(0, 0, 0), std=(1, 1, 1)).float()
    ref = F.interpolate(img[None, None].float() / 255.0, size=(20, 30), mode="area")[0, 0]
    torch.testing.assert_close(out[0, :20, :30, 0], ref, rtol=0, atol=1e-2)
    assert out[0, 20:].abs().sum() == 0 and out[0, :, 30:].abs().sum() == 0
    torch.testing.assert_close(out[0, :20, :30, 1], out[0, :20, :30, 0])


def test_resample_reference_bilinear_upscale():
    img = (torch.arange(12 * 10).reshape(12, 10) % 251).to(torch.uint8)
    out = resample.resample_regions([resample.Region(img, (0, 0, 10, 12), 24, 20)], 24, 20,
                                    mean=(0, 0, 0), std=(1, 1, 1)).float()
    ref = F.interpolate(img[None, None].float() / 255.0, size=(24, 20), mode="bilinear", align_corners=False)[0, 0]
    torch.testing.assert_close(out[0, :, :, 0], ref, rtol=0, atol=1e-2)


def test_dbpost_reference_boxes():
    prob = torch.zeros(1, 64, 96)
    prob[0, 10:20, 8:80] = 0.9
    prob[0, 40:44, 30:34] = 0.8
    prob[0, 44:48, 34:38] = 0.8   # diagonal neighbour -> same component
    prob[0, 55:58, 5:50] = 0.4    # below box threshold
    mask = (prob > 0.3).to(torch.uint8)
    res = dbpost.db_postprocess(prob, mask, dbpost.DBPostConfig())
    assert len(res) == 1 and res[0].shape == (2, 8)
    b0, b1 = res[0]
    # first in reading order: the wide line, unclipped by d = 72*10*1.5 / (2*82)
    d = 72 * 10 * 1.5 / (2 * 82)
    np.testing.assert_allclose(b0[:4], [8 - d, 10 - d, 80 + d, 20 + d], rtol=1e-5)
    assert abs(b0[4] - 0.9) < 1e-5 and b0[5] == 720
    assert b1[5] == 32
    d2 = 8 * 8 * 1.5 / (2 * 16)
    np.testing.assert_allclose(b1[:4], [30 - d2, 40 - d2, 38 + d2, 48 + d2], rtol=1e-5)


def test_reading_order_sort():
    boxes = np.array(
        [[300, 100, 400, 120, 1, 1], [10, 102, 200, 122, 1, 1], [10, 10, 100, 30, 1, 1], [50, 200, 90, 220, 1, 1]],
        dtype=np.float32,
    )
    out = dbpost.sort_boxes_reading_order(boxes)
    assert out[:, 0].tolist() == [10, 10, 300, 50] and out[:, 1].tolist() == [10, 102, 100, 200]


def test_ctc_reference_collapse():
    V = 5
    seq = [0, 1, 1, 0, 1, 2, 2, 2, 0, 0, 3, 3, 4]
    logits = torch.full((1, len(seq), V), -5.0)
    for t, s in enumerate(seq):
        logits[0, t, s] = 5.0
    ids, ln, conf = ctc.ctc_greedy_decode(logits)
    assert ln.tolist() == [5] and ids[0, :5].tolist() == [1, 1, 2, 3, 4]
    assert ids[0, 5:].eq(-1).all() and 0.9 < conf[0] <= 1.0
    ids, ln, _ = ctc.ctc_greedy_decode(logits, torch.tensor([6], dtype=torch.int32))
    assert ln.tolist() == [3] and ids[0, :3].tolist() == [1, 1, 2]
    assert ctc.ids_to_strings(ids, ln, "abcd") == ["aab"]


def test_lstm_reference_matches_nn_lstm_and_packing_roundtrip():
    torch.manual_seed(0)
    N, T, I, H = 3, 7, 10, 64
    m = torch.nn.LSTM(I, H, batch_first=True, bidirectional=True)
    p = lstm.pack_lstm_weights(
        [m.weight_ih_l0, m.weight_ih_l0_reverse], [m.weight_hh_l0, m.weight_hh_l0_reverse],
        [m.bias_ih_l0, m.bias_ih_l0_reverse], [m.bias_hh_l0, m.bias_hh_l0_reverse],
    )
    x = torch.randn(N, T, I)
    with torch.no_grad():
        ref, _ = m(x)
        out = lstm.lstm_layer(x, p)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)
    # variable lengths == packed sequence semantics
    lens = torch.tensor([7, 4, 1], dtype=torch.int32)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens.long(), batch_first=True)
    with torch.no_grad():
        ref_p, _ = m(packed)
    ref_pad, _ = torch.nn.utils.rnn.pad_packed_sequence(ref_p, batch_first=True, total_length=T)
    with torch.no_grad():
        out = lstm.lstm_layer(x, p, lens)
    torch.testing.assert_close(out, ref_pad, rtol=1e-4, atol=1e-5)
    # permutation is a bijection and the fragment tensor holds exactly W_hh's values
    perm = lstm.lstm_gate_permutation(H)
    assert sorted(perm.tolist()) == list(range(4 * H))
    assert p.w_frag.shape == (2, 4 * H // 16, H // 32, 64, 8)
    torch.testing.assert_close(
        p.w_frag[0].float().sum(), m.weight_hh_l0.detach().to(torch.bfloat16).float().sum(), rtol=1e-3, atol=1e-2
    )


def test_training_convolution_function_on_cpu_matches_autograd():
    """ops/conv_train.py off the GPU: the data gradient is still formed the kernels' way (180-degree-turned filter,
    gradient samples spread on the padded input grid for strided layers), so its arithmetic is checked here."""
    import torch

    from distributed_llms_amd.ops.conv_train import conv2d_nhwc_train

    g = torch.Generator().manual_seed(0)
    for cin, cout, k, s, p, H, W in ((16, 8, (3, 3), (1, 1), (1, 1), 7, 9), (8, 16, (3, 3), (2, 2), (1, 1), 10, 13),
                                     (8, 8, (3, 3), (2, 1), (1, 1), 9, 12), (16, 16, (3, 1), (1, 1), (0, 0), 3, 10),
                                     (8, 8, (1, 1), (2, 2), (0, 0), 6, 7)):
        x = torch.randn(2, H, W, cin, generator=g).to(torch.bfloat16).requires_grad_(True)
        w = (torch.randn(cout, cin, *k, generator=g) * 0.1).requires_grad_(True)
        y = conv2d_nhwc_train(x, w, s, p)
        gy = torch.randn(y.shape, generator=g).to(torch.bfloat16)
        y.backward(gy)
        x2 = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
        w2 = w.detach().to(torch.bfloat16).float().requires_grad_(True)
        y2 = torch.nn.functional.conv2d(x2, w2, None, s, p)
        y2.backward(gy.float().permute(0, 3, 1, 2))
        rel = lambda a, b: (a.float() - b).abs().max().item() / max(b.abs().max().item(), 1e-6)   # noqa: E731
        assert rel(y, y2.permute(0, 2, 3, 1)) < 1e-2
        assert rel(x.grad, x2.grad.permute(0, 2, 3, 1)) < 1e-2
        assert rel(w.grad, w2.grad) < 1e-4


def test_row_cumsum_cpu_path():
    import torch

    from distributed_llms_amd.ops.scan import row_cumsum_i32

    x = torch.randint(-50, 50, (3, 1030), dtype=torch.int32)
    out = torch.empty_like(x)
    assert torch.equal(row_cumsum_i32(x, out), torch.cumsum(x, 1, dtype=torch.int32))
    assert torch.equal(row_cumsum_i32(x, out, exclusive=True), torch.cumsum(x, 1, dtype=torch.int32) - x)
