"""HIP kernel numerics vs plain PyTorch fp32 references (run on a real MI355X)."""
import numpy as np
import pytest
import torch

from distributed_llms_amd.ops import conv as C
from distributed_llms_amd.ops import ctc, dbpost, lstm, resample

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _rand_bf16(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def _close(a, b, rtol=2e-2, atol=2e-2):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol).float().mean().item()
    assert bad < 1e-3, f"mismatch fraction {bad:.4f}, max err {err.max().item():.4f}"
    # bf16 rounding may push a few elements past the tolerance, but never far past it: a tile edge that reads or
    # writes the wrong element is off by the magnitude of the data, not by a rounding step
    worst = (err / tol).max().item()
    assert worst < 8.0, f"an element is off by {worst:.1f}x the tolerance (max err {err.max().item():.4f})"


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, act
    (2, 37, 53, 8, 32, 3, 2, 1, "relu"),       # stem: Cin=8 (K=72, ragged K tile), odd sizes
    (2, 40, 48, 32, 64, 3, 1, 1, "relu"),
    (1, 33, 29, 64, 64, 3, 1, 1, "none"),
    (2, 24, 20, 64, 128, 3, 2, 1, "hardswish"),
    (1, 16, 24, 128, 256, 3, 1, 1, "relu"),
    (1, 9, 11, 256, 512, 3, 2, 1, "relu"),
    (2, 20, 28, 64, 256, 1, 1, 0, "none"),      # 1x1 lateral
    (2, 20, 28, 64, 128, 1, 2, 0, "none"),      # 1x1 stride-2 shortcut
    (1, 12, 40, 256, 16, 3, 1, 1, "sigmoid"),   # narrow N tile
    (1, 8, 8, 512, 512, 3, 1, 1, "gelu"),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_matches_reference(case):
    B, H, W, Cin, Cout, k, s, p, act = case
    x = _rand_bf16(B, H, W, Cin, seed=1)
    w = _rand_bf16(Cout, k, k, Cin, scale=(2.0 / (k * k * Cin)) ** 0.5, seed=2)
    b = torch.randn(Cout, generator=torch.Generator().manual_seed(3)) * 0.1
    ref = C.conv2d_nhwc_reference(x, w, b, stride=s, padding=p, act=act, out_dtype=torch.float32)
    # cfg 0 = the dispatcher's own choice; the explicit ids force every tile shape onto every problem
    cfgs = [0, 16, 32, 64, 128, 1064, 1128, 2032, 3900, 4128, 8128]    # 4xxx: 128-deep K tile
    if k == 3 and p == 1 and s == 1 and Cin % 64 == 0:
        cfgs += [3064, 3128]          # LDS-patch 3x3 kernel, both channel-tile widths
    for cfg in cfgs:
        out = C.conv2d_nhwc(x.to(DEV), w.to(DEV), b.to(DEV), stride=s, padding=p, act=act, cfg=cfg)
        assert out.shape == ref.shape
        _close(out, ref)


@pytest.mark.parametrize("case", [(2, 48, 200, 8, 2, 64, "relu"), (1, 47, 131, 8, 2, 64, "hardswish"), (3, 17, 64, 8, 2, 32, "relu"),
                                  (1, 96, 1030, 8, 2, 32, "none"), (2, 1, 5, 8, 2, 64, "relu"),
                                  (2, 21, 150, 32, 1, 32, "relu"), (1, 8, 64, 32, 1, 32, "hardswish"), (1, 3, 7, 32, 1, 32, "none")])
def test_conv_stem_kernel_matches_reference_and_generic_path(case):
    """cfg 9008 = the small-Cin 3x3 first-layer kernels (csrc/kernels/conv_stem.hip); cfg 0 must pick them."""
    B, H, W, Cin, st, Cout, act = case
    x = _rand_bf16(B, H, W, Cin, seed=11)
    w = _rand_bf16(Cout, 3, 3, Cin, scale=(2.0 / (9 * Cin)) ** 0.5, seed=12)
    b = torch.randn(Cout, generator=torch.Generator().manual_seed(13)) * 0.1
    ref = C.conv2d_nhwc_reference(x, w, b, stride=st, padding=1, act=act, out_dtype=torch.float32)
    outs = {cfg: C.conv2d_nhwc(x.to(DEV), w.to(DEV), b.to(DEV), stride=st, padding=1, act=act, cfg=cfg) for cfg in (9008, 0, 64)}
    for o in outs.values():
        assert o.shape == ref.shape
        _close(o, ref)
    assert torch.equal(outs[9008], outs[0]), "auto mode must route first layers to the stem kernel"
    # fused concat: write at a channel offset of a wider buffer, neighbours untouched
    Ho, Wo = ref.shape[1:3]
    buf = torch.full((B, Ho, Wo, Cout + 16), 7.0, dtype=torch.bfloat16, device=DEV)
    C.conv2d_nhwc(x.to(DEV), w.to(DEV), b.to(DEV), stride=st, padding=1, act=act, out=buf, out_coff=8, cfg=9008)
    _close(buf[..., 8:8 + Cout], ref)
    assert float((buf[..., :8] - 7).abs().max()) == 0 and float((buf[..., 8 + Cout:] - 7).abs().max()) == 0
    with pytest.raises(Exception):      # a shape the kernels do not serve must be refused, not mis-run
        C.conv2d_nhwc(x.to(DEV), w.to(DEV), b.to(DEV), stride=3 - st, padding=1, act=act, cfg=9008)


@pytest.mark.parametrize("shape", [(2, 12, 70, 64, 128, (1, 1)), (3, 12, 45, 128, 256, (2, 1)), (2, 6, 130, 256, 64, (2, 1)),
                                   (1, 3, 200, 64, 512, (1, 1)), (2, 37, 29, 64, 64, (1, 1)), (1, 5, 17, 192, 96, (2, 1))])
@pytest.mark.parametrize("cfg", [3064, 3128])
def test_conv3x3_patch_kernel_shapes_strides_and_fusions(shape, cfg):
    """Every tile shape of the patch kernel (8x16 / 4x32 / 2x64 / 1x128 by output height), ragged edges,
    stride (2,1), fused residual + ReLU, and the replicate-into-concat store."""
    B, H, W, Cin, Cout, stride = shape
    x = _rand_bf16(B, H, W, Cin, seed=21)
    w = _rand_bf16(Cout, 3, 3, Cin, scale=(2.0 / (9 * Cin)) ** 0.5, seed=22)
    b = torch.randn(Cout, generator=torch.Generator().manual_seed(23)) * 0.1
    Ho, Wo = C.out_hw(H, W, 3, 3, stride[0], stride[1], 1, 1)
    res = _rand_bf16(B, Ho, Wo, Cout, seed=24)
    ref = C.conv2d_nhwc_reference(x, w, b, stride=stride, padding=1, act="relu", res=res, res_mode=C.RES_SAME,
                                  out_dtype=torch.float32)
    out = C.conv2d_nhwc(x.to(DEV), w.to(DEV), b.to(DEV), stride=stride, padding=1, act="relu", res=res.to(DEV),
                        res_mode=C.RES_SAME, cfg=cfg)
    _close(out, ref)
    gen = C.conv2d_nhwc(x.to(DEV), w.to(DEV), b.to(DEV), stride=stride, padding=1, act="relu", res=res.to(DEV),
                        res_mode=C.RES_SAME, cfg=3900)
    _close(out, gen.float().cpu(), rtol=1e-2, atol=1e-2)      # same maths as the generic kernel
    if Cout % 16 == 0:
        ref2 = C.conv2d_nhwc_reference(x, w, None, stride=stride, padding=1, out_mode=C.OUT_REPLICATE, out_f=2,
                                       out_dtype=torch.float32)
        cat = torch.zeros(B, 2 * Ho, 2 * Wo, Cout + 32, dtype=torch.bfloat16, device=DEV)
        C.conv2d_nhwc(x.to(DEV), w.to(DEV), None, stride=stride, padding=1, out_mode=C.OUT_REPLICATE, out_f=2, out=cat,
                      out_coff=16, cfg=cfg)
        _close(cat[..., 16 : 16 + Cout], ref2)
        assert cat[..., :16].abs().max().item() == 0 and cat[..., 16 + Cout :].abs().max().item() == 0


@pytest.mark.parametrize("case", [(2, 40, 48, 64, 128, 3, 1, 1), (1, 33, 29, 64, 64, 3, 2, 1), (2, 20, 28, 64, 256, 1, 1, 0),
                                  (1, 16, 24, 128, 256, 3, 1, 1)])
def test_conv_persistent_kernel_walks_several_tiles_per_workgroup(case, monkeypatch):
    """cfg 8128 with the grid forced down to 5 workgroups: every workgroup processes many output tiles in turn
    (tap table reused, LDS output image and operand buffers recycled between tiles)."""
    B, H, W, Cin, Cout, k, s, p = case
    monkeypatch.setenv("DLAMD_CONV_PERSIST_GRID", "5")
    x = _rand_bf16(B, H, W, Cin, seed=31)
    w = _rand_bf16(Cout, k, k, Cin, scale=(2.0 / (k * k * Cin)) ** 0.5, seed=32)
    b = torch.randn(Cout, generator=torch.Generator().manual_seed(33)) * 0.1
    ref = C.conv2d_nhwc_reference(x, w, b, stride=s, padding=p, act="relu", out_dtype=torch.float32)
    out = C.conv2d_nhwc(x.to(DEV), w.to(DEV), b.to(DEV), stride=s, padding=p, act="relu", cfg=8128)
    _close(out, ref)
    res = _rand_bf16(*out.shape, seed=34)
    ref2 = C.conv2d_nhwc_reference(x, w, b, stride=s, padding=p, act="relu", res=res, res_mode=C.RES_SAME, out_dtype=torch.float32)
    out2 = C.conv2d_nhwc(x.to(DEV), w.to(DEV), b.to(DEV), stride=s, padding=p, act="relu", res=res.to(DEV), res_mode=C.RES_SAME,
                         cfg=8128, out_dtype=torch.float32)           # f32 output: the direct (non-LDS-image) store path
    _close(out2, ref2)


def test_conv_asymmetric_kernel_and_stride():
    x = _rand_bf16(3, 6, 50, 256, seed=4)
    w = _rand_bf16(256, 3, 3, 256, scale=0.03, seed=5)
    ref = C.conv2d_nhwc_reference(x, w, None, stride=(2, 1), padding=(1, 1), act="relu", out_dtype=torch.float32)
    out = C.conv2d_nhwc(x.to(DEV), w.to(DEV), None, stride=(2, 1), padding=(1, 1), act="relu")
    _close(out, ref)
    w2 = _rand_bf16(128, 3, 1, 256, scale=0.04, seed=6)
    ref = C.conv2d_nhwc_reference(x[:, :3], w2, None, stride=1, padding=0, out_dtype=torch.float32)
    out = C.conv2d_nhwc(x[:, :3].contiguous().to(DEV), w2.to(DEV), None, stride=1, padding=0)
    assert out.shape == (3, 1, 50, 128)
    _close(out, ref)


def test_conv_residual_same_and_upsampled():
    x = _rand_bf16(2, 20, 28, 64, seed=7)
    w = _rand_bf16(64, 3, 3, 64, scale=0.06, seed=8)
    r = _rand_bf16(2, 20, 28, 64, seed=9)
    ref = C.conv2d_nhwc_reference(x, w, None, stride=1, padding=1, act="relu", res=r, res_mode=C.RES_SAME,
                                  out_dtype=torch.float32)
    out = C.conv2d_nhwc(x.to(DEV), w.to(DEV), None, stride=1, padding=1, act="relu", res=r.to(DEV),
                        res_mode=C.RES_SAME)
    _close(out, ref)
    w1 = _rand_bf16(64, 1, 1, 64, scale=0.12, seed=10)
    r2 = _rand_bf16(2, 10, 14, 64, seed=11)
    ref = C.conv2d_nhwc_reference(x, w1, None, res=r2, res_mode=C.RES_UP2, out_dtype=torch.float32)
    out = C.conv2d_nhwc(x.to(DEV), w1.to(DEV), None, res=r2.to(DEV), res_mode=C.RES_UP2)
    _close(out, ref)


def test_conv_lateral_streaming_kernel_residuals_slices_and_tails():
    """Shallow-K 1x1 into 256 channels (cfg 9100): weights resident in LDS, pixels streamed through the waves."""
    # 64 -> 256 with the nearest-2x residual, odd map sizes (partial last tile, coarse map rounded up), bias
    x = _rand_bf16(2, 37, 45, 64, seed=40)
    w = _rand_bf16(256, 1, 1, 64, scale=0.12, seed=41)
    b = torch.randn(256) * 0.1
    r2 = _rand_bf16(2, 19, 23, 256, seed=42)
    ref = C.conv2d_nhwc_reference(x, w, b, res=r2, res_mode=C.RES_UP2, out_dtype=torch.float32)
    out = C.conv2d_nhwc(x.to(DEV), w.to(DEV), b.to(DEV), res=r2.to(DEV), res_mode=C.RES_UP2, cfg=9100)
    _close(out, ref)
    # 128 -> 256, same-resolution residual, activation, input a channel slice of a wider tensor, output into a
    # concat buffer at a channel offset
    xw = _rand_bf16(3, 9, 130, 160, seed=43).to(DEV)
    xs = xw[..., 32:160]
    w2 = _rand_bf16(256, 1, 1, 128, scale=0.09, seed=44)
    r1 = _rand_bf16(3, 9, 130, 256, seed=45)
    ref = C.conv2d_nhwc_reference(xs.cpu(), w2, None, act="relu", res=r1, res_mode=C.RES_SAME, out_dtype=torch.float32)
    cat = torch.zeros(3, 9, 130, 328, dtype=torch.bfloat16, device=DEV)
    C.conv2d_nhwc(xs, w2.to(DEV), None, act="relu", res=r1.to(DEV), res_mode=C.RES_SAME, out=cat, out_coff=64, cfg=9100)
    _close(cat[..., 64:320], ref)
    assert cat[..., :64].abs().max().item() == 0 and cat[..., 320:].abs().max().item() == 0
    # large enough for several tiles per workgroup; the automatic choice takes the same kernel from 32 k pixels up
    x3 = _rand_bf16(1, 300, 400, 64, seed=46).to(DEV)
    r3 = _rand_bf16(1, 150, 200, 256, seed=47).to(DEV)
    forced = C.conv2d_nhwc(x3, w.to(DEV), b.to(DEV), res=r3, res_mode=C.RES_UP2, cfg=9100)
    auto = C.conv2d_nhwc(x3, w.to(DEV), b.to(DEV), res=r3, res_mode=C.RES_UP2)
    generic = C.conv2d_nhwc(x3, w.to(DEV), b.to(DEV), res=r3, res_mode=C.RES_UP2, cfg=3900)
    assert torch.equal(forced, auto)
    _close(forced, generic.float().cpu())
    with pytest.raises(Exception):
        C.conv2d_nhwc(x3, _rand_bf16(128, 1, 1, 64, seed=48).to(DEV), None, cfg=9100)


def test_conv_store_modes():
    x = _rand_bf16(2, 10, 14, 64, seed=12)
    # pixel shuffle (deconv 2x2 s2): Cout = 4 * 64
    w = _rand_bf16(256, 1, 1, 64, scale=0.12, seed=13)
    b = torch.randn(256) * 0.1
    ref = C.conv2d_nhwc_reference(x, w, b, act="relu", out_mode=C.OUT_PIXEL_SHUFFLE, out_dtype=torch.float32)
    out = C.conv2d_nhwc(x.to(DEV), w.to(DEV), b.to(DEV), act="relu", out_mode=C.OUT_PIXEL_SHUFFLE)
    assert out.shape == (2, 20, 28, 64)
    _close(out, ref)
    # replicate x4 into a concat buffer at channel offset 64
    w3 = _rand_bf16(64, 3, 3, 64, scale=0.06, seed=14)
    ref = C.conv2d_nhwc_reference(x, w3, None, padding=1, out_mode=C.OUT_REPLICATE, out_f=4, out_dtype=torch.float32)
    cat = torch.zeros(2, 40, 56, 256, dtype=torch.bfloat16, device=DEV)
    C.conv2d_nhwc(x.to(DEV), w3.to(DEV), None, padding=1, out_mode=C.OUT_REPLICATE, out_f=4, out=cat, out_coff=64)
    _close(cat[..., 64:128], ref)
    assert cat[..., :64].abs().max().item() == 0 and cat[..., 128:].abs().max().item() == 0
    # probability map + bitmask
    w4 = _rand_bf16(4, 1, 1, 64, scale=0.3, seed=15)
    b4 = torch.randn(4) * 0.1
    pref, mref = C.conv2d_nhwc_reference(x, w4, b4, act="sigmoid", out_mode=C.OUT_PROB, thr=0.3,
                                         out_dtype=torch.float32)
    p, m = C.conv2d_nhwc(x.to(DEV), w4.to(DEV), b4.to(DEV), act="sigmoid", out_mode=C.OUT_PROB, thr=0.3)
    assert p.shape == (2, 20, 28) and m.shape == (2, 20, 28)
    _close(p, pref, atol=1e-2)
    agree = ((m.cpu() > 0) == (mref > 0)) | ((pref - 0.3).abs() < 1e-2)
    assert bool(agree.all())


def test_linear_f32_out_and_strided_input():
    x = _rand_bf16(5, 37, 512, seed=16)
    w = _rand_bf16(96, 512, scale=0.05, seed=17)
    b = torch.randn(96)
    ref = x.float() @ w.float().t() + b
    out = C.linear_bf16(x.to(DEV), w.to(DEV), b.to(DEV), out_dtype=torch.float32)
    assert out.dtype == torch.float32
    _close(out, ref)
    # channel-sliced input (pixel stride 128, Cin 64)
    big = _rand_bf16(2, 12, 12, 128, seed=18).to(DEV)
    xs = big[..., 64:]
    w3 = _rand_bf16(32, 3, 3, 64, scale=0.06, seed=19)
    ref = C.conv2d_nhwc_reference(xs.cpu().contiguous(), w3, None, padding=1, out_dtype=torch.float32)
    out = C.conv2d_nhwc(xs, w3.to(DEV), None, padding=1)
    _close(out, ref)


def test_resample_pages_and_crops():
    g = torch.Generator().manual_seed(0)
    page = torch.randint(0, 256, (351, 248), dtype=torch.uint8, generator=g)
    rgb = torch.randint(0, 256, (120, 200, 3), dtype=torch.uint8, generator=g)
    regs = [
        resample.Region(page, (0, 0, 248, 351), 128, 90),            # shrink ~2.7x (box filter)
        resample.Region(rgb, (0, 0, 200, 120), 96, 160),             # shrink 1.25x
        resample.Region(page, (10.5, 20.25, 200.0, 60.0), 48, 128),  # crop, mixed
        resample.Region(rgb, (5, 5, 45, 25), 48, 96),                # enlarge (bilinear)
        resample.Region(page, (30, 40, 70, 300), 48, 120, rot=1),    # tall box rotated
    ]
    ref = resample.resample_regions_reference(regs, 128, 160)
    dregs = [resample.Region(r.image.to(DEV), r.box, r.out_h, r.out_w, r.rot) for r in regs]
    out = resample.resample_regions(dregs, 128, 160)
    assert out.shape == (5, 128, 160, 8)
    _close(out, ref, rtol=1e-2, atol=2e-2)
    assert out[..., 3:].abs().max().item() == 0


def test_db_postprocess_matches_scipy():
    g = torch.Generator().manual_seed(0)
    B, H, W = 3, 160, 192
    prob = torch.rand(B, H, W, generator=g) * 0.25
    # paint text-line-like blobs plus diagonal touching blobs (8-connectivity) and noise
    for b in range(B):
        for k in range(6):
            y0, x0 = 10 + 22 * k, 8 + 5 * k
            prob[b, y0 : y0 + 9, x0 : x0 + 120 - 10 * k] = 0.7 + 0.25 * torch.rand(9, 120 - 10 * k, generator=g)
        prob[b, 150:154, 10:14] = 0.9
        prob[b, 154:158, 14:18] = 0.9   # touches diagonally
        prob[b, 100:103, 170:190] = 0.45  # low score component (filtered by box_thresh)
    prob = prob.to(torch.bfloat16)
    mask = (prob.float() > 0.3).to(torch.uint8)
    cfg = dbpost.DBPostConfig()
    scale = torch.tensor([[2.5, 2.5], [1.0, 2.0], [3.0, 1.5]], dtype=torch.float32)
    valid = torch.tensor([[W, H], [W, H], [W - 10, H - 4]], dtype=torch.int32)
    ref = dbpost.db_postprocess(prob, mask, cfg, scale, valid)
    out = dbpost.db_postprocess(prob.to(DEV), mask.to(DEV), cfg, scale.to(DEV), valid.to(DEV))
    for r, o in zip(ref, out):
        assert r.shape == o.shape and r.shape[0] >= 7
        np.testing.assert_allclose(o[:, :4], r[:, :4], rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(o[:, 4], r[:, 4], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(o[:, 5], r[:, 5])


def _slanted_prob_map(H=320, W=480, seed=3):
    """Probability map with filled rectangles turned by known angles (what the detector emits for a skewed scan)."""
    from PIL import Image, ImageDraw

    from distributed_llms_amd.train.page_trainer import rect_corners

    rects = np.array([[240, 40, 400, 18, np.deg2rad(-5.0)], [230, 110, 380, 16, np.deg2rad(3.0)],
                      [250, 170, 300, 20, np.deg2rad(1.0)], [240, 240, 420, 14, np.deg2rad(8.0)],
                      [100, 295, 150, 16, 0.0]], dtype=np.float64)
    im = Image.new("L", (W, H), 0)
    d = ImageDraw.Draw(im)
    for q in rect_corners(rects):
        d.polygon([(float(x), float(y)) for x, y in q], fill=1)
    m = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy())
    g = torch.Generator().manual_seed(seed)
    prob = torch.rand(H, W, generator=g) * 0.2
    prob[m > 0] = 0.75 + 0.2 * torch.rand(int(m.sum()), generator=g)
    return prob.to(torch.bfloat16)[None], rects


def test_db_postprocess_oriented_boxes_for_slanted_lines():
    prob, rects = _slanted_prob_map()
    mask = (prob.float() > 0.3).to(torch.uint8)
    cfg = dbpost.DBPostConfig(unclip_ratio=1.5)
    scale = torch.tensor([[2.5, 2.6]], dtype=torch.float32)
    ref = dbpost.db_postprocess_reference(prob, mask, cfg, scale)[0]
    ws = dbpost.DBPostWorkspace(1, prob.shape[1], prob.shape[2], cfg, torch.device(DEV))
    bx, nb = dbpost.db_postprocess_gpu(prob.to(DEV), mask.to(DEV), ws, scale.to(DEV))
    out = bx[0, : int(nb[0])].cpu().numpy()
    assert out.shape == ref.shape == (5, 8)
    o, r = out[np.argsort(out[:, 1])], ref[np.argsort(ref[:, 1])]
    np.testing.assert_allclose(o[:, :4], r[:, :4], rtol=1e-5, atol=2e-2)
    np.testing.assert_allclose(o[:, 6], r[:, 6], atol=2e-5)
    np.testing.assert_allclose(o[:, 5], r[:, 5])
    # geometry: angle within 0.3 degrees of the truth, size = truth + unclip within ~2 px (det pixels), centre within 1
    for row, t in zip(o, rects[np.argsort(rects[:, 1])]):
        d = t[2] * t[3] * 1.5 / (2 * (t[2] + t[3]))
        want_angle = np.arctan2(np.sin(t[4]) * 2.6, np.cos(t[4]) * 2.5)
        assert abs(row[6] - want_angle) < np.deg2rad(0.3), (row, t)
        L, T = (row[2] - row[0]), (row[3] - row[1])
        assert abs(L / 2.5 - (t[2] + 2 * d)) < 4.5 and abs(T / 2.6 - (t[3] + 2 * d)) < 3.5, (L, T, t, d)
        assert abs(0.5 * (row[0] + row[2]) / 2.5 - t[0]) < 1.5 and abs(0.5 * (row[1] + row[3]) / 2.6 - t[1]) < 1.5


def test_db_postprocess_page_skew_consensus_for_short_lines():
    """Every wide component votes for the page's skew; short lines, whose own axis is noisy, take the consensus, and
    a long line that disagrees with it by more than 3 degrees keeps its own angle."""
    from PIL import Image, ImageDraw

    from distributed_llms_amd.train.page_trainer import rect_corners

    H, W, skew = 400, 480, np.deg2rad(2.0)
    rows = [[240, 30 + 28 * i, 380 - 10 * (i % 3), 12, skew] for i in range(9)]          # body lines
    rows += [[90, 300, 52, 12, skew], [300, 300, 40, 11, skew], [200, 330, 60, 12, skew]]   # short last lines
    rows += [[240, 372, 400, 12, np.deg2rad(-6.0)]]                                          # a turned caption
    rects = np.array(rows, dtype=np.float64)
    im = Image.new("L", (W, H), 0)
    d = ImageDraw.Draw(im)
    for q in rect_corners(rects):
        d.polygon([(float(x), float(y)) for x, y in q], fill=1)
    m = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy())
    g = torch.Generator().manual_seed(11)
    prob = torch.rand(H, W, generator=g) * 0.2
    prob[m > 0] = 0.75 + 0.2 * torch.rand(int(m.sum()), generator=g)
    prob = prob.to(torch.bfloat16)[None]
    mask = (prob.float() > 0.3).to(torch.uint8)
    cfg = dbpost.DBPostConfig(unclip_ratio=1.5)
    ref = dbpost.db_postprocess_reference(prob, mask, cfg, None)[0]
    ws = dbpost.DBPostWorkspace(1, H, W, cfg, torch.device(DEV))
    bx, nb = dbpost.db_postprocess_gpu(prob.to(DEV), mask.to(DEV), ws, None)
    out = bx[0, : int(nb[0])].cpu().numpy()
    assert out.shape == ref.shape == (len(rows), 8)
    key = lambda a: np.lexsort((a[:, 0], np.round(a[:, 1])))      # noqa: E731
    o, r = out[key(out)], ref[key(ref)]
    np.testing.assert_allclose(o[:, :4], r[:, :4], rtol=1e-5, atol=2e-2)
    np.testing.assert_allclose(o[:, 6], r[:, 6], atol=2e-5)
    page = float(ws.page_angle[0])
    assert abs(page - skew) < np.deg2rad(0.15), np.rad2deg(page)
    cy = 0.5 * (out[:, 1] + out[:, 3])
    body = out[cy < 350]
    assert len(body) == 12 and np.all(body[:, 6] == body[0, 6]) and abs(body[0, 6] - page) < 1e-6    # one angle for the page
    caption = out[cy >= 350]
    assert len(caption) == 1 and abs(caption[0, 6] - np.deg2rad(-6.0)) < np.deg2rad(0.3)


def test_resample_affine_crop_of_slanted_lines():
    g = torch.Generator().manual_seed(5)
    page = torch.randint(0, 256, (400, 600), dtype=torch.uint8, generator=g)
    # smooth it a little so that sub-pixel differences between float32 paths stay small
    page = torch.nn.functional.avg_pool2d(page[None, None].float(), 3, 1, 1)[0, 0].to(torch.uint8)
    rgb = torch.randint(0, 256, (120, 200, 3), dtype=torch.uint8, generator=g)
    regs = [
        resample.Region(page, (50.0, 100.0, 550.0, 160.0), 48, 400, 0, 0.06),      # shrink 1.25x, 3.4 degrees
        resample.Region(page, (50.0, 200.0, 550.0, 330.0), 48, 184, 0, -0.09),     # shrink 2.7x: 3 x 3 taps
        resample.Region(page, (100.0, 50.0, 300.0, 70.0), 48, 480, 0, 0.03),       # enlarge: one tap
        resample.Region(rgb, (20.0, 30.0, 180.0, 70.0), 32, 128, 0, -0.05),
        resample.Region(page, (300.0, 40.0, 340.0, 380.0), 48, 408, 1, 0.04),      # tall box, turned 90 degrees too
        resample.Region(page, (-20.0, -10.0, 200.0, 30.0), 48, 264, 0, 0.08),      # sticks out of the image: clamped
    ]
    ref = resample.resample_regions_reference(regs, 48, 480, resample.HALF_MEAN, resample.HALF_STD)
    dregs = [resample.Region(r.image.to(DEV), r.box, r.out_h, r.out_w, r.rot, r.angle) for r in regs]
    out = resample.resample_regions(dregs, 48, 480, resample.HALF_MEAN, resample.HALF_STD)
    _close(out, ref, rtol=1e-2, atol=3e-2)
    # and the angle does something: the same box without it differs
    flat = resample.resample_regions([resample.Region(page.to(DEV), regs[0].box, 48, 400)], 48, 480,
                                     resample.HALF_MEAN, resample.HALF_STD)
    assert (flat[0].float() - out[0].float()).abs().mean().item() > 0.02


def test_resample_fused_with_first_conv_matches_the_two_step_path():
    """resample + 3x3 stride-2 conv in one kernel (gray and RGB sources, box filter / bilinear / affine / rotated
    regions, odd sizes, both channel counts) against the resample kernel followed by the conv kernel."""
    g = torch.Generator().manual_seed(9)
    page = torch.randint(0, 256, (400, 600), dtype=torch.uint8, generator=g)
    page = torch.nn.functional.avg_pool2d(page[None, None].float(), 3, 1, 1)[0, 0].to(torch.uint8)
    rgb = torch.randint(0, 256, (150, 260, 3), dtype=torch.uint8, generator=g)
    regs = [
        resample.Region(page, (0.0, 0.0, 600.0, 400.0), 150, 230),                 # shrink 2.6x
        resample.Region(page, (50.0, 100.0, 550.0, 160.0), 48, 400, 0, 0.06),      # slanted line
        resample.Region(page, (100.0, 50.0, 300.0, 70.0), 48, 470),                # enlarge
        resample.Region(rgb, (20.0, 30.0, 250.0, 140.0), 90, 190),                 # RGB source
        resample.Region(page, (300.0, 40.0, 340.0, 380.0), 48, 408, 1),            # turned by 90 degrees
        resample.Region(page, (10.0, 10.0, 590.0, 390.0), 151, 471),               # fills the bucket, shrink 1.2 / 2.5
    ]
    dregs = [resample.Region(r.image.to(DEV), r.box, r.out_h, r.out_w, r.rot, r.angle) for r in regs]
    items = torch.from_numpy(resample.pack_items(dregs).view(np.uint8).copy()).to(DEV)
    for (Hd, Wd), cout, mean, std, act in (((151, 471), 32, resample.IMAGENET_MEAN, resample.IMAGENET_STD, "relu"),
                                           ((152, 480), 64, resample.HALF_MEAN, resample.HALF_STD, "hardswish")):
        w = torch.zeros(cout, 3, 3, 8)
        w[..., :3] = torch.randn(cout, 3, 3, 3, generator=g) * 0.3
        w = w.to(torch.bfloat16).to(DEV)
        b = (torch.randn(cout, generator=g) * 0.2).to(DEV)
        x = resample.resample_items_gpu(items, len(regs), Hd, Wd, mean, std)
        want = C.conv2d_nhwc(x, w, b, stride=2, padding=1, act=act)
        got = resample.resample_stem_items_gpu(items, len(regs), Hd, Wd, mean, std, w, b, act)
        assert got.shape == want.shape == (len(regs), (Hd - 1) // 2 + 1, (Wd - 1) // 2 + 1, cout)
        _close(got, want, rtol=1e-2, atol=1e-2)
        ref = C.conv2d_nhwc_reference(x.cpu(), w.cpu(), b.cpu(), stride=2, padding=1, act=act, out_dtype=torch.float32)
        _close(got, ref, rtol=1e-2, atol=2e-2)


def test_db_postprocess_random_noise_components():
    g = torch.Generator().manual_seed(1)
    prob = torch.rand(2, 96, 128, generator=g).to(torch.bfloat16)
    mask = (prob.float() > 0.55).to(torch.uint8)
    cfg = dbpost.DBPostConfig(box_thresh=0.0, min_size=1, max_components=8192, max_boxes=8192)
    ref = dbpost.db_postprocess(prob, mask, cfg)
    out = dbpost.db_postprocess(prob.to(DEV), mask.to(DEV), cfg)
    for r, o in zip(ref, out):
        assert r.shape == o.shape
        # same multiset of boxes (ordering of equal-key boxes may differ)
        rs = r[np.lexsort(r[:, :4].T[::-1])]
        os_ = o[np.lexsort(o[:, :4].T[::-1])]
        np.testing.assert_allclose(os_[:, :4], rs[:, :4], atol=1e-3)
        np.testing.assert_allclose(os_[:, 5], rs[:, 5])


@pytest.mark.parametrize("W,thr", [(100, 0.55), (100, 0.42), (72, 0.5), (64, 0.45), (200, 0.47)])
def test_db_postprocess_row_wrap_and_segment_seams(W, thr):
    """Widths that are not a multiple of the 64-pixel wave segment: runs must not continue across a row start, runs
    that cross a segment seam must be joined, and dense masks build large components out of many run overlaps."""
    g = torch.Generator().manual_seed(W)
    H = 57
    prob = torch.rand(2, H, W, generator=g)
    prob[0, 10, W - 7:] = 0.99          # ends at the right border ...
    prob[0, 11, :5] = 0.99              # ... and the next row starts at the left border: different components
    prob[0, 9, :] = 0.0
    prob[0, 12, :] = 0.0
    prob[0, 11, 5:12] = 0.0
    prob[0, 10, W - 14:W - 7] = 0.0
    prob[1, 30, :] = 0.99               # a full-width run crosses every seam
    prob = prob.to(torch.bfloat16)
    mask = (prob.float() > thr).to(torch.uint8)
    cfg = dbpost.DBPostConfig(box_thresh=0.0, min_size=1, max_components=8192, max_boxes=8192)
    ref = dbpost.db_postprocess(prob, mask, cfg)
    out = dbpost.db_postprocess(prob.to(DEV), mask.to(DEV), cfg)
    for r, o in zip(ref, out):
        assert r.shape == o.shape
        rs = r[np.lexsort(r[:, :4].T[::-1])]
        os_ = o[np.lexsort(o[:, :4].T[::-1])]
        np.testing.assert_allclose(os_[:, :4], rs[:, :4], atol=1e-3)
        np.testing.assert_allclose(os_[:, 5], rs[:, 5])


@pytest.mark.parametrize("V,dtype", [(97, torch.bfloat16), (6624, torch.bfloat16), (50, torch.float32)])
def test_ctc_greedy_decode(V, dtype):
    g = torch.Generator().manual_seed(0)
    N, T = 7, 83
    logits = (torch.randn(N, T, V, generator=g) * 3).to(dtype)
    # force repeats and blanks
    logits[:, 10:14, 5] += 30
    logits[:, 20:25, 0] += 30
    lens = torch.tensor([83, 80, 1, 0, 40, 83, 13], dtype=torch.int32)
    ref_ids, ref_len, ref_conf = ctc.ctc_greedy_reference(logits, lens)
    ids, ln, conf = ctc.ctc_greedy_decode(logits.to(DEV), lens.to(DEV))
    assert torch.equal(ln.cpu(), ref_len)
    assert torch.equal(ids.cpu(), ref_ids)
    torch.testing.assert_close(conf.cpu(), ref_conf, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("H,ndir", [(256, 2), (128, 2), (64, 1)])
def test_lstm_matches_reference(H, ndir):
    torch.manual_seed(0)
    N, T, I = 21, 19, 96
    w_ih = [torch.randn(4 * H, I) * 0.1 for _ in range(ndir)]
    w_hh = [torch.randn(4 * H, H) * 0.06 for _ in range(ndir)]
    b_ih = [torch.randn(4 * H) * 0.1 for _ in range(ndir)]
    b_hh = [torch.randn(4 * H) * 0.1 for _ in range(ndir)]
    # round weights to bf16 so the oracle sees the same parameters
    w_ih = [w.to(torch.bfloat16).float() for w in w_ih]
    w_hh = [w.to(torch.bfloat16).float() for w in w_hh]
    p = lstm.pack_lstm_weights(w_ih, w_hh, b_ih, b_hh)
    x = (torch.randn(N, T, I) * 0.8).to(torch.bfloat16)
    lens = torch.randint(1, T + 1, (N,), dtype=torch.int32)
    lens[0] = T
    ref = lstm.lstm_reference(x, p, lens)
    out = lstm.lstm_layer(x.to(DEV), p.to(DEV), lens.to(DEV))
    assert out.shape == ref.shape
    _close(out, ref, rtol=3e-2, atol=3e-2)
    # matches torch.nn.LSTM on full-length sequences too
    m = torch.nn.LSTM(I, H, batch_first=True, bidirectional=(ndir == 2))
    with torch.no_grad():
        for d in range(ndir):
            sfx = "_reverse" if d == 1 else ""
            getattr(m, "weight_ih_l0" + sfx).copy_(w_ih[d])
            getattr(m, "weight_hh_l0" + sfx).copy_(w_hh[d])
            getattr(m, "bias_ih_l0" + sfx).copy_(b_ih[d])
            getattr(m, "bias_hh_l0" + sfx).copy_(b_hh[d])
        ref2, _ = m(x.float())
    out2 = lstm.lstm_layer(x.to(DEV), p.to(DEV), None)
    _close(out2, ref2, rtol=3e-2, atol=3e-2)
