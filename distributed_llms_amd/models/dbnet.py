"""DBNet-style text detector.

ResNet-18-shaped backbone + FPN neck + DB probability head, re-designed for
the MI355X inference path:

* activations are NHWC bf16 end to end; every conv is one MFMA implicit-GEMM
  launch with BatchNorm folded and bias/ReLU/residual fused in the epilogue;
* FPN top-down "nearest-2x upsample + add" is fused into the lateral 1x1
  conv's residual read (no upsampled tensor is ever materialised);
* FPN "upsample x2/x4/x8 + concat" is fused into the smooth convs' store: each
  smooth conv writes its 64 channels, replicated fxf, straight into its channel
  slice of the 1/4-scale concat buffer;
* both 2x2/s2 transposed convs are GEMMs with a pixel-shuffle store; the last one
  applies sigmoid and emits the probability map and the >thr bitmask together;
* a second, block-level head on the same 1/4-scale features (one 3x3 conv) marks
  figure regions and display formulas: the engine turns both maps into boxes
  with the same connected-component kernels, keeps figures out of the text flow
  and hands their crops to the image writer, and sends formula crops to the
  formula recogniser.  The reference gets its block layout from a separate YOLO
  layout model and its formula boxes from another (services/ocr_service/
  download_models_hf.py:38-39).

In the reference the detector is the PaddleOCR-torch DB model downloaded by
services/ocr_service/download_models_hf.py:41 and executed on CPU inside
magic-pdf (ocr_service.py:215); the architecture here is a from-scratch
equivalent with random-init weights (no checkpoint is reachable offline).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops import conv as C
from .layers import BasicBlock, ConvBNAct, Deconv2x2, prepare_all


class DBNet(nn.Module):
    IN_CH_PACKED = 8  # NHWC input pixels are 8 x bf16 = 16 bytes (3 real channels)

    def __init__(self, widths=(64, 128, 256, 512), stem=32, inner=256, bin_thresh: float = 0.3):
        super().__init__()
        w1, w2, w3, w4 = widths
        self.bin_thresh = bin_thresh
        self.stem1 = ConvBNAct(3, stem, 3, 2)
        self.stem2 = ConvBNAct(stem, stem, 3, 1)
        self.stem3 = ConvBNAct(stem, w1, 3, 2)
        self.layer1 = nn.ModuleList([BasicBlock(w1, w1), BasicBlock(w1, w1)])
        self.layer2 = nn.ModuleList([BasicBlock(w1, w2, 2), BasicBlock(w2, w2)])
        self.layer3 = nn.ModuleList([BasicBlock(w2, w3, 2), BasicBlock(w3, w3)])
        self.layer4 = nn.ModuleList([BasicBlock(w3, w4, 2), BasicBlock(w4, w4)])
        self.in5 = ConvBNAct(w4, inner, 1, 1, 0, act="none", bn=False)
        self.in4 = ConvBNAct(w3, inner, 1, 1, 0, act="none", bn=False)
        self.in3 = ConvBNAct(w2, inner, 1, 1, 0, act="none", bn=False)
        self.in2 = ConvBNAct(w1, inner, 1, 1, 0, act="none", bn=False)
        q = inner // 4
        self.p5 = ConvBNAct(inner, q, 3, 1, act="none", bn=False)
        self.p4 = ConvBNAct(inner, q, 3, 1, act="none", bn=False)
        self.p3 = ConvBNAct(inner, q, 3, 1, act="none", bn=False)
        self.p2 = ConvBNAct(inner, q, 3, 1, act="none", bn=False)
        self.head1 = ConvBNAct(inner, q, 3, 1, act="relu")
        self.head2 = Deconv2x2(q, q, act="relu", bn=True)
        self.head3 = Deconv2x2(q, 1, act="sigmoid", bn=False)
        # block head at 1/4 scale: channel 0 figure logit, channel 1 display-formula logit (6 spare channels)
        self.fig_head = ConvBNAct(q, 8, 3, 1, act="none", bn=False)
        self.fuse_head_tail = True      # device inference: ops/conv.db_head_fused when q == 64
        self._head_tail = None
        self.inner, self.q = inner, q
        # prior-probability init (focal-loss style): near-zero last-layer weights + bias -2.2 make an untrained
        # network predict p ~ 0.1 everywhere, i.e. an empty mask - no hallucinated boxes before training
        with torch.no_grad():
            self.head3.deconv.weight.mul_(1e-2)
            self.head3.deconv.bias.fill_(-2.2)
            self.fig_head.conv.weight.mul_(1e-2)
            self.fig_head.conv.bias.fill_(-3.0)

    # ------------------------------------------------------------------ training path (NCHW)
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B,3,H,W] normalised -> probability map [B,H,W] (H, W multiples of 32)."""
        return self.forward_all(x)[0]

    def forward_all(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (text probability [B,H,W], block logits [B,2,H/4,W/4]: channel 0 figure, channel 1 formula)."""
        h = self.forward_features(x)
        return self.head3(self.head2(h))[:, 0], self.fig_head(h)[:, :2]

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        """-> the quarter-scale feature map [B,q,H/4,W/4] both heads read."""
        y = self.stem3(self.stem2(self.stem1(x)))
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                y = blk(y)
            feats.append(y)
        c2, c3, c4, c5 = feats
        i5 = self.in5(c5)
        o4 = self.in4(c4) + F.interpolate(i5, scale_factor=2, mode="nearest")
        o3 = self.in3(c3) + F.interpolate(o4, scale_factor=2, mode="nearest")
        o2 = self.in2(c2) + F.interpolate(o3, scale_factor=2, mode="nearest")
        p5 = F.interpolate(self.p5(i5), scale_factor=8, mode="nearest")
        p4 = F.interpolate(self.p4(o4), scale_factor=4, mode="nearest")
        p3 = F.interpolate(self.p3(o3), scale_factor=2, mode="nearest")
        p2 = self.p2(o2)
        fuse = torch.cat([p5, p4, p3, p2], dim=1)
        return self.head1(fuse)

    # ------------------------------------------------------------------ inference path (NHWC bf16)
    @torch.no_grad()
    def infer(self, x: torch.Tensor, concat_buf: Optional[torch.Tensor] = None, with_figures: bool = False):
        """x: [B,H,W,8] bf16 -> (prob [B,H,W] bf16, mask [B,H,W] u8); with_figures adds the block head's output
        [B,H/4,W/4,8] bf16 (channel 0 = figure logit, channel 1 = formula logit)."""
        B, H, W, _ = x.shape
        assert H % 32 == 0 and W % 32 == 0, "detector input must be a multiple of 32"
        return self.infer_after_stem1(self.stem1.infer(x), concat_buf, with_figures)

    def stem1_packed(self) -> Tuple[torch.Tensor, torch.Tensor, str]:
        """(weight [32,3,3,8] bf16, bias f32, activation) of the first convolution with BN folded - what
        ``ops.resample.resample_stem_items_gpu`` needs to produce ``stem1``'s output straight from the page rasters."""
        if self.stem1._packed is None:
            self.stem1.prepare()
        return self.stem1._packed + (self.stem1.act,)

    @torch.no_grad()
    def infer_after_stem1(self, y1: torch.Tensor, concat_buf: Optional[torch.Tensor] = None, with_figures: bool = False):
        """``infer`` from the first convolution's output [B,H/2,W/2,32] on."""
        B, H, W = y1.shape[0], 2 * y1.shape[1], 2 * y1.shape[2]
        assert H % 32 == 0 and W % 32 == 0, "detector input must be a multiple of 32"
        x = y1
        y = self.stem3.infer(self.stem2.infer(y1))
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                y = blk.infer(y)
            feats.append(y)
        c2, c3, c4, c5 = feats
        i5 = self.in5.infer(c5)
        o4 = self.in4.infer(c4, res=i5, res_mode=C.RES_UP2)
        o3 = self.in3.infer(c3, res=o4, res_mode=C.RES_UP2)
        o2 = self.in2.infer(c2, res=o3, res_mode=C.RES_UP2)
        q = self.q
        if concat_buf is None:
            concat_buf = torch.empty((B, H // 4, W // 4, 4 * q), dtype=torch.bfloat16, device=x.device)
        self.p5.infer(i5, out=concat_buf, out_coff=0, out_mode=C.OUT_REPLICATE, out_f=8)
        self.p4.infer(o4, out=concat_buf, out_coff=q, out_mode=C.OUT_REPLICATE, out_f=4)
        self.p3.infer(o3, out=concat_buf, out_coff=2 * q, out_mode=C.OUT_REPLICATE, out_f=2)
        self.p2.infer(o2, out=concat_buf, out_coff=3 * q)
        h = self.head1.infer(concat_buf)
        fig = self.fig_head.infer(h) if with_figures else None
        if with_figures:
            return self._infer_tail(h) + (fig,)
        return self._infer_tail(h)

    def _infer_tail(self, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        q = self.q
        if self.fuse_head_tail and h.is_cuda and q == 64 and h.shape[2] % 2 == 0:
            # both deconvs + sigmoid + threshold in one kernel: the half-resolution 64-channel map never hits HBM
            if self._head_tail is None:
                if self.head2._packed is None:
                    self.head2.prepare()
                if self.head3._packed is None:
                    self.head3.prepare()
                (w2, b2), (w3, b3) = self.head2._packed, self.head3._packed
                self._head_tail = (w2.reshape(256, 64).contiguous(), b2.float().contiguous(),
                                   C.db_head_a2_fragments(w3.reshape(4, 64)), b3.float().contiguous())
            return C.db_head_fused(h.contiguous(), *self._head_tail, thr=self.bin_thresh)
        h = self.head2.infer(h)
        prob, mask = self.head3.infer(h, prob_thr=self.bin_thresh)
        return prob, mask

    def prepare(self) -> "DBNet":
        self.eval()
        prepare_all(self)
        self._head_tail = None
        return self

    def train(self, mode: bool = True):
        if mode:
            self._head_tail = None
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._head_tail = None
        return super()._apply(fn, *a, **k)

    def flops_per_page(self, H: int, W: int) -> float:
        """Dense conv FLOPs (2*MACs) of one forward at input H x W."""
        total = 0.0

        def conv(cin, cout, k, h, w):
            return 2.0 * cin * cout * k * k * h * w

        s = self.stem1.cout
        w1, w2, w3, w4 = (self.layer1[0].c1.cout, self.layer2[0].c1.cout, self.layer3[0].c1.cout, self.layer4[0].c1.cout)
        total += conv(8, s, 3, H // 2, W // 2) + conv(s, s, 3, H // 2, W // 2) + conv(s, w1, 3, H // 4, W // 4)
        total += 4 * conv(w1, w1, 3, H // 4, W // 4)
        total += conv(w1, w2, 3, H // 8, W // 8) + 3 * conv(w2, w2, 3, H // 8, W // 8) + conv(w1, w2, 1, H // 8, W // 8)
        total += conv(w2, w3, 3, H // 16, W // 16) + 3 * conv(w3, w3, 3, H // 16, W // 16) + conv(w2, w3, 1, H // 16, W // 16)
        total += conv(w3, w4, 3, H // 32, W // 32) + 3 * conv(w4, w4, 3, H // 32, W // 32) + conv(w3, w4, 1, H // 32, W // 32)
        inner, q = self.inner, self.q
        for cin, sc in ((w4, 32), (w3, 16), (w2, 8), (w1, 4)):
            total += conv(cin, inner, 1, H // sc, W // sc) + conv(inner, q, 3, H // sc, W // sc)
        total += conv(inner, q, 3, H // 4, W // 4) + conv(q, 4 * q, 1, H // 4, W // 4) + conv(q, 4, 1, H // 2, W // 2)
        return total
