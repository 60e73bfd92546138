"""CRNN line recognizer: conv stack -> 2 x BiLSTM(256) -> Linear -> CTC.

Input crops are ``[N, 48, W, 8]`` NHWC bf16 (3 real channels), W a bucket width
(multiple of 32).  The conv stack reduces height 48 -> 1 and width by 4, so
T = W / 4 time steps; each sample carries its own valid length
ceil(valid_w / 4) which the persistent LSTM kernel and the CTC decoder honour.

Inference path: 6 MFMA implicit-GEMM convs, per LSTM layer one projection GEMM
+ one persistent recurrent kernel, one FC GEMM, CTC greedy decode - 13 kernel
launches per batch, captured in one hipGraph per (batch, width) bucket by the
runtime.

Training path (``forward``): the same parameters through PyTorch autograd
(nn.LSTM / MIOpen), used by train/ with the HIP CTC loss and fused AdamW.

Reference: the recognizer there is PaddleOCR-torch's rec model fetched by
services/ocr_service/download_models_hf.py:41 and run on CPU inside magic-pdf.
"""
from __future__ import annotations

import string
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from ..ops import ctc as ctc_ops
from ..ops import lstm as lstm_ops
from ..ops.conv import linear_bf16
from ..utils.scripts import CYRILLIC_EXTRA, GREEK_EXTRA, resolve_scripts, to_visual
from .layers import ConvBNAct, prepare_all

# 95 printable ASCII characters; vocabulary index 0 is the CTC blank.
ASCII_CHARSET = "".join(chr(c) for c in range(32, 127))
# Western European letters and the signs that come with them (Latin-1 plus the oe ligatures and the euro sign).  They
# FOLLOW the ASCII classes, so a model grown from an ASCII checkpoint keeps its class indices.
LATIN_EXTRA = ("ÀÁÂÃÄÅÆÇÈÉÊËÌÍÎÏÑÒÓÔÕÖØÙÚÛÜÝ" "àáâãäåæçèéêëìíîïñòóôõöøùúûüýÿß" "Œœ€£§°«»¿¡")
DEFAULT_CHARSET = ASCII_CHARSET + LATIN_EXTRA
# Cyrillic and Greek letters that no Latin letter looks like follow in turn (utils/scripts.py: letters that share a
# shape across the alphabets share the Latin class, and the decoded line is resolved back to code points by word).
MULTI_CHARSET = DEFAULT_CHARSET + CYRILLIC_EXTRA + GREEK_EXTRA


class CRNN(nn.Module):
    def __init__(self, charset: str = DEFAULT_CHARSET, hidden: int = 256, height: int = 48,
                 widths=(64, 128, 256, 256, 512, 512)):
        super().__init__()
        assert height == 48, "conv stack is laid out for 48-pixel-high crops"
        self.charset = charset
        self.num_classes = len(charset) + 1
        self.vocab_padded = (self.num_classes + 3) // 4 * 4
        self.height, self.hidden = height, hidden
        c1, c2, c3, c4, c5, c6 = widths
        self.convs = nn.ModuleList([
            ConvBNAct(3, c1, 3, 2),                      # 24 x W/2
            ConvBNAct(c1, c2, 3, 2),                     # 12 x W/4
            ConvBNAct(c2, c3, 3, 1),                     # 12 x W/4
            ConvBNAct(c3, c4, 3, (2, 1)),                # 6  x W/4
            ConvBNAct(c4, c5, 3, (2, 1)),                # 3  x W/4
            ConvBNAct(c5, c6, (3, 1), 1, padding=(0, 0)),  # 1  x W/4
        ])
        self.rnn1 = nn.LSTM(c6, hidden, batch_first=True, bidirectional=True)
        self.rnn2 = nn.LSTM(2 * hidden, hidden, batch_first=True, bidirectional=True)
        self.fc = nn.Linear(2 * hidden, self.num_classes)
        self._packed = None
        self.fused_train_lstm = hidden in (64, 128, 256)   # device tensors: HIP recurrence kernels in forward()
        # training on the in-repo convolution kernels (ops/conv_train.py: forward, data and weight gradient) instead of
        # the library's; set by the trainers (TrainConfig.own_convs)
        self.own_conv_train = False

    # ------------------------------------------------------------------ character set follows the weights
    def set_charset(self, charset: str) -> None:
        """Resize the classifier to ``charset``; classes shared with the current one (a common prefix) keep their rows,
        new classes start from small random weights and a low bias."""
        old_w, old_b = self.fc.weight.data, self.fc.bias.data
        keep = 1
        while keep - 1 < min(len(charset), len(self.charset)) and charset[keep - 1] == self.charset[keep - 1]:
            keep += 1
        self.charset = charset
        self.num_classes = len(charset) + 1
        self.vocab_padded = (self.num_classes + 3) // 4 * 4
        fc = nn.Linear(old_w.shape[1], self.num_classes).to(device=old_w.device, dtype=old_w.dtype)
        with torch.no_grad():
            fc.weight.mul_(0.1)
            fc.bias.fill_(-2.0)
            k = min(keep, self.num_classes, old_w.shape[0])
            fc.weight[:k] = old_w[:k]
            fc.bias[:k] = old_b[:k]
        self.fc = fc
        self._packed = None

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """A checkpoint with fewer (or more) classes than this model reads the leading part of ``MULTI_CHARSET``:
        the model takes the checkpoint's character set (weights trained on printable ASCII have 96 classifier rows,
        on the Western European set 164, with Cyrillic and Greek 258)."""
        w = state_dict.get("fc.weight")
        if w is not None and w.shape[0] != self.fc.out_features and 1 < w.shape[0] <= len(MULTI_CHARSET) + 1:
            self.set_charset(MULTI_CHARSET[: w.shape[0] - 1])
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    @staticmethod
    def time_steps(width: int) -> int:
        return width // 4

    @staticmethod
    def valid_steps(valid_w: torch.Tensor) -> torch.Tensor:
        return torch.div(valid_w + 3, 4, rounding_mode="floor").clamp(min=1)

    # ------------------------------------------------------------------ training path
    def forward(self, x: torch.Tensor, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: [N,3,48,W] -> logits [N, T, num_classes] (fp32)."""
        if x.is_cuda and self.own_conv_train and self.training:
            y = self._conv_stack_own_kernels(x)            # [N, T, C]
        else:
            y = x
            for c in self.convs:
                y = c(y)
            y = y.squeeze(2).transpose(1, 2)  # [N, T, C]
        if y.is_cuda and self.fused_train_lstm:
            # persistent HIP recurrences (ops/lstm_train.py): 2 launches per layer and direction pair instead of
            # one GEMM + cell kernel per time step
            from ..ops.lstm_train import bilstm_train

            ld = None if lens is None else lens.to(y.device, torch.int32)
            y = bilstm_train(y, self.rnn1, ld)
            y = bilstm_train(y, self.rnn2, ld)
            if self.own_conv_train and self.training:
                # the classifier and its gradients on the in-repo kernels as well: with the convolutions, batch norm,
                # recurrences, projections, CTC loss and AdamW already there, no library GEMM is left in the step
                from ..ops.linear_train import linear_train

                return linear_train(y, self.fc.weight, self.fc.bias).float()
            if not torch.is_autocast_enabled():
                y = y.to(self.fc.weight.dtype)
        elif lens is not None:
            packed = nn.utils.rnn.pack_padded_sequence(y, lens.cpu().long(), batch_first=True, enforce_sorted=False)
            o, _ = self.rnn1(packed)
            o, _ = self.rnn2(o)
            y, _ = nn.utils.rnn.pad_packed_sequence(o, batch_first=True, total_length=y.shape[1])
        else:
            y, _ = self.rnn1(y)
            y, _ = self.rnn2(y)
        return self.fc(y)

    def _conv_stack_own_kernels(self, x: torch.Tensor) -> torch.Tensor:
        """The convolution stack in NHWC bf16 on the in-repo kernels (``ops.conv_train``): the 3-channel first layer
        runs forward on the stem kernel and takes its weight gradient from an im2col of the image times the output
        gradient (``ops.conv_train.conv2d_first_layer_train``), batch norm + ReLU are one in-repo Function on the NHWC
        result (``ops.bn_train``), so no layout copies sit between the layers."""
        from ..ops.bn_train import bn_act_train
        from ..ops.conv_train import conv2d_first_layer_train, conv2d_nhwc_train

        c0 = self.convs[0]
        assert c0.conv.bias is None and c0.bn is not None
        y = bn_act_train(conv2d_first_layer_train(x, c0.conv.weight, c0.stride, c0.padding), c0.bn, c0.act)
        for c in self.convs[1:]:
            assert c.conv.bias is None and c.bn is not None
            z = conv2d_nhwc_train(y, c.conv.weight, c.stride, c.padding)
            y = bn_act_train(z, c.bn, c.act)                                  # batch statistics + ReLU, one Function
        return y.squeeze(1)                                                  # [N, 1, T, C] -> [N, T, C]

    # ------------------------------------------------------------------ inference path
    def prepare(self) -> "CRNN":
        self.eval()
        prepare_all(self)
        dev = self.fc.weight.device

        def pack(m: nn.LSTM) -> lstm_ops.PackedLSTM:
            return lstm_ops.pack_lstm_weights(
                [m.weight_ih_l0.detach(), m.weight_ih_l0_reverse.detach()],
                [m.weight_hh_l0.detach(), m.weight_hh_l0_reverse.detach()],
                [m.bias_ih_l0.detach(), m.bias_ih_l0_reverse.detach()],
                [m.bias_hh_l0.detach(), m.bias_hh_l0_reverse.detach()],
            ).to(dev)

        fw = torch.zeros(self.vocab_padded, self.fc.in_features, device=dev)
        fw[: self.num_classes] = self.fc.weight.detach().float()
        fb = torch.full((self.vocab_padded,), -1e4, device=dev)  # padded classes can never win the argmax
        fb[: self.num_classes] = self.fc.bias.detach().float()
        self._packed = (pack(self.rnn1), pack(self.rnn2), fw.to(torch.bfloat16).contiguous(), fb.contiguous())
        return self

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def infer_logits(self, x: torch.Tensor, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: [N,48,W,8] bf16 -> logits [N, T, vocab_padded] bf16."""
        if self._packed is None:
            self.prepare()
        r1, r2, fw, fb = self._packed
        y = x
        for c in self.convs:
            y = c.infer(y)
        N, _, T, Cc = y.shape
        y = y.reshape(N, T, Cc)
        y = lstm_ops.lstm_layer(y, r1, lens)
        y = lstm_ops.lstm_layer(y, r2, lens)
        return linear_bf16(y, fw, fb)

    @torch.no_grad()
    def infer(self, x: torch.Tensor, lens: Optional[torch.Tensor] = None
              ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """-> (ids [N,T] int32 padded -1, lengths [N] int32, confidence [N] f32)."""
        logits = self.infer_logits(x, lens)
        return ctc_ops.ctc_greedy_decode(logits, lens, blank=0)

    @torch.no_grad()
    def first_conv_packed(self):
        """(weight [C1,3,3,8] bf16, bias f32, activation, stride, padding) of the first convolution with BN folded:
        what ``ops.resample.resample_stem_items_gpu`` needs to produce its output straight from the page rasters."""
        c = self.convs[0]
        if c._packed is None:
            c.prepare()
        return c._packed + (c.act, c.stride, c.padding)

    @torch.no_grad()
    def infer_ragged(self, xs: List[torch.Tensor], lay: "lstm_ops.RaggedLayout", first_conv_done: bool = False
                     ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Recognise crops of several width buckets in one pass.

        xs[b]: [N_b, 48, W_b, 8] bf16 (or, with ``first_conv_done``, the first convolution's output
        [N_b, 24, W_b / 2, C1] from the fused resample kernel).  The conv stack runs per bucket (it needs a uniform width) and
        writes its [N_b, T_b, C] features straight into one packed [R, C] row matrix; everything after
        it - both LSTM layers (input GEMM + ONE persistent recurrent launch each), the classifier GEMM
        and the CTC decode - runs once over all buckets, so the sequential recurrence sees every crop
        of the batch at the same time instead of one under-filled launch per bucket.
        ``lay`` describes the packing (sequence order = concatenation order of ``xs``)."""
        if self._packed is None:
            self.prepare()
        r1, r2, fw, fb = self._packed
        C6 = self.convs[-1].cout
        feat = torch.empty((lay.rows, C6), dtype=torch.bfloat16, device=xs[0].device)
        r0 = 0
        for x in xs:
            y = x
            for c in self.convs[1:-1] if first_conv_done else self.convs[:-1]:
                y = c.infer(y)
            N, T = x.shape[0], x.shape[2] // (2 if first_conv_done else 4)
            self.convs[-1].infer(y, out=feat[r0 : r0 + N * T].view(N, 1, T, C6))
            r0 += N * T
        assert r0 == lay.rows
        y = lstm_ops.lstm_layer_ragged(feat, r1, lay)
        y = lstm_ops.lstm_layer_ragged(y, r2, lay)
        logits = linear_bf16(y, fw, fb)
        return ctc_ops.ctc_greedy_decode_ragged(logits, lay.lens, lay.row_off, lay.tmax, blank=0)

    @property
    def multi_script(self) -> bool:
        return len(self.charset) > len(DEFAULT_CHARSET)

    def decode_strings(self, ids: torch.Tensor, lens: torch.Tensor) -> List[str]:
        out = ctc_ops.ids_to_strings(ids, lens, self.charset)
        if self.multi_script:
            # shape classes -> code points, word by word; ASCII lines return at once
            out = [s if s.isascii() else resolve_scripts(s) for s in out]
        return out

    def encode_text(self, text: str) -> List[int]:
        idx = self.__dict__.get("_class_of")
        if idx is None or idx[0] is not self.charset:
            idx = (self.charset, {c: i + 1 for i, c in enumerate(self.charset)})
            self.__dict__["_class_of"] = idx
        if self.multi_script:
            text = to_visual(text)
        return [idx[1][c] for c in text if c in idx[1]]

    def flops_per_crop(self, W: int) -> float:
        def conv(cin, cout, kh, kw, h, w):
            return 2.0 * cin * cout * kh * kw * h * w

        ch = [c.cout for c in self.convs]
        T = W // 4
        f = conv(8, ch[0], 3, 3, 24, W // 2) + conv(ch[0], ch[1], 3, 3, 12, T) + conv(ch[1], ch[2], 3, 3, 12, T)
        f += conv(ch[2], ch[3], 3, 3, 6, T) + conv(ch[3], ch[4], 3, 3, 3, T) + conv(ch[4], ch[5], 3, 1, 1, T)
        H = self.hidden
        f += 2.0 * T * (ch[5] * 8 * H + 2 * H * 8 * H) + 2.0 * T * 2 * (2 * H * 4 * H)  # projections + recurrences
        f += 2.0 * T * 2 * H * self.vocab_padded
        return f
