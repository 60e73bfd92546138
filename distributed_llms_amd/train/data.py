"""Synthetic line-crop dataset for recognizer training.

Rendered on the host with PIL (``pdf.synth.render_line``), resized to the
recognizer height, normalised to [-1, 1] and padded to a fixed bucket width so
every step has one static shape (good for MIOpen kernel caching and hipGraphs).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from ..pdf import synth


class SyntheticLines:
    def __init__(self, charset: str, height: int = 48, width: int = 512, max_chars: int = 40, seed: int = 0,
                 font_px: int = 34, vocab: int = 0):
        self.charset = charset
        self.idx = {c: i + 1 for i, c in enumerate(charset)}
        self.height, self.width, self.max_chars = height, width, max_chars
        self.seed = seed
        self.font_px = font_px
        self.vocab = vocab  # > 0: draw from a small fixed set of strings (overfit tests)
        self._fixed: List[str] = []
        if vocab > 0:
            rng = np.random.default_rng(seed)
            self._fixed = [synth.random_text(rng, max_chars, min_words=1) for _ in range(vocab)]

    def sample_text(self, rng: np.random.Generator) -> str:
        if self._fixed:
            return self._fixed[int(rng.integers(len(self._fixed)))]
        t = synth.random_text(rng, int(rng.integers(4, self.max_chars + 1)), min_words=1)
        return "".join(c for c in t if c in self.idx) or "a"

    def render(self, text: str, seed: int) -> np.ndarray:
        img = synth.render_line(text, self.height, self.font_px, seed)
        if img.shape[1] > self.width:
            from PIL import Image

            img = np.asarray(Image.fromarray(img).resize((self.width, self.height), Image.BILINEAR))
        return img

    def batch(self, step: int, batch_size: int, rank: int = 0, world: int = 1
              ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, List[str]]:
        """-> (images [N,3,H,W] f32 in [-1,1], input_lens [N], targets [sum L], target_lens [N], texts).
        Deterministic in (seed, step, rank): rank r of a DP=w run draws the r-th slice of the global batch."""
        x = np.full((batch_size, self.height, self.width), 255, dtype=np.uint8)
        in_lens = np.zeros((batch_size,), dtype=np.int64)
        texts, tgt, tgt_lens = [], [], []
        for j in range(batch_size):
            gid = (step * world + rank) * batch_size + j
            rng = np.random.default_rng([self.seed, gid])
            t = self.sample_text(rng)
            img = self.render(t, int(rng.integers(1 << 30)))
            w = img.shape[1]
            x[j, :, :w] = img
            x[j, :, w:] = img[:, -1:].mean()
            ids = [self.idx[c] for c in t if c in self.idx]
            # CTC needs T >= L (+ repeats); T = ceil(w / 4)
            in_lens[j] = max((w + 3) // 4, 1)
            texts.append(t)
            tgt.extend(ids)
            tgt_lens.append(len(ids))
        xf = torch.from_numpy(x).float().div_(127.5).sub_(1.0)
        xf = xf[:, None].expand(-1, 3, -1, -1).contiguous()
        return (xf, torch.from_numpy(in_lens), torch.tensor(tgt, dtype=torch.long),
                torch.tensor(tgt_lens, dtype=torch.long), texts)
