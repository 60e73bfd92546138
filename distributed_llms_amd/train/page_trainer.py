"""Teach the detector and recognizer to read synthetic scanned pages.

No dataset or checkpoint is reachable offline, so the only way to show that upload -> detect -> crop ->
recognise -> decode really produces the text on the page is to train both models on generated pages and evaluate
on pages the training never saw: other seeds, and FONTS that are held out entirely (``synth.HELD_OUT_FONTS``).
One script, one GPU:

* a background producer renders fresh pages on host threads (PIL releases the GIL) and uploads them: mostly
  ``synth.render_layout_page`` (one or two columns, titles, running header / footer, figures; one of
  ``synth.TRAIN_FONTS`` per page; skew up to +-5 degrees; blur, sensor noise, contrast and JPEG-quality jitter),
  the rest the plain single-column ``render_page`` family the benchmark uses;
* DBNet step: the page is shrunk to detector scale with the *inference* resample kernel, random crops,
  text target = the (slanted) line rectangles shrunk by the exact inverse of the serving unclip, figure target =
  the figure rectangles at 1/4 scale for the block head, loss = balanced BCE + dice on both;
* CRNN step: line crops are cut from the full-resolution page with the *inference* affine crop kernel (so
  training sees exactly the serving distribution, slant and all), sorted into width buckets, CTC loss from the
  fused HIP kernel;
* both use the flat-bucket fused AdamW; weights are saved as bf16 safetensors for ``OCR_WEIGHTS_DIR``;
* evaluation runs held-out pages through the real ``OcrEngine`` (detector boxes, not ground truth) and
  reports detection recall and character error rate.
"""
from __future__ import annotations

import argparse
import json
import os
import queue
import threading
import time
from typing import List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from ..models.crnn import CRNN, DEFAULT_CHARSET, MULTI_CHARSET
from ..models.dbnet import DBNet
from ..ops import ctc_loss as ctc_ops
from ..ops import resample as rs
from ..ops.optim import FlatAdamW
from ..parallel.ddp import FlatBuckets
from ..pdf import synth
from .trainer import _edit_distance


def augment_scan(pg: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """Contrast / brightness jitter and, half of the time, a JPEG round trip at a random quality."""
    import io

    from PIL import Image

    c, b = float(rng.uniform(0.7, 1.15)), float(rng.uniform(-18, 10))
    out = np.clip((pg.astype(np.float32) - 128.0) * c + 128.0 + b, 0, 255).astype(np.uint8)
    if rng.random() < 0.5:
        buf = io.BytesIO()
        Image.fromarray(out).save(buf, format="JPEG", quality=int(rng.integers(35, 95)))
        out = np.asarray(Image.open(io.BytesIO(buf.getvalue())))
    return np.ascontiguousarray(out)


def render_training_page(seed: int, fonts=None, plain_share: float = 0.2, max_skew: float = 5.0, formula_share: float = 0.4,
                         i18n_share: float = 0.35, script_share: float = 0.0, table_share: float = 0.0,
                         inline_share: float = 0.0):
    """One page of the training distribution (see the module docstring) and its truth.  ``i18n_share`` of the layout
    pages are written in the Western European vocabulary (accented letters, sharp s, ligatures, currency signs);
    ``script_share`` of them in Russian or Greek (half each, ``utils.scripts``)."""
    rng = np.random.default_rng(seed)
    if rng.random() < plain_share:
        pg, truth = synth.render_page(seed, font_px=int(rng.integers(34, 50)), noise=float(rng.uniform(0.5, 4.0)))
        return pg, truth
    if table_share > 0 and np.random.default_rng(seed ^ 0x7AB1).random() < table_share:
        # ruled and unruled tables: text next to ruling lines, short cells, numbers
        r2 = np.random.default_rng(seed ^ 0x7AB2)
        names = [f for f in synth.available_fonts(fonts or synth.TRAIN_FONTS) if not f.startswith("cm")]
        pg, truth, _ = synth.render_table_page(seed, rows=int(r2.integers(3, 14)), cols=int(r2.integers(2, 6)),
                                               ruled=bool(r2.random() < 0.65), row_pitch=float(r2.uniform(1.3, 2.4)),
                                               column=bool(r2.random() < 0.3), font_name=names[int(r2.integers(len(names)))],
                                               font_px=int(r2.integers(28, 48)))
        return augment_scan(pg, rng), truth
    skew = float(rng.uniform(-max_skew, max_skew)) if rng.random() < 0.7 else 0.0
    blur = float(rng.uniform(0.3, 1.1)) if rng.random() < 0.4 else 0.0
    if script_share > 0 and np.random.default_rng(seed ^ 0x5C21).random() < script_share:
        pg, truth = synth.render_layout_page(seed, fonts=fonts or synth.SCRIPT_TRAIN_FONTS, skew=skew, blur=blur,
                                             noise=float(rng.uniform(0.5, 5.0)), script="cyrillic" if seed % 2 else "greek")
        return augment_scan(pg, rng), truth
    pg, truth = synth.render_layout_page(seed, fonts=fonts or synth.TRAIN_FONTS, skew=skew, blur=blur,
                                         noise=float(rng.uniform(0.5, 5.0)),
                                         formulas=float(rng.uniform(0.3, 0.9)) if rng.random() < formula_share else 0.0,
                                         i18n=i18n_share,
                                         inline=float(rng.uniform(0.1, 0.5)) if inline_share > 0 and
                                         np.random.default_rng(seed ^ 0x1F0A).random() < inline_share else 0.0)
    return augment_scan(pg, rng), truth


class PageProducer:
    """Renders pages on host threads and keeps a bounded queue of (uint8 page tensor on device, truth)."""

    def __init__(self, device, threads: int = 6, depth: int = 24, seed0: int = 100_000, fonts=None,
                 script_share: float = 0.0, table_share: float = 0.0, inline_share: float = 0.0):
        self.device = device
        self.inline_share = inline_share
        self.fonts = fonts
        self.script_share = script_share
        self.table_share = table_share
        self.q: "queue.Queue" = queue.Queue(maxsize=depth)
        self._seed = seed0
        self._lock = threading.Lock()
        self._stop = threading.Event()
        self.threads = [threading.Thread(target=self._run, daemon=True) for _ in range(threads)]
        for t in self.threads:
            t.start()

    def _run(self):
        while not self._stop.is_set():
            with self._lock:
                self._seed += 1
                seed = self._seed
            pg, truth = render_training_page(seed, self.fonts, script_share=self.script_share, table_share=self.table_share,
                                             inline_share=self.inline_share)
            while not self._stop.is_set():
                try:
                    self.q.put((torch.from_numpy(pg), truth), timeout=1.0)
                    break
                except queue.Full:
                    continue

    def get(self) -> Tuple[torch.Tensor, synth.PageTruth]:
        pg, truth = self.q.get()
        return pg.to(self.device, non_blocking=True), truth

    def stop(self):
        self._stop.set()


def line_rects(truth: synth.PageTruth, hw: Tuple[int, int]) -> np.ndarray:
    """Text lines of a (possibly skewed) page as [n, 5] cx, cy, w, h, angle; angle in radians in the crop kernel's
    sense (the line runs along (cos, sin) in image coordinates, y down) - the opposite sign of the page's
    counter-clockwise skew."""
    r = truth.rotated_rects((hw[1], hw[0])).astype(np.float64)
    r[:, 4] = -np.deg2rad(r[:, 4])
    return r


def rect_corners(r: np.ndarray) -> np.ndarray:
    """[n, 5] cx, cy, w, h, angle -> [n, 4, 2] corner points."""
    c, s_ = np.cos(r[:, 4]), np.sin(r[:, 4])
    ux, uy = 0.5 * r[:, 2] * c, 0.5 * r[:, 2] * s_
    vx, vy = -0.5 * r[:, 3] * s_, 0.5 * r[:, 3] * c
    cx, cy = r[:, 0], r[:, 1]
    return np.stack([np.stack([cx - ux - vx, cy - uy - vy], 1), np.stack([cx + ux - vx, cy + uy - vy], 1),
                     np.stack([cx + ux + vx, cy + uy + vy], 1), np.stack([cx - ux + vx, cy - uy + vy], 1)], 1)


def shrink_boxes(boxes: np.ndarray, unclip_ratio: float = 1.5) -> np.ndarray:
    """DB label generation for rectangles, matched to the serving post-process: inset every side by the d for
    which the closed-form unclip (grow by A' u / L' of the shrunk box, ops/dbpost) lands back on the original
    box.  With w' = w - 2d, h' = h - 2d that is (4u + 8) d^2 - (2u + 2)(w + h) d + u w h = 0.  (The textbook
    d = A (1 - r^2) / L with r = 0.4 leaves a long line at 0.16 h, which unclip 1.5 only grows back to 0.4 h:
    ascenders and descenders were cut off and the recognizer read garbage - measured, see docs/TRAINING.md.)"""
    w = boxes[:, 2] - boxes[:, 0]
    h = boxes[:, 3] - boxes[:, 1]
    u = unclip_ratio
    a, b, c = 4 * u + 8, (2 * u + 2) * (w + h), u * w * h
    d = (b - np.sqrt(np.maximum(b * b - 4 * a * c, 0.0))) / (2 * a)
    d = np.minimum(d, 0.45 * np.minimum(w, h))
    out = boxes.copy()
    out[:, 0] += d; out[:, 1] += d; out[:, 2] -= d; out[:, 3] -= d
    return out


def _raster_polys(polys: np.ndarray, W: int, H: int) -> np.ndarray:
    from PIL import Image, ImageDraw

    im = Image.new("L", (W, H), 0)
    d = ImageDraw.Draw(im)
    for q in polys:
        d.polygon([(float(x), float(y)) for x, y in q], fill=1)
    return np.asarray(im, dtype=np.uint8)


def det_targets(truth: synth.PageTruth, hw: Tuple[int, int], scale: float, Hd: int, Wd: int,
                unclip_ratio: float = 1.5) -> Tuple[np.ndarray, np.ndarray]:
    """-> (text kernel mask [Hd, Wd], block masks [2, Hd/4, Wd/4]: figures, display formulas) at detector scale."""
    r = line_rects(truth, hw)
    if r.shape[0]:
        up = np.stack([-0.5 * r[:, 2], -0.5 * r[:, 3], 0.5 * r[:, 2], 0.5 * r[:, 3]], 1)
        sh = shrink_boxes(up, unclip_ratio)
        r = r.copy()
        r[:, 2], r[:, 3] = sh[:, 2] - sh[:, 0], sh[:, 3] - sh[:, 1]
        r[:, :4] *= scale
    text = _raster_polys(rect_corners(r), Wd, Hd) if r.shape[0] else np.zeros((Hd, Wd), np.uint8)
    fig = np.zeros((2, Hd // 4, Wd // 4), np.uint8)
    frm = truth.formulas
    if getattr(truth, "inline", np.zeros((0, 4))).shape[0]:
        # a formula inside a line is marked over its width and the middle 70 % of the line's height: bands of
        # neighbouring lines stay apart on the quarter-scale map, the engine takes the height from the text line
        q = truth.inline.astype(np.float32).copy()
        hq = q[:, 3] - q[:, 1]
        q[:, 1] += 0.15 * hq
        q[:, 3] -= 0.15 * hq
        # the polygon fill marks every map pixel the outline touches: half a map pixel off each end keeps the marked
        # pixels to those whose centre lies in the formula (the ends matter here: they are where the line is cut)
        half = 2.0 / scale
        wide = (q[:, 2] - q[:, 0]) > 4 * half
        q[wide, 0] += half
        q[wide, 2] -= half
        frm = np.concatenate([frm.reshape(-1, 4), q], 0)
    for k, bx in enumerate((truth.figures, frm)):
        if bx.shape[0]:
            f = synth.rotate_boxes(bx, truth.angle, (hw[1], hw[0])).astype(np.float64)
            f[:, 4] = -np.deg2rad(f[:, 4])
            f[:, :4] *= scale / 4.0
            fig[k] = _raster_polys(rect_corners(f), Wd // 4, Hd // 4)
    return text, fig


def det_batch(pages, device, Hd=1344, Wd=960, crop=(640, 640), per_page=2):
    """-> x [N,3,h,w] f32 normalised (detector scale), text target [N,h,w], block targets [N,2,h/4,w/4]."""
    xs, ts, fs = [], [], []
    for img, truth in pages:
        H, W = img.shape
        oh, ow = Hd, int(round(W * Hd / H))
        x = rs.resample_regions([rs.Region(img, (0.0, 0.0, float(W), float(H)), oh, ow)], Hd, Wd)[0]   # [Hd,Wd,8] bf16
        x = x[..., :3].permute(2, 0, 1).float()
        tm, fm = det_targets(truth, (H, W), oh / H, Hd, Wd)
        t = torch.from_numpy(tm).to(device).float()
        f = torch.from_numpy(fm).to(device).float()
        for _ in range(per_page):
            cy = 32 * np.random.randint(0, (Hd - crop[0]) // 32 + 1)
            cx = 32 * np.random.randint(0, (Wd - crop[1]) // 32 + 1)
            xs.append(x[:, cy:cy + crop[0], cx:cx + crop[1]])
            ts.append(t[cy:cy + crop[0], cx:cx + crop[1]])
            fs.append(f[:, cy // 4:(cy + crop[0]) // 4, cx // 4:(cx + crop[1]) // 4])
    return torch.stack(xs), torch.stack(ts), torch.stack(fs)


def det_loss(prob: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    prob = prob.float().clamp(1e-6, 1 - 1e-6)
    pos = target > 0.5
    bce = -(target * prob.log() + (1 - target) * (1 - prob).log())
    npos = pos.sum().clamp(min=1)
    nneg = (~pos).sum().clamp(min=1)
    l_bce = bce[pos].sum() / npos + bce[~pos].sum() / nneg
    inter = (prob * target).sum()
    dice = 1 - 2 * inter / (prob.sum() + target.sum() + 1e-6)
    return l_bce + dice


def fig_loss(logit: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Block head: balanced BCE on logits + dice on the sigmoid, per channel (figures, formulas)."""
    total = logit.new_zeros((), dtype=torch.float32)
    for c in range(logit.shape[1]):
        lg, tg = logit[:, c].float(), target[:, c]
        pos = tg > 0.5
        bce = F.binary_cross_entropy_with_logits(lg, tg, reduction="none")
        l_bce = bce[pos].sum() / pos.sum().clamp(min=1) + bce[~pos].sum() / (~pos).sum().clamp(min=1)
        p = torch.sigmoid(lg)
        total = total + l_bce + 1 - (2 * (p * tg).sum() + 1.0) / (p.sum() + tg.sum() + 1.0)
    return total


REC_BUCKETS = (1280, 2048)      # static training shapes: every new conv shape costs MIOpen a kernel build


def rec_batch(pages, model: CRNN, device, max_lines=64, jitter=0.12, angle_jitter=0.007):
    """Crop lines with the inference kernel. -> x [N,3,48,Wb] f32, in_lens, targets, tgt_lens, texts.
    N and Wb are static (one of REC_BUCKETS, exactly ``max_lines`` rows) so the autograd path sees two shapes."""
    regions, texts = [], []
    for img, truth in pages:
        H, W = img.shape
        rects = line_rects(truth, (H, W))
        items = []
        parts_of = getattr(truth, "line_parts", None) or {}
        for li, (r, t) in enumerate(zip(rects, truth.texts)):
            if li not in parts_of:
                items.append((r, t))
                continue
            # a line with formulas inside: the recogniser reads the stretches of text between them, cut somewhere in the
            # word space next to the formula (the engine cuts there too)
            bx = truth.boxes[li]
            sp = 0.12 * r[3]
            for kind, xa, xb, txt in parts_of[li]:
                if kind != "t":
                    continue
                xa = max(float(bx[0]), xa - np.random.uniform(0.0, 2.0) * sp)
                xb = min(float(bx[2]), xb + np.random.uniform(0.0, 2.0) * sp)
                m = 0.5 * (xa + xb) - 0.5 * float(bx[0] + bx[2])
                items.append((np.array([r[0] + m * np.cos(r[4]), r[1] + m * np.sin(r[4]), xb - xa, r[3], r[4]]), txt))
        for r, t in items:
            # detector boxes are not pixel-exact: jitter every side by a fraction of the line height, and the angle
            # by what the second-moment estimate is off on a short line
            j = np.random.uniform(-jitter, jitter, size=4) * r[3]
            if truth.layout == "table" and np.random.rand() < 0.6:
                # the detector's box of a cell often reaches over the ruling line next to it: crops that start or end
                # with a piece of a rule must read as the cell's text alone
                j[0] -= np.random.uniform(0.0, 0.7) * r[3]
                j[2] += np.random.uniform(0.0, 0.7) * r[3]
            cx, cy = r[0] + 0.5 * (j[0] + j[2]), r[1] + 0.5 * (j[1] + j[3])
            w, h = r[2] + (j[2] - j[0]), r[3] + (j[3] - j[1])
            if np.random.rand() < 0.6:
                # The detector's boxes are taller than the ink: the binarised kernel is ~2 detector pixels thicker
                # than its target and the unclip multiplies that by 2.5, i.e. +10..16 page pixels whatever the font
                # size - half the line height again for a 30-pixel font.  Train on that, not only on tight boxes.
                grow = np.random.uniform(6.0, 22.0)
                h += grow
                w += np.random.uniform(0.3, 1.0) * grow
            if w < 8 or h < 8:
                continue
            ang = float(r[4] + np.random.uniform(-angle_jitter, angle_jitter)) if r[4] != 0.0 else 0.0
            ow = int(np.clip(round(w * 48.0 / h), 8, REC_BUCKETS[-1]))
            regions.append((img, (float(cx - 0.5 * w), float(cy - 0.5 * h), float(cx + 0.5 * w), float(cy + 0.5 * h)), ow, ang))
            texts.append(t)
    widths = np.array([r[2] for r in regions])
    Wb = REC_BUCKETS[0] if np.random.rand() < float((widths <= REC_BUCKETS[0]).mean()) else REC_BUCKETS[-1]
    pool = np.nonzero(widths <= Wb)[0] if Wb == REC_BUCKETS[0] else np.nonzero(widths > REC_BUCKETS[0])[0]
    if pool.size == 0:
        pool, Wb = np.arange(len(regions)), REC_BUCKETS[-1]
    sel = np.random.choice(pool, size=max_lines, replace=pool.size < max_lines)
    x = rs.resample_regions([rs.Region(regions[i][0], regions[i][1], 48, regions[i][2], 0, regions[i][3]) for i in sel],
                            48, Wb, rs.HALF_MEAN, rs.HALF_STD)
    x = x[..., :3].permute(0, 3, 1, 2).float()
    in_lens = torch.tensor([max((regions[i][2] + 3) // 4, 1) for i in sel])
    tg, tl, tx = [], [], []
    for i in sel:
        ids = model.encode_text(texts[i])
        tg.extend(ids)
        tl.append(len(ids))
        tx.append(texts[i])
    return x, in_lens, torch.tensor(tg, dtype=torch.long), torch.tensor(tl, dtype=torch.long), tx


class _Opt:
    def __init__(self, model, lr, wd=0.01, clip=5.0):
        self.flat = FlatBuckets(model.parameters(), 256 << 20)
        self.opt = FlatAdamW([(b.flat_param, b.flat_grad) for b in self.flat.buckets], lr=lr, weight_decay=wd,
                             max_grad_norm=clip, shards=[(0, b.used) for b in self.flat.buckets])
        self.base_lr = lr

    def step(self, it, total, warmup=100):
        lr = self.base_lr * min(1.0, (it + 1) / warmup) * (0.5 * (1 + np.cos(np.pi * min(1.0, it / max(total, 1)))) * 0.95 + 0.05)
        self.opt.lr = float(lr)
        self.opt.step()

    def zero(self):
        self.flat.zero_grad()


def save_bf16(model, path):
    from safetensors.torch import save_file

    sd = {k: (v.detach().to(torch.bfloat16) if v.is_floating_point() else v.detach()).cpu().contiguous()
          for k, v in model.state_dict().items()}
    tmp = f"{path}.tmp{os.getpid()}"
    save_file(sd, tmp)
    os.replace(tmp, path)


def evaluate_pages(engine, pages_truth, verbose=False):
    """Pages through the real inference engine (detector boxes, GPU post-process, layout order, affine crops, CRNN,
    CTC).  Two error rates: per line (a truth line is matched to the detected line whose centre is closest, within
    half a line height) and per page (the engine's text in ITS reading order against the truth in reading order, so
    a column read across counts as errors)."""
    from ..runtime.engine import PageImage

    dev = engine.device
    pages = [PageImage(torch.from_numpy(pg).pin_memory() if dev.type == "cuda" else torch.from_numpy(pg), "eval", k)
             for k, (pg, _) in enumerate(pages_truth)]
    res = engine.process_pages(pages)
    n_truth = n_found = ed = nchar = ed_page = 0
    fig_truth = fig_found = 0
    frm_truth = frm_found = frm_ed = frm_ch = frm_false = 0
    inl_truth = inl_found = inl_ed = inl_ch = inl_false = 0
    import re

    spans_of = lambda text: re.findall(r"\$([^$]*)\$", text)      # noqa: E731
    for r, (pg, tr) in zip(res, pages_truth):
        rects = line_rects(tr, pg.shape)
        eqs = [l for l in r.lines if getattr(l, "kind", "text") == "equation"]
        r.lines = [l for l in r.lines if getattr(l, "kind", "text") != "equation"]
        frects = synth.rotate_boxes(tr.formulas, tr.angle, (pg.shape[1], pg.shape[0])) if tr.formulas.shape[0] else np.zeros((0, 5))
        hit = set()
        for fr, latex in zip(frects, tr.formula_texts):
            frm_truth += 1
            frm_ch += len(latex)
            best, bk = 1e9, -1
            for k, l in enumerate(eqs):
                dd = np.hypot(0.5 * (l.box[0] + l.box[2]) - fr[0], 0.5 * (l.box[1] + l.box[3]) - fr[1])
                if dd < best:
                    best, bk = dd, k
            if bk >= 0 and bk not in hit and best <= 0.5 * fr[3] + 0.1 * fr[2]:
                hit.add(bk)
                frm_found += 1
                frm_ed += min(len(latex), _edit_distance(eqs[bk].text, latex))
            else:
                frm_ed += len(latex)
        frm_false += len(eqs) - len(hit)
        lc = np.array([[0.5 * (l.box[0] + l.box[2]), 0.5 * (l.box[1] + l.box[3])] for l in r.lines], np.float64).reshape(-1, 2)
        used = set()
        for rc, t in zip(rects, tr.texts):
            n_truth += 1
            nchar += len(t)
            if lc.shape[0] == 0:
                ed += len(t)
                continue
            dist = np.hypot(lc[:, 0] - rc[0], lc[:, 1] - rc[1])
            k = int(np.argmin(dist))
            lw = r.lines[k].box[2] - r.lines[k].box[0]
            if dist[k] <= 0.5 * rc[3] + 0.05 * rc[2] and k not in used and 0.8 * rc[2] <= lw <= 1.25 * rc[2] + 2 * rc[3]:
                used.add(k)
                n_found += 1
                ed += _edit_distance(r.lines[k].text, t)
                want, got = spans_of(t), spans_of(r.lines[k].text)
                inl_truth += len(want)
                inl_found += min(len(want), len(got))
                inl_false += max(0, len(got) - len(want))
                if len(want) == len(got):
                    for a_, b_ in zip(want, got):
                        inl_ed += min(len(a_), _edit_distance(b_.strip(), a_))
                        inl_ch += len(a_)
            else:
                ed += len(t)
                inl_truth += len(spans_of(t))
        ed_page += _edit_distance("\n".join(l.text for l in r.lines), "\n".join(tr.texts))
        fig_truth += tr.figures.shape[0]
        fig_found += min(len(r.figures), tr.figures.shape[0])
        if verbose and r.lines:
            print("   sample:", repr(r.lines[0].text), "| truth:", repr(tr.texts[0]), flush=True)
    return {"det_recall": n_found / max(n_truth, 1), "cer": ed / max(nchar, 1), "page_cer": ed_page / max(nchar, 1),
            "boxes_per_page": float(np.mean([len(r.lines) for r in res])),
            "truth_per_page": float(np.mean([len(t.texts) for _, t in pages_truth])),
            "figures_found": fig_found, "figures_truth": fig_truth,
            "formulas_found": frm_found, "formulas_truth": frm_truth, "formulas_false": frm_false,
            "formula_cer": frm_ed / max(frm_ch, 1),
            "inline_found": inl_found, "inline_truth": inl_truth, "inline_false": inl_false,
            "inline_cer": inl_ed / max(inl_ch, 1)}


def held_out_pages(seeds, skew=3.0, fonts=None):
    """Evaluation pages in fonts the training never drew, alternately skewed by +-``skew`` degrees."""
    fonts = fonts or synth.HELD_OUT_FONTS
    out = []
    for k, sd in enumerate(seeds):
        out.append(synth.render_layout_page(sd, fonts=fonts, layout=synth.LAYOUTS[k % len(synth.LAYOUTS)],
                                            skew=skew * (1 if k % 2 == 0 else -1), noise=2.0))
    return out


def evaluate(det, rec, device, seeds, verbose=False, formula_weights=None, inline=False):
    """Three held-out sets: the benchmark's page family (unseen seeds), layout pages in the training fonts, layout
    pages in held-out fonts at +-3 degrees of skew."""
    from ..runtime.engine import EngineConfig, OcrEngine

    formula = None
    if formula_weights and os.path.exists(formula_weights):
        from ..models.trocr import TrOCR, formula_config
        from ..utils.weights import load_model_weights

        formula = TrOCR(formula_config())
        load_model_weights(formula, formula_weights)
    eng = OcrEngine(device, det, rec, EngineConfig(max_batch_pages=4, inline_formulas=inline), formula_rec=formula)
    out = {}
    out["plain"] = evaluate_pages(eng, [synth.render_page(sd) for sd in seeds[:4]], verbose)
    out["layout_train_fonts"] = evaluate_pages(eng, [render_training_page(sd, plain_share=0.0) for sd in seeds], verbose)
    out["held_out_fonts_skew3"] = evaluate_pages(eng, held_out_pages(seeds), verbose)
    # Western European text in fonts never drawn in training (the Lato files cover the whole character set)
    lato = [f for f in synth.HELD_OUT_FONTS if f.startswith("Lato")]
    out["western_european_held_out_fonts"] = evaluate_pages(
        eng, [synth.render_layout_page(sd, fonts=lato, i18n=1.0, layout=synth.LAYOUTS[k % len(synth.LAYOUTS)],
                                       skew=2.0 * (1 if k % 2 == 0 else -1)) for k, sd in enumerate(seeds[:6])], verbose)
    if rec.multi_script:
        for name, sc in (("russian_held_out_fonts", "cyrillic"), ("greek_held_out_fonts", "greek")):
            out[name] = evaluate_pages(
                eng, [synth.render_layout_page(sd, fonts=synth.SCRIPT_HELD_OUT_FONTS, script=sc,
                                               layout=synth.LAYOUTS[k % len(synth.LAYOUTS)],
                                               skew=2.0 * (1 if k % 2 == 0 else -1)) for k, sd in enumerate(seeds)], verbose)
    out["formula_pages"] = evaluate_pages(eng, [synth.render_layout_page(sd, fonts=synth.TRAIN_FONTS, formulas=0.8,
                                                                         layout=synth.LAYOUTS[k % 3])
                                                for k, sd in enumerate(seeds[:4])], verbose)
    if inline:
        out["inline_formula_pages"] = evaluate_pages(eng, [synth.render_layout_page(sd, fonts=synth.TRAIN_FONTS, inline=0.35,
                                                                                    layout=synth.LAYOUTS[k % 3],
                                                                                    skew=1.5 * (k % 3 - 1))
                                                           for k, sd in enumerate(seeds[:6])], verbose)
    det.train()
    rec.train()
    return out


def write_meta(out_dir: str, steps: int, ev, log, charset: str = "", history: str = "") -> None:
    """``train_meta.json``: what the weights were trained on (fonts drawn, fonts never drawn) and how they score."""
    json.dump({"steps": steps, "eval": ev, "log": log[-20:], "charset": charset, "history": history,
               "train_fonts": synth.available_fonts(synth.TRAIN_FONTS),
               "held_out_fonts": synth.available_fonts(synth.HELD_OUT_FONTS),
               "augmentation": {"skew_deg": 5.0, "blur_sigma": [0.3, 1.1], "noise_sigma": [0.5, 5.0],
                                "contrast": [0.7, 1.15], "jpeg_quality": [35, 95], "font_px": [30, 50],
                                "layouts": list(synth.LAYOUTS) + ["plain"],
                                "display_formulas": "40 % of the layout pages carry formulas (synth.random_formula, mathtext)",
                                "inline_formulas": "synth.random_inline_formula inside paragraph lines (--inline-share)",
                                "western_european_pages": "35 % of the layout pages (synth._WORDS_I18N)",
                                "script_train_fonts": synth.available_fonts(synth.SCRIPT_TRAIN_FONTS),
                                "script_held_out_fonts": synth.available_fonts(synth.SCRIPT_HELD_OUT_FONTS)}},
              open(os.path.join(out_dir, "train_meta.json"), "w"), indent=1)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--out", default="weights")
    ap.add_argument("--det-lr", type=float, default=1e-3)
    ap.add_argument("--rec-lr", type=float, default=1e-3)
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--eval-every", type=int, default=500)
    ap.add_argument("--time-limit", type=float, default=900.0, help="stop cleanly (and save) after this many seconds")
    ap.add_argument("--threads", type=int, default=12)
    ap.add_argument("--init", default="", help="directory with weights to start from (step count restarts at 0)")
    ap.add_argument("--charset", choices=("latin", "multi"), default="latin",
                    help="multi: Latin + Cyrillic + Greek shape classes (utils/scripts.py)")
    ap.add_argument("--script-share", type=float, default=0.0, help="share of the layout pages written in Russian / Greek")
    ap.add_argument("--table-share", type=float, default=0.0, help="share of the pages that carry a (ruled or unruled) table")
    ap.add_argument("--inline-share", type=float, default=0.0,
                    help="share of the layout pages whose paragraphs carry formulas inside the lines")
    ap.add_argument("--head-only", action="store_true",
                    help="train nothing but the detector's block head (figure / formula channels) on frozen features: the text "
                         "map, the figure features and the recogniser stay exactly as they are")
    ap.add_argument("--rec-only", action="store_true", help="leave the detector as it is: recogniser steps only")
    ap.add_argument("--lines", type=int, default=64, help="line crops per recogniser step")
    ap.add_argument("--history", default="", help="free text recorded in train_meta.json")
    ap.add_argument("--library-convs", action="store_true",
                    help="recogniser convolutions through the library instead of the in-repo training kernels (ops/conv_train.py)")
    args = ap.parse_args(argv)
    device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    torch.manual_seed(0)
    np.random.seed(0)
    det, rec = DBNet().to(device), CRNN().to(device)
    os.makedirs(args.out, exist_ok=True)
    dp, rp = os.path.join(args.out, "dbnet.safetensors"), os.path.join(args.out, "crnn.safetensors")
    fw = os.path.join(args.out, "formula.safetensors")      # used by the evaluation when present
    start = 0
    if args.init and not (args.resume and os.path.exists(dp)):
        from ..utils.weights import load_model_weights

        load_model_weights(det, os.path.join(args.init, "dbnet.safetensors"), strict=False)     # no block head in older files
        load_model_weights(rec, os.path.join(args.init, "crnn.safetensors"))
        print(f"initialised from {args.init}", flush=True)
    if args.resume and os.path.exists(dp) and os.path.exists(rp):
        from ..utils.weights import load_model_weights

        load_model_weights(det, dp, strict=False)
        load_model_weights(rec, rp)
        meta = os.path.join(args.out, "train_meta.json")
        if os.path.exists(meta):
            start = int(json.load(open(meta)).get("steps", 0))
        print(f"resumed weights from {args.out} (step {start})", flush=True)
    want = MULTI_CHARSET if args.charset == "multi" else DEFAULT_CHARSET
    if rec.charset != want and len(rec.charset) < len(want):
        # weights trained on fewer characters: grow the classifier, old classes keep their rows
        n_old = len(rec.charset)
        rec.set_charset(want)
        rec.to(device)
        print(f"character set grown from {n_old} to {len(rec.charset)} characters", flush=True)
    det.train()
    rec.train()
    rec.own_conv_train = device.type == "cuda" and not args.library_convs
    if args.head_only:
        det.eval()                   # batch-norm statistics stay as shipped
        rec.eval()
    dopt, ropt = _Opt(det.fig_head if args.head_only else det, args.det_lr), _Opt(rec, args.rec_lr)
    it = start
    prod = PageProducer(device, threads=args.threads, script_share=args.script_share, table_share=args.table_share,
                        inline_share=args.inline_share)
    t0 = time.perf_counter()
    alive = threading.Event()

    def heartbeat():   # first steps build MIOpen kernels for every new conv shape: say so instead of going silent
        while not alive.wait(30.0):
            print(json.dumps({"heartbeat_s": round(time.perf_counter() - t0, 1), "step": it}), flush=True)

    threading.Thread(target=heartbeat, daemon=True).start()
    amp = device.type == "cuda"
    log = []
    try:
        while it < args.steps and time.perf_counter() - t0 < args.time_limit:
            pages = [prod.get() for _ in range(3)]
            # ---- detector step
            ld = lf = 0.0
            if args.head_only:
                x, _, ftgt = det_batch(pages, device, per_page=4)
                dopt.zero()
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    with torch.no_grad():
                        feat = det.forward_features(x)
                    flogit = det.fig_head(feat)[:, :2]
                lf = fig_loss(flogit, ftgt)
                lf.backward()
                dopt.step(it, args.steps)
                it += 1
                if it % 25 == 0:
                    rec_ = {"step": it, "fig_loss": round(float(lf), 4), "elapsed_s": round(time.perf_counter() - t0, 1)}
                    log.append(rec_)
                    print(json.dumps(rec_), flush=True)
                continue
            if not args.rec_only:
                x, tgt, ftgt = det_batch(pages, device)
                dopt.zero()
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    prob, flogit = det.forward_all(x)
                ld = det_loss(prob, tgt)
                lf = fig_loss(flogit, ftgt)
                (ld + 0.5 * lf).backward()
                dopt.step(it, args.steps)
            # ---- recognizer step
            xr, il, tg, tl, _ = rec_batch(pages, rec, device, max_lines=args.lines)
            ropt.zero()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                logits = rec(xr, il)
            nll = ctc_ops.ctc_loss(logits.float().contiguous(), tg, il, tl)
            lr_ = (nll / tl.to(nll.device).clamp(min=1).float()).mean()
            lr_.backward()
            ropt.step(it, args.steps)
            it += 1
            if it % 25 == 0:
                rec_ = {"step": it, "det_loss": round(float(ld), 4), "fig_loss": round(float(lf), 4),
                        "ctc_per_char": round(float(lr_), 4),
                        "elapsed_s": round(time.perf_counter() - t0, 1)}
                log.append(rec_)
                print(json.dumps(rec_), flush=True)
            if it % args.eval_every == 0 or it == args.steps:
                ev = evaluate(det, rec, device, seeds=[900_001, 900_002, 900_003, 900_004], verbose=True, formula_weights=fw,
                              inline=args.inline_share > 0)
                print(json.dumps({"step": it, "eval": ev}), flush=True)
                save_bf16(det, dp)
                save_bf16(rec, rp)
                write_meta(args.out, it, ev, log, rec.charset, args.history)
    finally:
        prod.stop()
        alive.set()
    ev = evaluate(det, rec, device, seeds=[900_001, 900_002, 900_003, 900_004, 900_005, 900_006, 900_007, 900_008], verbose=True,
                  formula_weights=fw, inline=args.inline_share > 0)
    save_bf16(det, dp)
    save_bf16(rec, rp)
    write_meta(args.out, it, ev, log, rec.charset, args.history)
    print(json.dumps({"final_step": it, "eval": ev, "weights": args.out}), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
