"""Train the formula recogniser (image of a display formula -> LaTeX) on synthetic formulas.

The reference ships formula recognition as a downloaded checkpoint (``MFR/unimernet_hf_small_2503``,
services/ocr_service/download_models_hf.py:40) run by magic-pdf on CPU.  No checkpoint is reachable offline, so the
model here - the encoder-decoder of models/trocr.py at ``formula_config()`` size - is trained on formulas drawn from
``pdf.synth.random_formula`` and rendered by matplotlib's mathtext in five math font sets, the way they appear in a
scanned page: ink on tinted paper, sensor noise, blur, contrast and JPEG jitter, a few pixels of context around the
box.  Crops go through the *serving* resample kernel into the 96 x 384 letter-box, targets are byte-level tokens of
the LaTeX source.  Rendering runs in worker processes (mathtext is single-threaded).

    python -m distributed_llms_amd.train.formula_trainer --steps 6000 --out weights

writes ``formula.safetensors`` (bf16) and ``formula_meta.json``.  What it can read is what the grammar can write:
variables, Greek letters, numbers, sub / superscripts, fractions, roots, big operators, a handful of functions and
relations - not arbitrary LaTeX.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import time
from typing import List, Tuple

import numpy as np
import torch

from ..models.trocr import FORMULA_HW, TrOCR, formula_config
from ..ops import resample as rs
from ..pdf import synth
from .trainer import _edit_distance
from .trocr_trainer import Seq2SeqConfig, Seq2SeqTrainer


def formula_sample(seed: int, inline_share: float = 0.0) -> Tuple[np.ndarray, str]:
    """One training crop (uint8 gray, variable size) and its LaTeX.  ``inline_share`` of the crops are short formulas as
    they are cut out of a text line: the line's full height around them, the cut somewhere in the word spaces."""
    rng = np.random.default_rng(seed)
    inline = inline_share > 0 and rng.random() < inline_share
    for _ in range(8):
        latex = synth.random_inline_formula(rng) if inline else synth.random_formula(rng)
        px = int(rng.integers(28, 52)) if inline else int(rng.integers(30, 64))
        r = synth.render_formula_baseline(latex, px, synth.FORMULA_FONTSETS[int(rng.integers(5))])
        if r is not None and r[0].shape[1] >= 12:
            cov, base = r
            break
    else:
        latex, (cov, base) = "x", synth.render_formula_baseline("x", 40)
        px, inline = 40, False
    pad = [int(rng.integers(2, 14)) for _ in range(4)]            # detector boxes are loose by a few pixels
    if inline:
        # the crop is as tall as the line's box: ascender height above the baseline, descender depth below, plus the
        # detector's slack; left and right it ends inside a word space
        top = int(px * rng.uniform(0.85, 1.15)) + int(rng.integers(0, 14))
        bot = int(px * rng.uniform(0.25, 0.45)) + int(rng.integers(0, 14))
        pad = [max(2, top - base), max(2, bot - (cov.shape[0] - base)), int(rng.integers(1, 12)), int(rng.integers(1, 12))]
    h, w = cov.shape
    paper, ink = float(rng.integers(215, 253)), float(rng.integers(5, 90))
    img = np.full((h + pad[0] + pad[1], w + pad[2] + pad[3]), paper, np.float32)
    c = cov.astype(np.float32) / 255.0
    img[pad[0] : pad[0] + h, pad[2] : pad[2] + w] = paper * (1 - c) + ink * c
    if rng.random() < 0.4:
        from PIL import Image, ImageFilter

        img = np.asarray(Image.fromarray(img.astype(np.uint8)).filter(ImageFilter.GaussianBlur(float(rng.uniform(0.3, 1.1)))),
                         dtype=np.float32)
    img += rng.normal(0.0, float(rng.uniform(0.5, 5.0)), size=img.shape)
    cgain, cbias = float(rng.uniform(0.7, 1.15)), float(rng.uniform(-18, 10))
    img = np.clip((img - 128.0) * cgain + 128.0 + cbias, 0, 255).astype(np.uint8)
    if rng.random() < 0.3:
        import io

        from PIL import Image

        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format="JPEG", quality=int(rng.integers(35, 95)))
        img = np.asarray(Image.open(io.BytesIO(buf.getvalue())))
    return np.ascontiguousarray(img), latex


def _render_chunk(args) -> List[Tuple[np.ndarray, str]]:
    seed0, n = args[:2]
    share = args[2] if len(args) > 2 else 0.0
    return [formula_sample(seed0 + k, share) for k in range(n)]


def letterbox_regions(imgs: List[torch.Tensor]) -> List[rs.Region]:
    """Aspect-preserving fit of every crop into the top-left of the 96 x 384 canvas (the rest stays zero)."""
    H, W = FORMULA_HW
    regs = []
    for im in imgs:
        h, w = int(im.shape[0]), int(im.shape[1])
        sc = min(W / w, H / h)
        regs.append(rs.Region(im, (0.0, 0.0, float(w), float(h)), max(1, min(H, int(round(h * sc)))),
                              max(1, min(W, int(round(w * sc))))))
    return regs


class FormulaTrainer(Seq2SeqTrainer):
    def __init__(self, model: TrOCR, cfg: Seq2SeqConfig, device, workers: int = 12, seed0: int = 1_000_000,
                 inline_share: float = 0.0):
        super().__init__(model, cfg, device)
        self.seed0 = seed0
        self.inline_share = inline_share
        self.pool = mp.get_context("spawn").Pool(workers) if workers > 0 else None
        self.pending = []
        self.per_task = 16

    def _submit(self, step: int):
        n = self.cfg.batch_size
        tasks = [(self.seed0 + step * n + k, min(self.per_task, n - k), self.inline_share) for k in range(0, n, self.per_task)]
        if self.pool is None:
            return [_render_chunk(t) for t in tasks]
        return self.pool.map_async(_render_chunk, tasks)

    def batch(self, step: int):
        # keep three batches in flight so the renderers never wait for the GPU
        while len(self.pending) < 3:
            self.pending.append((step + len(self.pending), self._submit(step + len(self.pending))))
        s, fut = self.pending.pop(0)
        if s != step:                      # evaluation asks for other steps: render them directly
            fut = self._submit(step)
        chunks = fut if isinstance(fut, list) else fut.get()
        samples = [x for c in chunks for x in c]
        return self._tensors(samples)

    def _tensors(self, samples):
        H, W = FORMULA_HW
        imgs = [torch.from_numpy(im).to(self.device) for im, _ in samples]
        x = rs.resample_regions(letterbox_regions(imgs), H, W, rs.HALF_MEAN, rs.HALF_STD)
        x = x[..., :3].permute(0, 3, 1, 2).float()
        texts = [t for _, t in samples]
        ids = [self.vocab.encode(t, add_special=True)[: self.model.cfg.max_len] for t in texts]
        L = max(len(i) for i in ids)
        tok = np.full((len(ids), L), self.vocab.pad, dtype=np.int64)
        for k, i in enumerate(ids):
            tok[k, : len(i)] = i
        return x, torch.from_numpy(tok).to(self.device), texts

    @torch.no_grad()
    def evaluate_formulas(self, n: int = 256, seed0: int = 9_000_000, through_kernels: bool = True, inline_share: float = 0.0):
        self.model.eval()
        samples = _render_chunk((seed0, n, inline_share))
        ok = ed = nch = 0
        for k in range(0, n, 128):
            x, _, texts = self._tensors(samples[k : k + 128])
            if through_kernels and x.is_cuda:
                xi = torch.zeros((x.shape[0], x.shape[2], x.shape[3], 8), dtype=torch.bfloat16, device=x.device)
                xi[..., :3] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
                toks, _ = self.model.generate(xi, max_len=self.model.cfg.max_len)
            else:
                toks = self.model.generate_reference(x, 96)
            for p, t in zip(self.model.tokens_to_text(toks, self.vocab), texts):
                ok += int(p == t)
                ed += _edit_distance(p, t)
                nch += len(t)
        self.model.train()
        self.model._pk = None
        return {"exact": ok / n, "cer": ed / max(nch, 1), "n": n}

    def close(self):
        if self.pool is not None:
            self.pool.terminate()


def save_bf16(model, path):
    from safetensors.torch import save_file

    sd = {k: (v.detach().to(torch.bfloat16) if v.is_floating_point() else v.detach()).cpu().contiguous()
          for k, v in model.state_dict().items()}
    tmp = f"{path}.tmp{os.getpid()}"
    save_file(sd, tmp)
    os.replace(tmp, path)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--out", default="weights")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--time-limit", type=float, default=900.0)
    ap.add_argument("--workers", type=int, default=12)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--inline-share", type=float, default=0.0,
                    help="share of the training crops that are short formulas cut out of a text line")
    args = ap.parse_args(argv)
    device = torch.device("cuda:0" if torch.cuda.is_available() and not args.cpu else "cpu")
    torch.manual_seed(0)
    model = TrOCR(formula_config())
    os.makedirs(args.out, exist_ok=True)
    path, meta_path = os.path.join(args.out, "formula.safetensors"), os.path.join(args.out, "formula_meta.json")
    start = 0
    if args.resume and os.path.exists(path):
        from ..utils.weights import load_model_weights

        load_model_weights(model, path)
        if os.path.exists(meta_path):
            start = int(json.load(open(meta_path)).get("steps", 0))
        print(f"resumed {path} at step {start}", flush=True)
    cfg = Seq2SeqConfig(batch_size=args.batch, lr=args.lr, warmup_steps=200 if start == 0 else 20, decay_steps=args.steps,
                        max_grad_norm=1.0, weight_decay=0.01)
    tr = FormulaTrainer(model, cfg, device, workers=args.workers, inline_share=args.inline_share)
    tr.step = start
    t0 = time.perf_counter()
    ev = {}
    try:
        while tr.step < args.steps and time.perf_counter() - t0 < args.time_limit:
            x, tok, _ = tr.batch(tr.step)
            loss = tr.train_step(x, tok)
            if tr.step % 50 == 0:
                print(json.dumps({"step": tr.step, "loss": round(loss, 4), "elapsed_s": round(time.perf_counter() - t0, 1)}), flush=True)
            if tr.step % 1500 == 0:
                ev = tr.evaluate_formulas(128)
                print(json.dumps({"step": tr.step, "eval": ev}), flush=True)
        ev = tr.evaluate_formulas(512)
        if args.inline_share > 0:
            ev["inline"] = tr.evaluate_formulas(512, seed0=9_500_000, inline_share=1.0)
    finally:
        tr.close()
    save_bf16(model, path)
    json.dump({"steps": tr.step, "eval": ev, "config": "models.trocr.formula_config()", "canvas_hw": list(FORMULA_HW),
               "fontsets": list(synth.FORMULA_FONTSETS)}, open(meta_path, "w"), indent=1)
    print(json.dumps({"final_step": tr.step, "eval": ev, "weights": path}), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
