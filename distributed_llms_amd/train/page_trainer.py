"""Teach the random-init detector and recognizer to read the synthetic page family.

No dataset or checkpoint is reachable offline, so the only way to show that upload -> detect -> crop ->
recognise -> decode really produces the text on the page is to train both models on pages from
``pdf.synth.render_page`` and evaluate on unseen seeds.  One script, one GPU, a few minutes:

* a background producer renders fresh pages on host threads (PIL releases the GIL) and uploads them;
* DBNet step: the page is shrunk to detector scale with the *inference* resample kernel, random crops,
  target = line boxes shrunk by the exact inverse of the serving unclip, loss = balanced BCE + dice;
* CRNN step: line crops are cut from the full-resolution page with the *inference* crop kernel (so training
  sees exactly the serving distribution), sorted into width buckets, CTC loss from the fused HIP kernel;
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

from ..models.crnn import CRNN
from ..models.dbnet import DBNet
from ..ops import ctc_loss as ctc_ops
from ..ops import resample as rs
from ..ops.optim import FlatAdamW
from ..parallel.ddp import FlatBuckets
from ..pdf import synth
from .trainer import _edit_distance


class PageProducer:
    """Renders pages on host threads and keeps a bounded queue of (uint8 page tensor on device, truth)."""

    def __init__(self, device, threads: int = 6, depth: int = 24, seed0: int = 100_000):
        self.device = device
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
            rng = np.random.default_rng(seed)
            pg, truth = synth.render_page(seed, font_px=int(rng.integers(34, 50)), noise=float(rng.uniform(0.5, 4.0)))
            try:
                self.q.put((torch.from_numpy(pg), truth), timeout=1.0)
            except queue.Full:
                continue

    def get(self) -> Tuple[torch.Tensor, synth.PageTruth]:
        pg, truth = self.q.get()
        return pg.to(self.device, non_blocking=True), truth

    def stop(self):
        self._stop.set()


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


def det_batch(pages, device, Hd=1344, Wd=960, crop=(640, 640), per_page=2):
    """-> x [N,3,h,w] f32 normalised (detector scale), target mask [N,h,w]."""
    xs, ts = [], []
    for img, truth in pages:
        H, W = img.shape
        oh, ow = Hd, int(round(W * Hd / H))
        x = rs.resample_regions([rs.Region(img, (0.0, 0.0, float(W), float(H)), oh, ow)], Hd, Wd)[0]   # [Hd,Wd,8] bf16
        x = x[..., :3].permute(2, 0, 1).float()
        sb = shrink_boxes(truth.boxes) * (oh / H)
        t = torch.zeros((Hd, Wd), dtype=torch.float32)
        for b in sb:
            x0, y0, x1, y1 = [int(round(v)) for v in b]
            t[max(y0, 0):max(y1, y0 + 1), max(x0, 0):max(x1, x0 + 1)] = 1.0
        t = t.to(device)
        for _ in range(per_page):
            cy = np.random.randint(0, Hd - crop[0] + 1)
            cx = np.random.randint(0, Wd - crop[1] + 1)
            xs.append(x[:, cy:cy + crop[0], cx:cx + crop[1]])
            ts.append(t[cy:cy + crop[0], cx:cx + crop[1]])
    return torch.stack(xs), torch.stack(ts)


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


REC_BUCKETS = (1280, 2048)      # static training shapes: every new conv shape costs MIOpen a kernel build


def rec_batch(pages, model: CRNN, device, max_lines=64, jitter=0.12):
    """Crop lines with the inference kernel. -> x [N,3,48,Wb] f32, in_lens, targets, tgt_lens, texts.
    N and Wb are static (one of REC_BUCKETS, exactly ``max_lines`` rows) so the autograd path sees two shapes."""
    regions, texts = [], []
    for img, truth in pages:
        H, W = img.shape
        for b, t in zip(truth.boxes, truth.texts):
            # detector boxes are not pixel-exact: jitter every side by a fraction of the line height
            j = (np.random.uniform(-jitter, jitter, size=4) * (b[3] - b[1])).astype(np.float32)
            x0, y0, x1, y1 = np.clip(b + j, 0, [W, H, W, H])
            if x1 - x0 < 8 or y1 - y0 < 8:
                continue
            ow = int(np.clip(round((x1 - x0) * 48.0 / (y1 - y0)), 8, REC_BUCKETS[-1]))
            regions.append((img, (float(x0), float(y0), float(x1), float(y1)), ow))
            texts.append(t)
    widths = np.array([r[2] for r in regions])
    Wb = REC_BUCKETS[0] if np.random.rand() < float((widths <= REC_BUCKETS[0]).mean()) else REC_BUCKETS[-1]
    pool = np.nonzero(widths <= Wb)[0] if Wb == REC_BUCKETS[0] else np.nonzero(widths > REC_BUCKETS[0])[0]
    if pool.size == 0:
        pool, Wb = np.arange(len(regions)), REC_BUCKETS[-1]
    sel = np.random.choice(pool, size=max_lines, replace=pool.size < max_lines)
    x = rs.resample_regions([rs.Region(regions[i][0], regions[i][1], 48, regions[i][2]) for i in sel], 48, Wb,
                            rs.HALF_MEAN, rs.HALF_STD)
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


def evaluate(det, rec, device, seeds, verbose=False):
    """Held-out pages through the real inference engine (detector boxes, GPU post-process, crops, CRNN, CTC)."""
    from ..runtime.engine import EngineConfig, OcrEngine, PageImage

    eng = OcrEngine(device, det, rec, EngineConfig(max_batch_pages=4))
    pages, truths = [], []
    for s in seeds:
        pg, tr = synth.render_page(s)
        pages.append(PageImage(torch.from_numpy(pg).pin_memory() if device.type == "cuda" else torch.from_numpy(pg), "eval", len(pages)))
        truths.append(tr)
    res = eng.process_pages(pages)
    n_truth = n_found = 0
    ed = nchar = 0
    for r, tr in zip(res, truths):
        used = set()
        for b, t in zip(tr.boxes, tr.texts):
            n_truth += 1
            best, bi = 0.0, -1
            for k, ln in enumerate(r.lines):
                ix = max(0.0, min(b[2], ln.box[2]) - max(b[0], ln.box[0])) * max(0.0, min(b[3], ln.box[3]) - max(b[1], ln.box[1]))
                un = (b[2] - b[0]) * (b[3] - b[1]) + (ln.box[2] - ln.box[0]) * (ln.box[3] - ln.box[1]) - ix
                iou = ix / max(un, 1e-6)
                if iou > best:
                    best, bi = iou, k
            if best >= 0.5 and bi not in used:
                used.add(bi)
                n_found += 1
                ed += _edit_distance(r.lines[bi].text, t)
            else:
                ed += len(t)
            nchar += len(t)
        if verbose and r.lines:
            print("   sample:", repr(r.lines[0].text), "| truth:", repr(tr.texts[0]))
    det.train()
    rec.train()
    return {"det_recall": n_found / max(n_truth, 1), "cer": ed / max(nchar, 1),
            "boxes_per_page": float(np.mean([len(r.lines) for r in res])),
            "truth_per_page": float(np.mean([len(t.texts) for t in truths]))}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--out", default="weights")
    ap.add_argument("--det-lr", type=float, default=1e-3)
    ap.add_argument("--rec-lr", type=float, default=1e-3)
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--eval-every", type=int, default=500)
    ap.add_argument("--time-limit", type=float, default=900.0, help="stop cleanly (and save) after this many seconds")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args(argv)
    device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    torch.manual_seed(0)
    np.random.seed(0)
    det, rec = DBNet().to(device), CRNN().to(device)
    os.makedirs(args.out, exist_ok=True)
    dp, rp = os.path.join(args.out, "dbnet.safetensors"), os.path.join(args.out, "crnn.safetensors")
    start = 0
    if args.resume and os.path.exists(dp) and os.path.exists(rp):
        from ..utils.weights import load_model_weights

        load_model_weights(det, dp)
        load_model_weights(rec, rp)
        meta = os.path.join(args.out, "train_meta.json")
        if os.path.exists(meta):
            start = int(json.load(open(meta)).get("steps", 0))
        print(f"resumed weights from {args.out} (step {start})", flush=True)
    det.train()
    rec.train()
    dopt, ropt = _Opt(det, args.det_lr), _Opt(rec, args.rec_lr)
    it = start
    prod = PageProducer(device, threads=args.threads)
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
            x, tgt = det_batch(pages, device)
            dopt.zero()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                prob = det(x)
            ld = det_loss(prob, tgt)
            ld.backward()
            dopt.step(it, args.steps)
            # ---- recognizer step
            xr, il, tg, tl, _ = rec_batch(pages, rec, device)
            ropt.zero()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                logits = rec(xr, il)
            nll = ctc_ops.ctc_loss(logits.float().contiguous(), tg, il, tl)
            lr_ = (nll / tl.to(nll.device).clamp(min=1).float()).mean()
            lr_.backward()
            ropt.step(it, args.steps)
            it += 1
            if it % 25 == 0:
                rec_ = {"step": it, "det_loss": round(float(ld), 4), "ctc_per_char": round(float(lr_), 4),
                        "elapsed_s": round(time.perf_counter() - t0, 1)}
                log.append(rec_)
                print(json.dumps(rec_), flush=True)
            if it % args.eval_every == 0 or it == args.steps:
                ev = evaluate(det, rec, device, seeds=[900_001, 900_002, 900_003, 900_004], verbose=True)
                print(json.dumps({"step": it, "eval": ev}), flush=True)
                save_bf16(det, dp)
                save_bf16(rec, rp)
                json.dump({"steps": it, "eval": ev, "log": log[-20:]}, open(os.path.join(args.out, "train_meta.json"), "w"))
    finally:
        prod.stop()
        alive.set()
    ev = evaluate(det, rec, device, seeds=[900_001, 900_002, 900_003, 900_004, 900_005, 900_006], verbose=True)
    save_bf16(det, dp)
    save_bf16(rec, rp)
    json.dump({"steps": it, "eval": ev, "log": log[-20:]}, open(os.path.join(args.out, "train_meta.json"), "w"))
    print(json.dumps({"final_step": it, "eval": ev, "weights": args.out}), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
