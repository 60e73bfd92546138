#!/usr/bin/env python3
"""CPU proxy for the reference's execution model.

The reference runs a third-party CPU OCR stack (magic-pdf) inside ``ProcessPoolExecutor(os.cpu_count())``
(services/ocr_service/ocr_service.py:42, 274-278).  That stack cannot be installed here, so this script runs the
SAME model graphs as the GPU engine (same DBNet + CRNN definitions, same random-init weights, same synthetic
300-DPI A4 JPEG pages, same ground-truth line boxes) through plain PyTorch CPU eager in a process pool - one
document per worker, one torch thread per worker so ``cpu_count`` workers do not oversubscribe the cores.
It is labelled "CPU proxy", never "reference"."""
import argparse
import io
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def process_document(args):
    seed, n_pages, threads = args
    import numpy as np
    import torch
    import torch.nn.functional as F
    from PIL import Image

    from distributed_llms_amd.models.crnn import CRNN
    from distributed_llms_amd.models.dbnet import DBNet
    from distributed_llms_amd.ops.ctc import ctc_greedy_reference
    from distributed_llms_amd.pdf import synth

    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    det, rec = DBNet().eval(), CRNN().eval()
    pages = []
    for i in range(n_pages):
        pg, truth = synth.render_page(seed * 100 + i)
        pages.append((synth.encode_page_jpeg(pg, restart_blocks=0), truth.boxes))
    t0 = time.perf_counter()
    n_lines = 0
    with torch.no_grad():
        for jpg, boxes in pages:
            img = torch.from_numpy(np.asarray(Image.open(io.BytesIO(jpg)))).float()           # decode
            x = F.interpolate(img[None, None], size=(1344, 960), mode="area")                  # page -> detector input
            x = ((x / 255.0).expand(-1, 3, -1, -1) - 0.45) / 0.225
            prob = det(x)                                                                       # DBNet
            (prob > 0.3).sum()                                                                  # (post-process stand-in)
            crops = []
            for b in boxes:                                                                     # line crops at full res
                x0, y0, x1, y1 = [int(v) for v in b]
                c = img[max(y0, 0):y1, max(x0, 0):x1][None, None]
                w = max(8, min(2560, int(round((x1 - x0) * 48.0 / max(y1 - y0, 1)))))
                crops.append((F.interpolate(c, size=(48, w), mode="bilinear", align_corners=False), w))
            # width buckets of 256 like the GPU engine
            buckets = {}
            for c, w in crops:
                buckets.setdefault(-(-w // 256) * 256, []).append((c, w))
            for wb, items in buckets.items():
                xb = torch.ones(len(items), 1, 48, wb) * 255.0
                for k, (c, w) in enumerate(items):
                    xb[k, :, :, :w] = c[0]
                xb = (xb / 127.5 - 1.0).expand(-1, 3, -1, -1)
                lens = torch.tensor([max((w + 3) // 4, 1) for _, w in items])
                logits = rec(xb, lens)                                                          # CRNN
                ctc_greedy_reference(logits, lens.to(torch.int32))
                n_lines += len(items)
    return time.perf_counter() - t0, n_pages, n_lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=os.cpu_count())
    ap.add_argument("--pages-per-doc", type=int, default=1)
    ap.add_argument("--docs", type=int, default=None)
    ap.add_argument("--threads", type=int, default=1)
    args = ap.parse_args()
    docs = args.docs or args.workers
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=args.workers) as ex:
        res = list(ex.map(process_document, [(s, args.pages_per_doc, args.threads) for s in range(docs)]))
    wall = time.perf_counter() - t0
    busy = max(r[0] for r in res)
    pages = sum(r[1] for r in res)
    print(json.dumps({"metric": "CPU proxy pages/sec (same graphs, torch CPU eager, process pool)",
                      "value": round(pages / busy, 4), "unit": "pages/s", "pages": pages, "lines": sum(r[2] for r in res),
                      "workers": args.workers, "threads_per_worker": args.threads, "cpus": os.cpu_count(),
                      "compute_s": round(busy, 2), "wall_s_incl_startup": round(wall, 2),
                      "p50_page_latency_s": round(float(sorted(r[0] / r[1] for r in res)[len(res) // 2]), 2)}), flush=True)


if __name__ == "__main__":
    main()
