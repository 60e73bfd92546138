"""Debug: per-line comparison of engine output vs truth on held-out synthetic pages (trained weights)."""
import sys, json
import numpy as np, torch
sys.path.insert(0, ".")
from distributed_llms_amd.pdf import synth
from distributed_llms_amd.runtime.engine import EngineConfig, OcrEngine, PageImage
from distributed_llms_amd.runtime.workers import build_engine
from distributed_llms_amd.models.dbnet import DBNet
from distributed_llms_amd.models.crnn import CRNN
from distributed_llms_amd.utils.weights import load_model_weights

dev = torch.device("cuda:0")
det, rec = DBNet().to(dev), CRNN().to(dev)
load_model_weights(det, "weights/dbnet.safetensors"); load_model_weights(rec, "weights/crnn.safetensors")
for use_prior in (False, True):
    for graphs in (True, False):
        eng = OcrEngine(dev, det, rec, EngineConfig(max_batch_pages=4, use_hip_graphs=graphs))
        pages, truths = [], []
        for s in (900001, 900002):
            pg, tr = synth.render_page(s)
            pages.append(PageImage(torch.from_numpy(pg).pin_memory(), "e", len(pages), layout_prior=tr.boxes if use_prior else None))
            truths.append(tr)
        res = eng.process_pages(pages)
        print("==== prior", use_prior, "graphs", graphs)
        for r, tr in zip(res, truths):
            print(" page lines", len(r.lines), "truth", len(tr.texts))
            for k, ln in enumerate(r.lines[:60]):
                # nearest truth by y
                j = int(np.argmin([abs(b[1] - ln.box[1]) + abs(b[0] - ln.box[0]) for b in tr.boxes]))
                b = tr.boxes[j]
                ok = ln.text == tr.texts[j]
                print("  %s box=%s truth=%s | %r | %r" % ("OK " if ok else "BAD", [int(v) for v in ln.box], [int(v) for v in b], ln.text, tr.texts[j]))
            break
