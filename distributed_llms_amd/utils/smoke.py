"""Round-end smoke test: one tiny end-to-end OCR pass + one CTC training step on cuda:0."""
from __future__ import annotations

import numpy as np
import torch


def run_smoke() -> None:
    from ..models.crnn import CRNN
    from ..pdf import synth
    from ..runtime.document import DocumentPipeline
    from ..runtime.engine import EngineConfig, OcrEngine

    dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    if dev.type == "cuda":
        from ..ops import _loader

        _loader.lib()  # fail loudly if the HIP kernel library is missing on a GPU box
    # ---- inference: 2-page scanned PDF (JPEG without restart markers: the self-synchronising GPU decoder) -> text
    #      through the full pipeline
    pages, priors = [], []
    for i in range(2):
        pg, truth = synth.render_page(i, size=(1240, 1754), font_px=28, margin=110)
        pages.append(pg)
        priors.append(truth.boxes)
    pdf = synth.make_scanned_pdf(pages, codec="jpeg", dpi=150, restart_blocks=0)
    eng = OcrEngine(dev, cfg=EngineConfig(max_batch_pages=2, det_limit_side=960))
    pipe = DocumentPipeline(eng, decode_threads=2)
    doc = pipe.process_pdf(pdf, "smoke", layout_priors=priors)
    assert doc.mode == "ocr" and len(doc.pages) == 2
    assert all(len(p.lines) == len(pr) for p, pr in zip(doc.pages, priors)), "every line must be recognised"
    assert isinstance(doc.text, str)
    pipe.close()

    # ---- training: one CTC step of the recognizer (forward + backward + optimizer)
    from ..train.trainer import CTCTrainer, TrainConfig

    tr = CTCTrainer(CRNN(), TrainConfig(batch_size=8, width=256, steps=1, lr=1e-3), device=dev)
    loss = tr.train_step(*tr.synthetic_batch(0))
    assert np.isfinite(loss), f"non-finite CTC loss {loss}"
    if dev.type == "cuda":
        torch.cuda.synchronize()
    print(f"[smoke] ok: {sum(len(p.lines) for p in doc.pages)} lines recognised, ctc loss {loss:.3f} on {dev}")
