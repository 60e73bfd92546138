"""Per-formula report on augmented layout pages: page skew, layout, found or missed, edit distance, both strings
(``python bench/formula_probe.py [WEIGHTS_DIR] [N_PAGES]``)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from distributed_llms_amd.pdf import synth
from distributed_llms_amd.runtime.engine import PageImage
from distributed_llms_amd.runtime.workers import build_engine
from distributed_llms_amd.train.page_trainer import render_training_page
from distributed_llms_amd.train.trainer import _edit_distance

wd = sys.argv[1] if len(sys.argv) > 1 else "weights"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
eng = build_engine("cuda:0", {"model": "full", "weights_dir": wd, "cfg": dict(max_batch_pages=4)})
pages = []
sd = 900_000
while len(pages) < n:
    pg, tr = render_training_page(sd, plain_share=0.0, formula_share=1.0)
    sd += 1
    if tr.formulas.shape[0]:
        pages.append((pg, tr))
res = eng.process_pages([PageImage(torch.from_numpy(pg).pin_memory(), "e", k) for k, (pg, _) in enumerate(pages)])
rows = []
for r, (pg, tr) in zip(res, pages):
    eqs = [l for l in r.lines if l.kind == "equation"]
    fr = synth.rotate_boxes(tr.formulas, tr.angle, (pg.shape[1], pg.shape[0]))
    for f, latex in zip(fr, tr.formula_texts):
        best, bk = 1e9, -1
        for k, l in enumerate(eqs):
            dd = float(np.hypot(0.5 * (l.box[0] + l.box[2]) - f[0], 0.5 * (l.box[1] + l.box[3]) - f[1]))
            if dd < best:
                best, bk = dd, k
        hit = bk >= 0 and best <= 0.5 * f[3] + 0.1 * f[2]
        got = eqs[bk].text if hit else ""
        rows.append({"skew": round(tr.angle, 2), "layout": tr.layout, "w": int(f[2]), "h": int(f[3]), "found": bool(hit),
                     "ed": _edit_distance(got, latex) if hit else len(latex), "len": len(latex), "truth": latex, "got": got,
                     "box": [round(float(v)) for v in eqs[bk].box] if hit else None})
found = sum(r["found"] for r in rows)
print(json.dumps({"formulas": len(rows), "found": found, "cer_found": round(sum(r["ed"] for r in rows if r["found"]) / max(1, sum(r["len"] for r in rows if r["found"])), 4),
                  "false": sum(len([l for l in r.lines if l.kind == "equation"]) for r in res) - found}))
for r in rows:
    if not r["found"] or r["ed"] > 0:
        print(json.dumps(r))
