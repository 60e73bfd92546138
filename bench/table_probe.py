"""Detections on synthetic table pages: figure boxes, line kinds, texts (``python bench/table_probe.py``)."""
import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from distributed_llms_amd.pdf import synth
from distributed_llms_amd.runtime.engine import PageImage
from distributed_llms_amd.runtime.workers import build_engine
eng = build_engine("cuda:0", {"model": "full", "weights_dir": "weights", "cfg": dict(max_batch_pages=4)})
for kw in (dict(rows=5, cols=3, ruled=True), dict(rows=5, cols=3, ruled=False), dict(rows=4, cols=3, ruled=False, column=True)):
    pg, tr, grid = synth.render_table_page(31_337, **kw)
    n0 = tr.texts.index(grid[0][0]); tb = tr.boxes[n0:n0 + sum(len(r) for r in grid)]
    print(kw, "table bbox", [int(tb[:,0].min()), int(tb[:,1].min()), int(tb[:,2].max()), int(tb[:,3].max())])
    r = eng.process_pages([PageImage(torch.from_numpy(pg).pin_memory(), "e", 0)])[0]
    print(" figures:", [[int(v) for v in f.box] for f in r.figures])
    print(" lines:", len(r.lines), "truth", len(tr.texts), "kinds", {k: sum(1 for l in r.lines if l.kind == k) for k in ("text", "cell", "equation")})
    for l in r.lines:
        if l.box[1] > tb[:,1].min() - 80 and l.box[3] < tb[:,3].max() + 80:
            print("   ", l.kind, l.block, [int(v) for v in l.box], repr(l.text))
