"""Tables through the engine: how many synthetic table pages come back with exactly one table of the right shape, the
character error rate of its cells, and how many pages show a figure that is not there
(``python bench/table_eval.py [WEIGHTS_DIR] [N_PAGES] [WRAP]``; WRAP: share of the body rows whose label runs onto a
second line, 0 by default)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from distributed_llms_amd.pdf import synth
from distributed_llms_amd.runtime.document import page_content_items
from distributed_llms_amd.runtime.engine import PageImage
from distributed_llms_amd.runtime.workers import build_engine
from distributed_llms_amd.train.trainer import _edit_distance

wd = sys.argv[1] if len(sys.argv) > 1 else "weights"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
wrap = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
eng = build_engine("cuda:0", {"model": "full", "weights_dir": wd, "cfg": dict(max_batch_pages=4)})
fonts = [f for f in synth.available_fonts(synth.HELD_OUT_FONTS) if f.startswith("Lato")] + ["DejaVuSans.ttf", "DejaVuSerif.ttf"]
rng = np.random.default_rng(7)
cases, pages = [], []
for k in range(n):
    column = bool(k % 3 == 0)
    kw = dict(rows=int(rng.integers(3, 12)), cols=int(rng.integers(2, 4 if column else 6)), ruled=bool(k % 2 == 0),
              row_pitch=float(rng.uniform(1.35, 2.3)), column=column, font_name=fonts[k % len(fonts)],
              font_px=int(rng.integers(30, 46)))
    pg, tr, grid = synth.render_table_page(70_000 + k, wrap=wrap, **kw)
    cases.append((kw, grid))
    pages.append(pg)
res = eng.process_pages([PageImage(torch.from_numpy(pg).pin_memory(), "e", k) for k, pg in enumerate(pages)])
out = {"ruled": dict(pages=0, one_table=0, shape_ok=0, ed=0, ch=0, false_figures=0),
       "unruled": dict(pages=0, one_table=0, shape_ok=0, ed=0, ch=0, false_figures=0)}
bad = []
for (kw, grid), r in zip(cases, res):
    o = out["ruled" if kw["ruled"] else "unruled"]
    o["pages"] += 1
    o["false_figures"] += int(len(r.figures) > 0)
    tabs = [it for it in page_content_items(r) if it["type"] == "table"]
    want = " ".join(c for row in grid for c in row)
    o["ch"] += len(want)
    if len(tabs) != 1:
        o["ed"] += len(want)
        bad.append({**kw, "tables": len(tabs)})
        continue
    o["one_table"] += 1
    cells = tabs[0]["cells"]
    ok = len(cells) == len(grid) and all(len(a) == len(b) for a, b in zip(cells, grid))
    o["shape_ok"] += int(ok)
    o["ed"] += min(len(want), _edit_distance(" ".join(c for row in cells for c in row), want))
    if not ok:
        bad.append({**kw, "got_shape": [len(cells), max(len(a) for a in cells)]})
for o in out.values():
    o["cell_cer"] = round(o.pop("ed") / max(o.pop("ch"), 1), 4)
if wrap > 0:
    out["wrapped_label_share"] = wrap
print(json.dumps(out))
for b in bad[:12]:
    print(json.dumps(b))
