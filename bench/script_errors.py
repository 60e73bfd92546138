"""Which characters the recogniser confuses on held-out-font pages of one script: replace / insert / delete counts from
the aligned truth and output lines (``python bench/script_errors.py WEIGHTS_DIR cyrillic|greek|latin``)."""
import collections
import difflib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from distributed_llms_amd.pdf import synth
from distributed_llms_amd.runtime.engine import PageImage
from distributed_llms_amd.runtime.workers import build_engine
from distributed_llms_amd.train.page_trainer import line_rects

wd, script = sys.argv[1], sys.argv[2]
eng = build_engine("cuda:0", {"model": "full", "weights_dir": wd, "cfg": dict(max_batch_pages=4)})
lato = [f for f in synth.HELD_OUT_FONTS if f.startswith("Lato")]
pages = [synth.render_layout_page(910_000 + k, fonts=lato, script=None if script == "latin" else script,
                                  layout=synth.LAYOUTS[k % len(synth.LAYOUTS)], skew=2.0 * (1 if k % 2 == 0 else -1)) for k in range(8)]
res = eng.process_pages([PageImage(torch.from_numpy(pg).pin_memory(), "e", k) for k, (pg, _) in enumerate(pages)])
conf = collections.Counter()
nchar = nerr = 0
by_font = collections.Counter()
for r, (pg, tr) in zip(res, pages):
    rects = line_rects(tr, pg.shape)
    lc = np.array([[0.5 * (l.box[0] + l.box[2]), 0.5 * (l.box[1] + l.box[3])] for l in r.lines]).reshape(-1, 2)
    for rc, t in zip(rects, tr.texts):
        nchar += len(t)
        if not lc.shape[0]:
            continue
        k = int(np.argmin(np.hypot(lc[:, 0] - rc[0], lc[:, 1] - rc[1])))
        got = r.lines[k].text
        for op, a0, a1, b0, b1 in difflib.SequenceMatcher(None, t, got, autojunk=False).get_opcodes():
            if op != "equal":
                conf[(t[a0:a1], got[b0:b1])] += 1
                nerr += max(a1 - a0, b1 - b0)
                by_font[tr.font_names[0]] += max(a1 - a0, b1 - b0)
print(json.dumps({"script": script, "chars": nchar, "errors": nerr, "cer": round(nerr / max(nchar, 1), 5),
                  "by_font": dict(by_font), "top": [[a, b, n] for (a, b), n in conf.most_common(40)]}, ensure_ascii=False))
