"""Formulas inside text lines, through the engine: how many are found, how well they and the text around them are read
(``python bench/inline_probe.py [WEIGHTS_DIR] [N_PAGES] [train|held_out] [INLINE_MIN_CONF] [INLINE_MIN_SCORE] [INLINE_INSET]``).  First line the summary, then one line per
text line that was not read exactly."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from distributed_llms_amd.pdf import synth
from distributed_llms_amd.runtime.engine import PageImage
from distributed_llms_amd.runtime.workers import build_engine
from distributed_llms_amd.train.page_trainer import augment_scan, line_rects
from distributed_llms_amd.train.trainer import _edit_distance

wd = sys.argv[1] if len(sys.argv) > 1 else "weights"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
fonts = synth.HELD_OUT_FONTS if len(sys.argv) > 3 and sys.argv[3] == "held_out" else synth.TRAIN_FONTS
ecfg = dict(max_batch_pages=4, inline_formulas=True)
if len(sys.argv) > 4:
    ecfg["inline_min_conf"] = float(sys.argv[4])
if len(sys.argv) > 5:
    ecfg["inline_min_score"] = float(sys.argv[5])
if len(sys.argv) > 6:
    ecfg["inline_inset"] = float(sys.argv[6])
eng = build_engine("cuda:0", {"model": "full", "weights_dir": wd, "cfg": ecfg})
pages = []
for k in range(n):
    rng = np.random.default_rng(77_000 + k)
    pg, tr = synth.render_layout_page(910_000 + k, fonts=fonts, layout=synth.LAYOUTS[k % 3], inline=0.35,
                                      formulas=0.3 if k % 2 else 0.0, skew=float(rng.uniform(-2.5, 2.5)) if k % 4 else 0.0,
                                      noise=float(rng.uniform(0.5, 4.0)))
    pages.append((augment_scan(pg, rng) if k % 2 else pg, tr))
res = eng.process_pages([PageImage(torch.from_numpy(pg).pin_memory(), "e", k) for k, (pg, _) in enumerate(pages)])
spans_of = lambda t: re.findall(r"\$([^$]*)\$", t)                # noqa: E731
strip = lambda t: re.sub(r"\s+", " ", re.sub(r"\$[^$]*\$", "$", t)).strip()      # noqa: E731
tot = dict(lines=0, lines_found=0, spans=0, spans_found=0, spans_false=0, exact=0, f_ed=0, f_ch=0, t_ed=0, t_ch=0,
           plain_lines=0, plain_false_spans=0, plain_ed=0, plain_ch=0)
rows = []
confs = []                       # (lowest formula confidence of a line read with formulas, the line has formulas)
for r, (pg, tr) in zip(res, pages):
    lines = [l for l in r.lines if l.kind != "equation"]
    lc = np.array([[0.5 * (l.box[0] + l.box[2]), 0.5 * (l.box[1] + l.box[3])] for l in lines], np.float64).reshape(-1, 2)
    rects = line_rects(tr, pg.shape)
    for li, (rc, t) in enumerate(zip(rects, tr.texts)):
        inline = li in tr.line_parts
        got, fconf = "", 1.0
        if lc.shape[0]:
            d = np.hypot(lc[:, 0] - rc[0], lc[:, 1] - rc[1])
            k = int(np.argmin(d))
            if d[k] <= 0.5 * rc[3] + 0.05 * rc[2]:
                got, fconf = lines[k].text, lines[k].formula_confidence
        if spans_of(got):
            confs.append((round(float(fconf), 3), bool(inline)))
        if not inline:
            tot["plain_lines"] += 1
            tot["plain_false_spans"] += len(spans_of(got))
            tot["plain_ed"] += _edit_distance(got, t)
            tot["plain_ch"] += len(t)
            if spans_of(got):
                rows.append({"false_span": True, "truth": t, "got": got})
            continue
        want, have = spans_of(t), spans_of(got)
        tot["lines"] += 1
        tot["lines_found"] += bool(got)
        tot["spans"] += len(want)
        tot["spans_found"] += min(len(want), len(have))
        tot["spans_false"] += max(0, len(have) - len(want))
        tot["t_ed"] += _edit_distance(strip(got), strip(t))
        tot["t_ch"] += len(strip(t))
        if len(want) == len(have):
            for a, b in zip(want, have):
                e = min(len(a), _edit_distance(b.strip(), a))
                tot["f_ed"] += e
                tot["f_ch"] += len(a)
                tot["exact"] += e == 0
        if got != t:
            rows.append({"skew": round(tr.angle, 2), "font": tr.font_names[0], "conf": round(float(fconf), 3), "truth": t, "got": got})
print(json.dumps({"pages": n, "fonts": "held out" if fonts is synth.HELD_OUT_FONTS else "training",
                  "lines_with_formulas": tot["lines"], "formulas": tot["spans"], "found": tot["spans_found"],
                  "false_in_those_lines": tot["spans_false"], "read_exactly": tot["exact"],
                  "formula_cer": round(tot["f_ed"] / max(1, tot["f_ch"]), 4),
                  "text_cer_of_those_lines": round(tot["t_ed"] / max(1, tot["t_ch"]), 4),
                  "other_lines": tot["plain_lines"], "false_in_other_lines": tot["plain_false_spans"],
                  "cer_other_lines": round(tot["plain_ed"] / max(1, tot["plain_ch"]), 4),
                  "confidence_of_lines_read_with_formulas": {
                      "true": sorted(c for c, t in confs if t), "false": sorted(c for c, t in confs if not t)}}))
for r_ in rows:
    print(json.dumps(r_, ensure_ascii=False))
