"""Layout analysis (runtime/layout.py + csrc/runtime/layout.cpp): reading order on multi-column pages, headers /
footers, figures, label-value rows; native implementation against the Python oracle; the born-digital (txt) branch
of the document pipeline on a two-column PDF whose content stream interleaves the columns."""
import numpy as np
import pytest
import torch

from distributed_llms_amd.pdf import reader, synth
from distributed_llms_amd.runtime import layout
from distributed_llms_amd.runtime.document import DocumentPipeline, text_layer_page


@pytest.mark.parametrize("lay", synth.LAYOUTS)
def test_xy_cut_recovers_the_generators_reading_order(lay):
    for seed in range(12):
        arr, tr = synth.render_layout_page(1000 + seed, size=(620, 877), layout=lay)
        order, block, kind = layout.order_page(tr.boxes, arr.shape[0], tr.figures)
        assert [int(i) for i, k in zip(order, kind) if k == 0] == list(range(len(tr.texts))), (lay, seed)
        assert int(kind.sum()) == len(tr.figures)
        if lay in ("two_col", "two_col_title"):
            assert len(set(block.tolist())) >= 2


def test_native_layout_equals_python_oracle():
    rng = np.random.default_rng(3)
    for seed in range(40):
        arr, tr = synth.render_layout_page(2000 + seed, size=(620, 877))
        boxes = tr.boxes + rng.uniform(-1.5, 1.5, size=tr.boxes.shape).astype(np.float32)   # detector-like jitter
        a = layout.order_page(boxes, 877.0, tr.figures)
        b = layout.order_page_native(boxes, 877.0, tr.figures)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    # degenerate inputs
    for n in (0, 1, 2):
        boxes = np.array([[10, 10 + 30 * i, 200, 30 + 30 * i] for i in range(n)], np.float32).reshape(-1, 4)
        a, b = layout.order_page(boxes, 800.0), layout.order_page_native(boxes, 800.0)
        assert np.array_equal(a[0], b[0]) and a[0].tolist() == list(range(n))


def test_label_value_rows_are_not_cut_into_columns():
    """An invoice-like block: labels on the left, right-aligned amounts far to the right.  The wide gap between them
    is not a gutter between text columns - each row reads label, then amount."""
    boxes = []
    for r in range(8):
        y = 100 + 40 * r
        boxes.append([60, y, 60 + 90 + 10 * (r % 3), y + 28])      # "Total:", "Amount due:" ...
        boxes.append([520, y, 600, y + 28])                        # "1,250.00"
    order, _, _ = layout.order_page(np.asarray(boxes, np.float32), 1000.0)
    assert order.tolist() == list(range(16))


def test_header_and_footer_bracket_the_body():
    body = [[100, 150 + 30 * i, 500, 172 + 30 * i] for i in range(10)]
    boxes = np.asarray(body + [[250, 940, 300, 960]] + [[100, 20, 400, 40]], np.float32)   # footer, then header
    order, block, _ = layout.order_page(boxes, 1000.0)
    assert order[0] == 11 and order[-1] == 10 and order[1:-1].tolist() == list(range(10))
    assert block[0] != block[1] and block[-1] != block[-2]


def test_two_column_born_digital_pdf_reads_column_major():
    left = [f"Left column line {i} with several words" for i in range(12)]
    right = [f"Right column line {i} continues the story" for i in range(12)]
    pdf = synth.make_text_pdf_columns([left, right], title="A Full Width Title Above Both Columns",
                                      header="Journal of Synthetic Documents", footer="7")
    with reader.PdfDocument(pdf) as doc:
        assert doc.classify().is_text
        recs = doc.page_lines(0)
        # the content stream interleaves the columns row by row, and so does the reader's raw line list
        raw = [t for _, _, t in recs]
        assert raw.index(left[1]) > raw.index(right[0])
        pr = text_layer_page(doc, 0, "d")
    got = [ln.text for ln in pr.lines]
    want = ["Journal of Synthetic Documents", "A Full Width Title Above Both Columns"] + left + right + ["7"]
    assert got == want


def test_txt_branch_of_the_pipeline_uses_layout_order():
    from distributed_llms_amd.runtime.workers import build_engine

    left = [f"Alpha {i} lorem ipsum dolor sit amet consect" for i in range(10)]
    right = [f"Beta {i} sed do eiusmod tempor incididunt" for i in range(10)]
    pdf = synth.make_text_pdf_columns([left, right], title="Two Columns")
    engine = build_engine("cpu", {"model": "tiny", "cfg": dict(det_limit_side=160, rec_max_width=256, rec_width_quantum=64)})
    pipe = DocumentPipeline(engine)
    try:
        doc = pipe.process_pdf(pdf, "txt2")
    finally:
        pipe.close()
    assert doc.mode == "txt"
    text = doc.text
    assert text.index("Alpha 9") < text.index("Beta 0"), "left column must be read to its end before the right one"
    assert text.startswith("Two Columns")
    # the title, the left column and the right column are separate paragraphs
    assert len(doc.content_list) >= 3


class _PaintedDetector(torch.nn.Module):
    """Stands in for the trained detector on CPU: paints the maps a perfect detector would emit from the page truth
    (text kernels, figure mask, formula mask), so the engine's box / ordering / crop / recognition plumbing runs."""

    def __init__(self, truth, page_hw):
        super().__init__()
        self.truth, self.page_hw = truth, page_hw
        self.fig_head = torch.nn.Identity()
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def prepare(self):
        return self

    def infer(self, x, concat_buf=None, with_figures=False):
        from distributed_llms_amd.train.page_trainer import shrink_boxes

        B, H, W, _ = x.shape
        sc = H / self.page_hw[0]
        prob = torch.zeros((B, H, W))
        for b in shrink_boxes(self.truth.boxes) * sc:
            x0, y0, x1, y1 = [int(round(float(v))) for v in b]
            prob[:, y0:max(y1, y0 + 1), x0:max(x1, x0 + 1)] = 0.9
        for b in self.truth.formulas * sc:            # a real text head also fires on formula pieces
            x0, y0, x1, y1 = [int(round(float(v))) for v in b]
            prob[:, (y0 + y1) // 2 - 2:(y0 + y1) // 2 + 2, x0 + 4:x1 - 4] = 0.9
        fig = torch.full((B, H // 4, W // 4, 8), -6.0)
        for k, boxes in enumerate((self.truth.figures, self.truth.formulas)):
            for b in boxes * sc / 4:
                x0, y0, x1, y1 = [int(round(float(v))) for v in b]
                fig[:, y0:y1, x0:x1, k] = 6.0
        out = (prob.to(torch.bfloat16), (prob > 0.3).to(torch.uint8))
        return out + (fig.to(torch.bfloat16),) if with_figures else out


def test_formulas_become_equation_items_in_reading_order():
    from distributed_llms_amd.models.crnn import CRNN
    from distributed_llms_amd.models.trocr import TrOCR, TrOCRConfig
    from distributed_llms_amd.runtime.document import build_content_list
    from distributed_llms_amd.runtime.engine import EngineConfig, OcrEngine, PageImage

    pg, tr = synth.render_layout_page(11, size=(1240, 1754), layout="single", formulas=0.9, noise=1.0)
    assert tr.formulas.shape[0] >= 2
    torch.manual_seed(0)
    frm = TrOCR(TrOCRConfig(image_size=192, enc_dim=64, enc_layers=1, enc_heads=1, enc_mlp=128, dec_dim=64, dec_layers=1,
                            dec_heads=1, dec_ffn=128, vocab=272, max_len=24))
    eng = OcrEngine("cpu", _PaintedDetector(tr, pg.shape), CRNN(hidden=64, widths=(16, 32, 32, 32, 64, 64)),
                    EngineConfig(det_limit_side=1344, max_batch_pages=1, formula_max_tokens=12, formula_min_side_px=6), formula_rec=frm)
    res = eng.process_pages([PageImage(torch.from_numpy(pg.copy()), "f", 0)])[0]
    kinds = [l.kind for l in res.lines]
    assert kinds.count("equation") == tr.formulas.shape[0] and kinds.count("text") == len(tr.texts)
    # every formula sits where the truth has it: after `formula_pos` text lines
    seen_text = 0
    got_pos = []
    for l in res.lines:
        if l.kind == "equation":
            got_pos.append(seen_text)
        else:
            seen_text += 1
    assert got_pos == tr.formula_pos
    for l, t in zip([l for l in res.lines if l.kind == "equation"], tr.formulas):
        ix = max(0.0, min(l.box[2], t[2]) - max(l.box[0], t[0])) * max(0.0, min(l.box[3], t[3]) - max(l.box[1], t[1]))
        assert ix / ((t[2] - t[0]) * (t[3] - t[1])) > 0.7
    items = build_content_list([res])
    eq = [it for it in items if it["type"] == "equation"]
    assert len(eq) == tr.formulas.shape[0] and all(it["text_format"] == "latex" and "bbox" in it for it in eq)
    # equations split the text flow: no paragraph spans a formula
    order = [it["type"] for it in items]
    assert order[0] == "text" and "equation" in order[1:-1]
