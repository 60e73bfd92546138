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
        frm = self.truth.formulas
        if self.truth.inline.shape[0]:                # formulas inside lines: the band the trainer paints
            q = self.truth.inline.copy()
            hq = q[:, 3] - q[:, 1]
            q[:, 1] += 0.15 * hq
            q[:, 3] -= 0.15 * hq
            frm = np.concatenate([frm.reshape(-1, 4), q], 0)
        for k, boxes in enumerate((self.truth.figures, frm)):
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


@pytest.mark.parametrize("kw", [dict(), dict(rows=4, cols=2), dict(row_pitch=2.6), dict(column=True, cols=3),
                                dict(rows=8, cols=5, ruled=False)])
def test_tables_are_found_and_read_row_by_row(kw):
    """A grid of cells is one block of kind 2 in row-major order - with tight rows, with rows a block gap apart, and
    inside one column of a two-column page - in the native implementation and in the Python oracle alike."""
    for seed in range(4):
        arr, tr, grid = synth.render_table_page(40 + seed, size=(1240, 1754), **kw)
        a = layout.order_page(tr.boxes, arr.shape[0])
        b = layout.order_page_native(tr.boxes, float(arr.shape[0]))
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        order, block, kind = b
        assert order.tolist() == list(range(len(tr.texts)))
        cells = [tr.texts[i] for i, k in zip(order, kind) if k == 2]
        assert cells == [c for row in grid for c in row]
        assert len({int(bk) for bk, k in zip(block, kind) if k == 2}) == 1
        # boxes as a detector reports them: taller than the ink and a little wider
        hm = float(np.median(tr.boxes[:, 3] - tr.boxes[:, 1]))
        loose = tr.boxes + np.array([-0.05, -0.12, 0.05, 0.12], np.float32) * hm
        a, b = layout.order_page(loose, arr.shape[0]), layout.order_page_native(loose, float(arr.shape[0]))
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert b[0].tolist() == list(range(len(tr.texts)))
        assert [tr.texts[i] for i, k in zip(b[0], b[2]) if k == 2] == [c for row in grid for c in row]


def test_table_becomes_one_content_item_and_survives_the_wire():
    from distributed_llms_amd.runtime.document import page_content_items
    from distributed_llms_amd.runtime.engine import PageResult, TextLine
    from distributed_llms_amd.runtime.wire import PackedShard, pack_shard

    arr, tr, grid = synth.render_table_page(7, size=(1240, 1754), rows=4, cols=3)
    grid[2][1] = "a | b <c>"                       # separators and markup inside a cell must not break anything
    flat = [c for row in grid for c in row]
    order, block, kind = layout.order_page_native(tr.boxes, float(arr.shape[0]))
    pr = PageResult("d", 0, arr.shape[1], arr.shape[0])
    n_before = tr.texts.index(grid[0][0])
    texts = tr.texts[:n_before] + flat + tr.texts[n_before + len(flat):]
    for k, i in enumerate(order):
        pr.lines.append(TextLine(tuple(float(v) for v in tr.boxes[i]), texts[i], 0.9, int(block[k]), 0.0,
                                 "cell" if kind[k] == 2 else "text"))
    items = page_content_items(pr)
    tabs = [it for it in items if it["type"] == "table"]
    assert len(tabs) == 1 and tabs[0]["cells"] == grid
    assert tabs[0]["table_body"].count("<tr>") == 4 and tabs[0]["table_body"].count("<td>") == 12
    assert "a | b &lt;c&gt;" in tabs[0]["table_body"]
    assert tabs[0]["text"].split("\n")[0] == " | ".join(grid[0])
    kinds = [it["type"] for it in items]
    k = kinds.index("table")
    assert 0 < k < len(kinds) - 1 and set(kinds[:k]) == {"text"} and set(kinds[k + 1:]) == {"text"}
    sh = PackedShard(pack_shard([(0, pr)]))
    back = sh.items(0)
    assert [it["type"] for it in back] == kinds
    assert back[k]["cells"] == grid and back[k]["table_body"] == tabs[0]["table_body"] and back[k]["text"] == tabs[0]["text"]
    assert [l.kind for l in sh.page_result(0).lines] == [l.kind for l in pr.lines]


def test_running_headers_and_footers_leave_the_content_list():
    """Lines in the margin bands that repeat across pages (digits aside) are dropped from the content list and kept
    in ``discarded``; a margin line that does not repeat stays, and a one-page document keeps everything."""
    from distributed_llms_amd.runtime.document import assemble_document, assemble_packed
    from distributed_llms_amd.runtime.engine import PageResult, TextLine
    from distributed_llms_amd.runtime.wire import PackedShard, pack_shard

    def page(i, header, footer):
        pr = PageResult("doc", i, 1000, 1400)
        pr.lines.append(TextLine((100, 40, 600, 70), header, 0.9, 0))
        pr.lines.append(TextLine((100, 200, 900, 240), f"body line one of page {i}", 0.9, 1))
        pr.lines.append(TextLine((100, 245, 900, 285), "and its second line", 0.9, 1))
        pr.lines.append(TextLine((450, 1330, 550, 1360), footer, 0.9, 2))
        return pr

    pages = [page(i, "Annual Report 2024" if i != 2 else "Chapter Three Begins", f"Page {i + 1} of 5") for i in range(5)]
    res = assemble_document("doc", "ocr", pages)
    texts = [it["text"] for it in res.content_list]
    assert not any("Annual Report" in t or t.startswith("Page ") for t in texts)
    assert "Chapter Three Begins" in texts                               # a margin line that does not repeat is content
    assert sum(1 for t in texts if t.startswith("body line one")) == 5
    assert sorted(d["type"] for d in res.discarded) == ["footer"] * 5 + ["header"] * 4
    assert all("margin" not in it for it in res.content_list)
    assert "Annual Report" not in res.text and "Chapter Three Begins" in res.text
    # the packed path (GPU workers -> front end) decides over the whole document too, not per shard
    shards = [PackedShard(pack_shard([(i, pages[i]) for i in idx])) for idx in ((0, 1), (2, 3, 4))]
    res2 = assemble_packed("doc", "ocr", shards, {}, 5)
    assert [it["text"] for it in res2.content_list] == texts and res2.discarded == res.discarded
    one = assemble_document("doc", "ocr", pages[:1])
    assert any("Annual Report" in it["text"] for it in one.content_list) and not one.discarded


def test_table_in_a_born_digital_pdf_becomes_a_table_item():
    """The text-layer branch runs the same layout analysis: positioned cells of a born-digital page (emitted row by
    row, each cell its own text-showing operator) come back as one table with its rows and columns; header and title
    stay outside of it."""
    from distributed_llms_amd.runtime.document import page_content_items

    cols = [["Item", "Paper A4", "Toner black", "Stapler", "Binder"], ["Qty", "12", "3", "1", "40"],
            ["Price", "4.50", "61.00", "8.99", "1.20"], ["Total", "54.00", "183.00", "8.99", "48.00"]]
    pdf = synth.make_text_pdf_columns(cols, title="Order 2024-117", header="ACME Office Supplies")
    with reader.PdfDocument(pdf) as doc:
        pr = text_layer_page(doc, 0, "t")
    items = page_content_items(pr)
    tabs = [it for it in items if it["type"] == "table"]
    assert len(tabs) == 1, [(it["type"], it.get("text")) for it in items]
    assert tabs[0]["cells"] == [[c[r] for c in cols] for r in range(5)]
    assert "<td>Toner black</td><td>3</td><td>61.00</td><td>183.00</td>" in tabs[0]["table_body"]
    texts = [it["text"] for it in items if it["type"] == "text"]
    assert "Order 2024-117" in texts and "ACME Office Supplies" in texts


def test_one_overlong_cell_does_not_glue_two_columns_together():
    """Detector boxes of an eight-row table inside the left column of a two-column page, row three's first cell running
    up to its neighbour (25 pixels apart): still one table of 8 x 3, in both implementations, and the grid keeps the
    long cell in its own column."""
    from distributed_llms_amd.runtime.document import table_grid
    from distributed_llms_amd.runtime.engine import TextLine

    para = [[187, 189, 1115, 253], [184, 255, 1104, 319], [182, 315, 1112, 386], [186, 387, 951, 445]]
    cells = [[192, 515, 368, 575], [512, 515, 737, 575], [831, 515, 1050, 575], [189, 659, 441, 726], [508, 660, 712, 725],
             [844, 669, 996, 722], [189, 808, 470, 875], [495, 808, 720, 874], [833, 812, 1004, 876], [189, 957, 447, 1023],
             [509, 958, 686, 1022], [838, 966, 1004, 1019], [194, 1115, 338, 1168], [511, 1106, 712, 1172],
             [828, 1110, 1027, 1170], [187, 1257, 399, 1323], [508, 1251, 687, 1321], [826, 1251, 1030, 1322],
             [197, 1408, 406, 1468], [514, 1412, 650, 1469], [837, 1400, 1005, 1470], [192, 1559, 368, 1619],
             [508, 1552, 710, 1618], [846, 1561, 1001, 1614]]
    after = [[182, 1767, 1059, 1837], [187, 1836, 1089, 1900], [187, 1902, 1110, 1966]]
    right = [[1308, 517 + 60 * k, 2200, 575 + 60 * k] for k in range(14)]
    boxes = np.asarray(para + cells + after + right, np.float32)
    a, b = layout.order_page(boxes, 3508.0), layout.order_page_native(boxes, 3508.0)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    order, block, kind = b
    assert [int(i) for i, k in zip(order, kind) if k == 2] == list(range(4, 28))
    grid = table_grid([TextLine(tuple(float(v) for v in bx), f"c{i}", 1.0, 1, 0.0, "cell") for i, bx in enumerate(cells)])
    assert grid == [[f"c{3 * r + c}" for c in range(3)] for r in range(8)]


def test_formulas_inside_lines_are_cut_out_and_put_back_in_place():
    """A page whose paragraphs carry formulas inside the lines: every such line comes back as ONE text line whose text
    holds the formulas as $...$ between the stretches of text, in order; display formulas stay equations."""
    from distributed_llms_amd.models.crnn import CRNN
    from distributed_llms_amd.models.trocr import TrOCR, TrOCRConfig
    from distributed_llms_amd.runtime import engine as E

    pg, tr = synth.render_layout_page(21, size=(1240, 1754), layout="single", formulas=0.5, inline=0.5, noise=1.0)
    assert tr.inline.shape[0] >= 4 and len(tr.line_parts) >= 3
    torch.manual_seed(0)
    frm = TrOCR(TrOCRConfig(image_size=192, enc_dim=64, enc_layers=1, enc_heads=1, enc_mlp=128, dec_dim=64, dec_layers=1,
                            dec_heads=1, dec_ffn=128, vocab=272, max_len=24))
    eng = E.OcrEngine("cpu", _PaintedDetector(tr, pg.shape), CRNN(hidden=64, widths=(16, 32, 32, 32, 64, 64)),
                      E.EngineConfig(det_limit_side=1344, max_batch_pages=1, formula_max_tokens=12, formula_min_side_px=4,
                                     inline_formulas=True, inline_min_conf=1e-9),
                      formula_rec=frm)
    eng._strings = lambda ids, lens: ["T"] * ids.shape[0]
    frm.tokens_to_text = lambda toks, *a: ["M"] * len(toks)
    # "formulas" without a command, subscript or superscript are what the text recogniser reads as well: whole lines
    res = eng.process_pages([E.PageImage(torch.from_numpy(pg.copy()), "f", 0)])[0]
    assert all(l.text == "T" for l in res.lines if l.kind == "text")
    frm.tokens_to_text = lambda toks, *a: ["\\M"] * len(toks)
    res = eng.process_pages([E.PageImage(torch.from_numpy(pg.copy()), "f", 0)])[0]
    text_lines = [l for l in res.lines if l.kind == "text"]
    assert len(text_lines) == len(tr.texts)
    assert sum(l.kind == "equation" for l in res.lines) == tr.formulas.shape[0]
    for li, (l, t) in enumerate(zip(text_lines, tr.texts)):
        if li in tr.line_parts:
            want = " ".join("T" if k == "t" else "$\\M$" for k, *_ in tr.line_parts[li])
            assert l.text == want, (li, l.text, want, t)
            tb = tr.boxes[li]
            assert abs(l.box[0] - tb[0]) < 12 and abs(l.box[2] - tb[2]) < 12
        else:
            assert l.text == "T"
    # formulas read with little confidence: their lines fall back to the whole line read as text
    eng.cfg.inline_min_conf = 2.0
    res = eng.process_pages([E.PageImage(torch.from_numpy(pg.copy()), "f", 0)])[0]
    text_lines = [l for l in res.lines if l.kind == "text"]
    assert len(text_lines) == len(tr.texts) and all(l.text == "T" for l in text_lines)
    for li in tr.line_parts:
        assert abs(text_lines[li].box[0] - tr.boxes[li][0]) < 12 and abs(text_lines[li].box[2] - tr.boxes[li][2]) < 12


def test_inline_assignment_and_split_follow_a_skewed_line():
    from distributed_llms_amd.runtime import engine as E

    th = np.deg2rad(3.0)
    line = np.array([[100, 500, 1900, 560, 0.9, 0, th, 0]], np.float32)         # centre (1000, 530), turned 3 degrees
    a = 600.0                                                                     # a formula 600 px right of the centre
    q = np.array([[1000 + a * np.cos(th) - 80, 530 + a * np.sin(th) - 15, 1000 + a * np.cos(th) + 80, 530 + a * np.sin(th) + 15, 0.8],
                  [300, 900, 900, 1000, 0.9]], np.float32)                        # and a display formula elsewhere
    spans, taken = E.assign_inline_formulas(line, q)
    assert taken.tolist() == [True, False] and abs(spans[0][0][0] - 520) < 1 and abs(spans[0][0][1] - 680) < 1
    parts = E.split_line_at_formulas(line[0], spans[0])
    assert parts[:, 7].tolist() == [E.KIND_TEXT, E.KIND_MATH_CONT, E.KIND_MATH_TEXT, E.KIND_TEXT_CONT, E.KIND_LINE_ALT]
    assert np.array_equal(parts[-1, :7], line[0, :7])                      # the whole line once more, as the fallback
    assert np.array_equal(parts[2, :7], parts[1, :7])                      # the formula's crop, for the text recogniser
    parts = parts[[0, 1, 3]]
    mid = 0.5 * (parts[1, :2] + parts[1, 2:4])
    assert np.allclose(mid, [1000 + a * np.cos(th), 530 + a * np.sin(th)], atol=0.5)
    assert np.allclose(parts[:, 3] - parts[:, 1], 60) and abs((parts[:, 2] - parts[:, 0]).sum() - 1800) < 1e-3
    # a line that begins with the formula; a sliver of text at the end is dropped
    parts = E.split_line_at_formulas(line[0], [(-900.0, -700.0, 0.7), (700.0, 885.0, 0.6)])
    assert parts[:, 7].tolist() == [E.KIND_MATH_HEAD, E.KIND_MATH_TEXT, E.KIND_TEXT_CONT, E.KIND_MATH_CONT, E.KIND_MATH_TEXT,
                                    E.KIND_LINE_ALT]


def test_a_formula_just_beyond_a_lines_end_joins_that_line():
    """The text detector tends to stop where the words stop: a formula on the same baseline within a line height of the
    line's end is the line's first / last piece, and the line (and its fallback) grows to hold it."""
    from distributed_llms_amd.runtime import engine as E

    line = np.array([[400, 500, 1900, 560, 0.9, 0, 0.0, 0]], np.float32)
    q = np.array([[150, 510, 380, 550, 0.8], [1930, 505, 2100, 555, 0.8], [150, 700, 380, 750, 0.8]], np.float32)
    spans, taken = E.assign_inline_formulas(line, q)
    assert taken.tolist() == [True, True, False]                 # the third lies a line further down: a display formula
    parts = E.split_line_at_formulas(line[0], spans[0])
    assert parts[:, 7].tolist() == [E.KIND_MATH_HEAD, E.KIND_MATH_TEXT, E.KIND_TEXT_CONT, E.KIND_MATH_CONT, E.KIND_MATH_TEXT,
                                    E.KIND_LINE_ALT]
    assert parts[:, 0].tolist() == [150, 150, 380, 1930, 1930, 150] and parts[:, 2].tolist() == [380, 380, 1930, 2100, 2100, 2100]
    # the pieces of a display formula that the text detector reports lie inside the formula's box: they do not claim it
    piece = np.array([[500, 1010, 700, 1060, 0.9, 0, 0.0, 0]], np.float32)
    spans, taken = E.assign_inline_formulas(piece, np.array([[400, 1000, 900, 1070, 0.9]], np.float32))
    assert not spans and not taken.any()


def test_widest_gap_picks_the_word_space_not_a_letter_gap():
    from distributed_llms_amd.runtime.engine import widest_gap

    prof = np.full(60, 240, np.uint8)
    prof[0:10] = 30; prof[12:22] = 40          # letters with a 2-pixel gap
    prof[34:50] = 35                           # next glyph after a 12-pixel space
    prof[52:60] = 30
    assert widest_gap(prof, 30) in (27, 28)
    assert widest_gap(np.full(40, 240, np.uint8), 20) is None


@pytest.mark.parametrize("kw", [dict(cols=3, rows=5), dict(column=True, cols=3, rows=6), dict(cols=4, rows=8, row_pitch=2.2),
                                dict(column=True, cols=2, rows=9, row_pitch=1.4)])
def test_table_cells_that_wrap_onto_a_second_line_join_the_cell_above(kw):
    """Labels that run onto a second line (at line spacing under the first, the next row further down): the table is
    still found as one table, in both implementations, the extra lines come in reading order after their row, and the
    grid holds each wrapped label as ONE cell."""
    from distributed_llms_amd.runtime.document import table_grid
    from distributed_llms_amd.runtime.engine import TextLine

    for seed in (711, 712, 713):
        arr, tr, grid = synth.render_table_page(seed, wrap=0.5, **kw)
        b = tr.boxes.copy()
        b[:, 1] -= 6          # detector boxes are taller than the ink
        b[:, 3] += 6
        a, c = layout.order_page(b, float(arr.shape[0])), layout.order_page_native(b, float(arr.shape[0]))
        assert all(np.array_equal(x, y) for x, y in zip(a, c))
        order, block, kind = c
        assert order.tolist() == list(range(len(tr.texts)))
        cells = [TextLine(tuple(b[i]), tr.texts[i], 1.0, int(block[i]), 0.0, "cell") for i, k in zip(order, kind) if k == 2]
        assert len(cells) > sum(len(r) for r in grid) or seed != 711          # (seed 711 wraps at least one label)
        assert table_grid(cells) == grid, (seed, kw)


def test_a_formula_that_does_not_count_falls_back_alone_and_the_line_whole_when_none_counts():
    """``OcrEngine._join_pieces``: of two formulas cut out of a line, the one read without any command, subscript or
    superscript is replaced by the same crop read as text while the other stays ``$...$``; when neither counts the
    line is what the text recogniser read whole; in a line of Greek prose a formula of Greek letters and operators
    alone does not count."""
    import types

    from distributed_llms_amd.runtime import engine as E

    line = np.array([100, 500, 1900, 560, 0.9, 3, 0.0, 0], np.float32)
    rows = E.split_line_at_formulas(line, [(-600.0, -400.0, 0.8), (200.0, 450.0, 0.9)])
    kinds = rows[:, 7].astype(int)
    assert kinds.tolist() == [E.KIND_TEXT, E.KIND_MATH_CONT, E.KIND_MATH_TEXT, E.KIND_TEXT_CONT, E.KIND_MATH_CONT, E.KIND_MATH_TEXT,
                              E.KIND_TEXT_CONT, E.KIND_LINE_ALT]
    b4 = rows[:, [0, 1, 2, 3, 6]]
    blk = rows[:, 5].astype(int)
    ok = np.ones(len(rows), bool)
    me = types.SimpleNamespace(cfg=E.EngineConfig(inline_formulas=True))
    join = lambda texts, confs: E.OcrEngine._join_pieces(me, range(len(rows)), 0, b4, blk, kinds, ok, texts, np.asarray(confs, np.float32))   # noqa: E731
    texts = ["we take", "x_{i}^{2}", "xi2", "and", "x = 3", "x=3", "as given", "we take xi2 and x=3 as given"]
    got = join(texts, [0.9, 0.97, 0.5, 0.9, 0.99, 0.8, 0.9, 0.85])
    assert got.text == "we take $x_{i}^{2}$ and x=3 as given" and got.kind == "text" and got.block == 3
    assert abs(got.box[0] - 100) < 1e-3 and abs(got.box[2] - 1900) < 1e-3 and abs(got.formula_confidence - 0.97) < 1e-6
    # the first formula read with little confidence as well: nothing counts, the whole line as read in one piece
    got = join(texts, [0.9, 0.4, 0.5, 0.9, 0.99, 0.8, 0.9, 0.85])
    assert got.text == "we take xi2 and x=3 as given" and got.formula_confidence == 1.0
    # Greek prose: Greek letters and an operator are text there, a subscript still makes a formula
    texts = ["το", "\\alpha \\cdot \\beta", "αβ", "και", "\\lambda_{i}", "λi", "είναι", "το αβ και λi είναι"]
    got = join(texts, [0.9, 0.99, 0.9, 0.9, 0.99, 0.8, 0.9, 0.9])
    assert got.text == "το αβ και $\\lambda_{i}$ είναι"


def test_low_confidence_pages_are_flagged_in_the_documents_warnings():
    from distributed_llms_amd.runtime.document import assemble_document, low_confidence_warning
    from distributed_llms_amd.runtime.engine import PageResult, TextLine

    assert low_confidence_warning(3, [0.99, 0.98, 0.97, 0.99, 0.99, 0.5]) is None           # one bad line is no bad page
    assert low_confidence_warning(3, [0.7, 0.75]) is None                                    # too few lines to tell
    w = low_confidence_warning(3, [0.7, 0.75, 0.8, 0.72, 0.69, 0.81])
    assert w.startswith("page 3:") and "0.73" in w or "0.74" in w

    def page(i, conf):
        p = PageResult("d", i, 1000, 1400)
        p.lines = [TextLine((50, 100 + 40 * k, 900, 130 + 40 * k), f"line {k} of the page", conf, 0) for k in range(8)]
        return p

    doc = assemble_document("d", "ocr", [page(0, 0.99), page(1, 0.72)])
    assert len(doc.warnings) == 1 and doc.warnings[0].startswith("page 2: text read with low confidence")


def test_native_and_oracle_agree_on_generated_pages_with_wrapped_and_missing_cells():
    """120 generated pages (text columns; tables under a paragraph with 2 - 5 columns, wrapped first cells, holes, row
    pitch from tight to wide): the C++ analysis and the Python oracle give the same order, blocks and kinds, and most of
    the tables are found."""
    rng = np.random.default_rng(5)
    tables = found = 0
    for it in range(120):
        bx = []
        h = rng.uniform(20, 70)
        mode = it % 4
        if mode in (0, 1):
            for c in range(1 + mode):
                y = 200.0
                while y < 3000:
                    bx.append([200 + c * 1100, y, 200 + c * 1100 + rng.uniform(300, 1000), y + h])
                    y += h * rng.uniform(1.3, 1.8)
        else:
            y = 200.0
            for _ in range(4):
                bx.append([200, y, 2200, y + h])
                y += h * 1.5
            cols, rows = int(rng.integers(2, 6)), int(rng.integers(3, 12))
            cw, pitch = 2000 / cols, h * rng.uniform(1.3, 2.6)
            for _ in range(rows):
                for c in range(cols):
                    if mode == 2 or rng.integers(7):
                        bx.append([200 + c * cw + 10, y, 200 + c * cw + rng.uniform(40, cw - 30), y + h])
                if rng.integers(3) == 0:
                    y += h * 1.25
                    bx.append([210, y, 200 + rng.uniform(40, cw - 30), y + h])
                y += pitch
            for _ in range(3):
                bx.append([200, y + h, 2200, y + 2 * h])
                y += h * 1.5
            tables += 1
        b = np.asarray(bx, np.float32)
        a, c = layout.order_page(b, 3508.0), layout.order_page_native(b, 3508.0)
        assert all(np.array_equal(x, y) for x, y in zip(a, c)), it
        assert sorted(c[0].tolist()) == list(range(len(bx)))
        found += int(mode >= 2 and (c[2] == 2).any())
        assert mode >= 2 or not (c[2] == 2).any(), it                 # columns of running text are no table
    assert found >= 0.8 * tables, (found, tables)
