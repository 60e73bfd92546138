"""Layout analysis (runtime/layout.py + csrc/runtime/layout.cpp): reading order on multi-column pages, headers /
footers, figures, label-value rows; native implementation against the Python oracle; the born-digital (txt) branch
of the document pipeline on a two-column PDF whose content stream interleaves the columns."""
import numpy as np
import pytest

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
