"""Continuously running engine loop (runtime/service.py) behind the inline extractor: concurrent documents share
page batches, results are routed back per document, a broken document fails alone, figures reach the image writer."""
import threading

import numpy as np
import pytest

from distributed_llms_amd.pdf import synth
from distributed_llms_amd.runtime.document import DocumentPipeline
from distributed_llms_amd.runtime.extractor import InlineExtractor
from distributed_llms_amd.runtime.workers import build_engine

SPEC = {"model": "tiny", "cfg": dict(det_limit_side=160, rec_max_width=256, rec_width_quantum=64, max_batch_pages=4)}


def _scan_pdf(n_pages, seed0=0):
    pages = [synth.render_page(seed0 + i, size=(248, 350), font_px=10, margin=16)[0] for i in range(n_pages)]
    return synth.make_scanned_pdf(pages, codec="flate", dpi=30)


@pytest.fixture(scope="module")
def extractor():
    ex = InlineExtractor(DocumentPipeline(build_engine("cpu", SPEC)), max_wait_ms=5.0, doc_deadline_s=120.0)
    yield ex
    ex.close()


def test_concurrent_documents_share_batches_and_keep_their_pages(extractor):
    pdfs = {f"doc{k}": _scan_pdf(1 + k % 3, seed0=10 * k) for k in range(8)}
    solo = {k: extractor.extract_document(v, k) for k, v in pdfs.items()}
    out, errs = {}, []

    def run(k):
        try:
            out[k] = extractor.extract_document(pdfs[k], k)
        except Exception as e:  # noqa: BLE001
            errs.append((k, e))

    th = [threading.Thread(target=run, args=(k,)) for k in pdfs]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errs
    for k in pdfs:
        assert [p.page_index for p in out[k].pages] == list(range(len(out[k].pages)))
        assert len(out[k].pages) == len(solo[k].pages)
        assert all(p.doc_id == k for p in out[k].pages)
        assert out[k].text == solo[k].text          # batching with other documents does not change a document's result
    st = extractor.describe()
    assert st["pushed"] == st["popped"] and st["documents_in_flight"] == 0
    assert st["mean_batch_pages"] > 1.0             # pages of different documents did share batches


def test_a_long_document_leaves_in_equal_batches(extractor):
    """10 pages at a limit of 4 per batch are 4 + 3 + 3, not 4 + 4 + 2."""
    svc = extractor.service
    assert svc.batcher.max_items == 4
    svc.recent_batch_sizes.clear()
    res = extractor.extract_document(_scan_pdf(10, seed0=300), "long")
    assert len(res.pages) == 10 and svc._outstanding == 0
    sizes = list(svc.recent_batch_sizes)
    assert sum(sizes) == 10 and max(sizes) <= 4, sizes
    if len(sizes) == 3:          # every page was decoded before the engine asked (a slow host may add deadline batches)
        assert max(sizes) - min(sizes) <= 1, sizes


def test_a_broken_document_fails_alone(extractor):
    good = _scan_pdf(2)
    res = {}

    def run(name, data):
        res[name] = extractor.extract(data, name)

    th = [threading.Thread(target=run, args=("bad", b"%PDF-1.4 no objects at all")),
          threading.Thread(target=run, args=("good", good))]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert res["bad"]["status"] == "failed" and res["good"]["status"] == "completed" and res["good"]["pages"] == 2
    assert extractor.service._outstanding == 0        # nothing is still counted as on its way to the engine


def test_process_pdfs_reports_errors_per_document():
    pipe = DocumentPipeline(build_engine("cpu", SPEC))
    try:
        docs = pipe.process_pdfs([(_scan_pdf(1), "a", None), (b"not a pdf", "b", None), (_scan_pdf(2), "c", None)])
    finally:
        pipe.close()
    assert [d.document_id for d in docs] == ["a", "b", "c"]
    assert docs[0].error is None and len(docs[0].pages) == 1
    assert docs[1].error and not docs[1].pages
    assert docs[2].error is None and len(docs[2].pages) == 2
