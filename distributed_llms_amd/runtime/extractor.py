"""Extraction back ends behind the job orchestrator (the L2 -> L3 boundary of SURVEY 1).

``extract(pdf_bytes, document_id)`` is the equivalent of the reference's
``process_pdf_sync`` (services/ocr_service/ocr_service.py:201-255) and returns the same
shape: ``{'status': 'completed', 'result': text}`` or ``{'status': 'failed', 'error': msg}``.

* ``InlineExtractor``  - engine inside the calling process (one GPU, or CPU for tests).
* ``PoolExtractor``    - front end parses + classifies (native, GIL-free), born-digital
  documents finish right there; scans are sharded page-wise over the GPU ``WorkerPool``.
"""
from __future__ import annotations

import logging
import threading
import time
from typing import Dict, List, Optional

from ..pdf import reader
from ..store.results import LocalFileWriter
from .document import DocumentPipeline, DocumentResult, ExtractionError, build_content_list, combine_text
from .engine import PageResult, TextLine

logger = logging.getLogger("distributed_llms_amd")


class Extractor:
    name = "base"

    def extract_document(self, pdf_bytes: bytes, document_id: str) -> DocumentResult:
        raise NotImplementedError

    def extract(self, pdf_bytes: bytes, document_id: str, document_hash: Optional[str] = None) -> Dict[str, str]:
        """Reference-shaped result dict; never raises (ocr_service.py:250-255)."""
        try:
            doc = self.extract_document(pdf_bytes, document_id)
            return {"status": "completed", "result": doc.text, "mode": doc.mode, "pages": len(doc.pages)}
        except Exception as e:  # noqa: BLE001 - every failure becomes a 'failed' status
            logger.error(f"Error processing PDF {document_id}: {e}")
            return {"status": "failed", "error": str(e)}

    def describe(self) -> Dict[str, object]:
        return {"backend": self.name}

    def close(self) -> None:
        pass


class InlineExtractor(Extractor):
    name = "inline"

    def __init__(self, pipeline: DocumentPipeline):
        self.pipeline = pipeline
        self._lock = threading.Lock()   # one document at a time on the engine's two streams

    def extract_document(self, pdf_bytes: bytes, document_id: str) -> DocumentResult:
        with self._lock:
            return self.pipeline.process_pdf(pdf_bytes, document_id)

    def describe(self):
        e = self.pipeline.engine
        return {"backend": self.name, "device": str(e.device), "pages_done": e.pages_done, "lines_done": e.lines_done}

    def close(self) -> None:
        self.pipeline.close()


def _text_layer_page(doc: reader.PdfDocument, i: int, document_id: str, include_invisible: bool) -> PageResult:
    pi = doc.page_info(i)
    pr = PageResult(document_id, i, int(pi.width_pt), int(pi.height_pt))
    y = 0.0
    for para in doc.page_text(i, include_invisible=include_invisible).split("\n\n"):
        para = " ".join(s.strip() for s in para.split("\n") if s.strip())
        if para:
            pr.lines.append(TextLine((0.0, y, 1.0, y + 1.0), para, 1.0))
            y += 3.0
    return pr


class PoolExtractor(Extractor):
    name = "gpu-pool"

    def __init__(self, pool, image_writer: Optional[LocalFileWriter] = None):
        self.pool = pool
        self.image_writer = image_writer

    def extract_document(self, pdf_bytes: bytes, document_id: str) -> DocumentResult:
        t0 = time.perf_counter()
        try:
            doc = reader.PdfDocument(pdf_bytes)
        except reader.PdfError as e:
            raise ExtractionError(f"cannot parse PDF: {e}") from e
        try:
            n = doc.page_count
            if n == 0:
                raise ExtractionError("PDF has no pages")
            cls = doc.classify()
            mode = "txt" if cls.is_text else "ocr"
            res = DocumentResult(document_id, mode)
            if mode == "txt":
                res.pages = [_text_layer_page(doc, i, document_id, False) for i in range(n)]
            else:
                with_image = [i for i in range(n) if doc.main_image(i) is not None]
                got = self.pool.submit(pdf_bytes, n, with_image).result(timeout=self.pool.doc_deadline_s + 5)
                pages: List[PageResult] = []
                for i in range(n):
                    pr = got.get(i)
                    pages.append(pr if pr is not None else _text_layer_page(doc, i, document_id, True))
                res.pages = pages
            res.content_list = build_content_list(res.pages)
            res.markdown = "\n\n".join(it["text"] for it in res.content_list)
            res.text = combine_text(res.content_list, res.markdown)
            res.timings = {"total_s": time.perf_counter() - t0}
            return res
        finally:
            doc.close()

    def describe(self):
        d = {"backend": self.name}
        d.update(self.pool.describe())
        return d

    def close(self) -> None:
        self.pool.stop()


def make_extractor(settings) -> Extractor:
    """Build the back end a ``Settings`` object asks for (OCR_NUM_GPUS, OCR_* knobs)."""
    import torch

    from .engine import EngineConfig, OcrEngine

    ngpu_visible = torch.cuda.device_count()
    want = settings.num_gpus if settings.num_gpus >= 0 else ngpu_visible
    ids = list(range(ngpu_visible))
    if getattr(settings, "gpu_ids", None):      # OCR_GPU_IDS: the subset of devices this front end drives
        ids = [int(t) for t in str(settings.gpu_ids).split(",") if t.strip() != "" and int(t) < ngpu_visible]
        if settings.num_gpus != 0:
            want = min(want, len(ids)) if settings.num_gpus > 0 else len(ids)
    cfg = dict(det_limit_side=settings.det_limit_side, rec_height=settings.rec_height,
               max_batch_pages=settings.max_batch_pages, use_hip_graphs=settings.use_hip_graphs)
    if want <= 0 or ngpu_visible == 0:
        logger.warning("no GPU requested/visible: OCR runs on the CPU oracle path (slow; text PDFs are unaffected)")
        spec_model = getattr(settings, "cpu_model", "full")
        from .workers import build_engine

        engine = build_engine("cpu", {"model": spec_model, "cfg": cfg, "weights_dir": settings.weights_dir, "recognizer": settings.recognizer})
        return InlineExtractor(DocumentPipeline(engine))
    if want == 1 and not getattr(settings, "force_pool", False):
        from .workers import build_engine

        engine = build_engine(f"cuda:{ids[0]}", {"model": "full", "cfg": cfg, "weights_dir": settings.weights_dir, "recognizer": settings.recognizer})
        engine.warmup(batch=min(4, settings.max_batch_pages))
        return InlineExtractor(DocumentPipeline(engine))
    from .workers import WorkerPool

    pool = WorkerPool([f"cuda:{i}" for i in ids[: min(want, len(ids))]],
                      {"model": "full", "cfg": cfg, "weights_dir": settings.weights_dir, "recognizer": settings.recognizer},
                      fault_inject=settings.fault_inject, max_wait_ms=settings.max_wait_ms,
                      hbm_budget_gb=settings.hbm_budget_gb, doc_deadline_s=settings.doc_deadline_s).start()
    return PoolExtractor(pool)
