"""Document job state machine.

Lifecycle in the status store:  queued -> processing -> completed | failed
(services/ocr_service/ocr_service.py:453, 265, 292/376, 305/328).  The write order on
success is the reference's: result-store upsert -> status {completed, result_preview}
-> status {completed, document_hash, result (full text)}; the result row therefore exists
before any reader can see ``completed``.  Failures write ``failed`` + error and drop
``output/error_{id}.json`` for the log agent.

What is new relative to the reference:
* the ``queued`` claim is atomic (``SET NX``), so two replicas cannot both start one id;
* status records carry ``owner`` (node id) so a restarted node can fail its orphans;
* a per-document deadline turns a stuck extraction into ``failed`` instead of hanging forever;
* every status write and store call runs in the I/O thread pool, never on the event loop;
* per-stage latencies are recorded for /metrics (5.1, 5.5).
"""
from __future__ import annotations

import asyncio
import hashlib
import datetime
import logging
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Optional, Set

from ..config import IO_THREADS, PREVIEW_CHARS
from ..store.results import ErrorSink, ResultStore, make_result_document
from ..store.status import StatusStore
from ..utils import metrics

logger = logging.getLogger("distributed_llms_amd")


def _now() -> str:
    return datetime.datetime.now().isoformat()


class JobManager:
    def __init__(self, status: StatusStore, results: ResultStore, errors: ErrorSink, extractor,
                 node_id: str = "", deadline_s: float = 600.0, io_threads: int = IO_THREADS,
                 extract_threads: int = 64):
        self.status, self.results, self.errors, self.extractor = status, results, errors, extractor
        self.node_id = node_id
        self.deadline_s = deadline_s
        self.io_pool = ThreadPoolExecutor(max_workers=io_threads, thread_name_prefix="dlamd-io")
        # upload digests run beside the receive loop; a pool of their own keeps a status write from queueing behind
        # a hundred megabytes of hashing
        self.hash_pool = ThreadPoolExecutor(max_workers=4, thread_name_prefix="dlamd-hash")
        # extraction calls block on the GPU workers; they get their own pool so I/O never queues behind them, and
        # enough threads that every concurrent upload has its pages in the batcher at the same time (with 8, a burst
        # of 32 four-page scans reached the engine as 8-page batches)
        self.extract_pool = ThreadPoolExecutor(max_workers=extract_threads, thread_name_prefix="dlamd-extract")
        self.tasks: Set[asyncio.Task] = set()

    # ------------------------------------------------------------------ helpers
    async def _io(self, fn, *args):
        return await asyncio.get_running_loop().run_in_executor(self.io_pool, fn, *args)

    async def claim(self, document_id: str, document_hash: str, filename: str) -> Optional[dict]:
        """Try to take ownership of ``document_id``.  Returns None when claimed, else the existing status dict
        (duplicate upload, ocr_service.py:441-449)."""
        data = {"status": "queued", "document_hash": document_hash, "filename": filename, "started_at": _now(),
                "owner": self.node_id}
        ok = await self._io(self.status.claim, document_id, data)
        if ok:
            return None
        existing = await self._io(self.status.get, document_id)
        if existing:
            return existing
        # the store is unreachable or the key vanished between the two calls: behave like the reference
        # (which only logs Redis errors) and proceed with a best-effort plain write
        await self._io(self.status.update, document_id, data)
        return None

    def start(self, pdf_bytes: bytes, document_id: str, document_hash: Optional[str], hash_future=None) -> asyncio.Task:
        """Fire-and-forget background job (ocr_service.py:460); the task is kept referenced until done.
        ``hash_future``: a ``concurrent.futures.Future`` of the digest when it is still being computed."""
        t = asyncio.create_task(self.process(pdf_bytes, document_id, document_hash, hash_future))
        self.tasks.add(t)
        t.add_done_callback(self.tasks.discard)
        metrics.JOBS_IN_FLIGHT.set(len(self.tasks))
        return t

    # ------------------------------------------------------------------ the job
    async def process(self, pdf_bytes: bytes, document_id: str, document_hash: Optional[str] = None,
                      hash_future=None) -> Dict[str, str]:
        t0 = time.perf_counter()
        fut = None
        ok = False
        try:
            loop = asyncio.get_running_loop()
            # extraction does not read the digest: it starts first, and a digest still on its way (the hash thread
            # runs a few pieces behind the upload) is awaited beside it, before the 'processing' record is written
            fut = loop.run_in_executor(self.extract_pool, self.extractor.extract, pdf_bytes, document_id, document_hash)
            if hash_future is not None:
                try:
                    document_hash = await asyncio.wrap_future(hash_future)
                except Exception:  # noqa: BLE001 - the streamed digest is an optimisation, never a reason to fail
                    raw = pdf_bytes.view() if hasattr(pdf_bytes, "view") else pdf_bytes
                    document_hash = await self._io(lambda: hashlib.sha256(raw).hexdigest())
                    del raw
            await self._io(self.status.update, document_id, {
                "status": "processing", "document_hash": document_hash, "started_at": _now(), "owner": self.node_id})
            try:
                result = await asyncio.wait_for(fut, timeout=self.deadline_s)
            except asyncio.TimeoutError:
                result = {"status": "failed", "error": f"document deadline of {self.deadline_s:.0f} s exceeded"}
            t1 = time.perf_counter()
            metrics.STAGE_SECONDS.labels("extract").observe(t1 - t0)
            if result["status"] == "completed":
                saved = await self._io(self.save_results, document_id, result)
                if not saved:
                    raise RuntimeError("result store rejected the document")
                final = {"status": "completed", "document_hash": document_hash, "result": result["result"],
                         "completed_at": _now(), "owner": self.node_id}
                if result.get("warnings"):
                    final["warnings"] = result["warnings"]      # pages with a damaged image stream / read with low confidence
                    for w in result["warnings"]:
                        metrics.PAGE_WARNINGS.labels("low_confidence" if "low confidence" in w else "damaged").inc()
                await self._io(self.status.update, document_id, final)
                metrics.STAGE_SECONDS.labels("store").observe(time.perf_counter() - t1)
                metrics.DOCUMENTS.labels("completed").inc()
                metrics.PAGES.inc(int(result.get("pages", 0) or 0))
                ok = True
                return {"status": "completed", "result": result["result"]}
            await self._fail(document_id, document_hash, result["error"])
            return {"status": "failed", "error": result["error"]}
        except Exception as e:  # noqa: BLE001 - own exceptions => failed (ocr_service.py:323-337)
            logger.error(f"Error in process_pdf_async for {document_id}: {str(e)}")
            await self._fail(document_id, document_hash, str(e))
            return {"status": "failed", "error": str(e)}
        finally:
            metrics.STAGE_SECONDS.labels("total").observe(time.perf_counter() - t0)
            metrics.JOBS_IN_FLIGHT.set(max(0, len(self.tasks) - 1))
            if hasattr(pdf_bytes, "release"):
                # an upload that was received into shared memory (api/app.py): this job owns the segment.  After a
                # deadline the extract thread may still be parsing it - the name goes now, the mapping when that
                # thread lets go of its views
                if fut is not None and not fut.done():
                    fut.add_done_callback(lambda _f, d=pdf_bytes: d.release())
                else:
                    pdf_bytes.release(reusable=ok)

    async def _fail(self, document_id: str, document_hash: Optional[str], error: str) -> None:
        await self._io(self.status.update, document_id, {
            "status": "failed", "document_hash": document_hash, "error": error, "completed_at": _now(),
            "owner": self.node_id})
        await self._io(self.errors.save, document_id, error)
        metrics.DOCUMENTS.labels("failed").inc()

    def save_results(self, document_id: str, result: dict) -> bool:
        """``save_results`` (ocr_service.py:339-385): upsert the six-field document, then the preview status."""
        try:
            if result["status"] != "completed":
                logger.info(f"Result of {document_id} is not stored: extraction did not complete")
                return False
            text = result["result"]
            self.results.upsert(make_result_document(document_id, text))
            preview = text[:PREVIEW_CHARS] + "..." if len(text) > PREVIEW_CHARS else text
            self.status.update(document_id, {"status": "completed", "result_preview": preview, "completed_at": _now()})
            logger.info(f"Stored the result document of {document_id}")
            return True
        except Exception as e:  # noqa: BLE001
            logger.error(f"Error saving results for {document_id} to MongoDB: {str(e)}")
            return False

    # ------------------------------------------------------------------ lifecycle
    async def drain(self, timeout: float = 30.0) -> None:
        if self.tasks:
            await asyncio.wait(list(self.tasks), timeout=timeout)

    def sweep_orphans(self) -> int:
        """On startup: documents this node left in queued/processing before a restart become failed."""
        return self.status.sweep_orphans(owner=self.node_id or None, reason="service restarted")

    def shutdown(self) -> None:
        self.io_pool.shutdown(wait=False)
        self.hash_pool.shutdown(wait=False)
        self.extract_pool.shutdown(wait=False)
