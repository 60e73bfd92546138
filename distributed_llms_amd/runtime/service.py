"""Continuously running engine loop: every document of a process streams through ONE software pipeline.

The reference extracts each upload in its own pool process, one document at a time per worker
(services/ocr_service/ocr_service.py:42, 274-278).  Here a single thread owns the GPU engine and never lets its
two-stream pipeline drain while work exists: pages of all submitted documents are decoded on the host pool, pushed
into the native dynamic batcher (grouped by detector input shape under the HBM budget) and popped by the engine
thread as page batches that freely mix documents.  Results are routed back per document.

* an idle engine waits for the batcher's deadline (``max_wait_ms`` after the first page of a bucket arrived) or a
  full batch; a busy engine only takes batches that are already due and otherwise advances its pipeline;
* decoded-but-not-yet-consumed pages are bounded (``max_pending_pages``) so a thousand-page upload cannot fill host
  memory ahead of the GPU;
* a failure while loading a page fails that document only; a failure inside the engine fails the documents that had
  pages in flight and the loop carries on;
* a document's future may resolve (one bad page, a cancelled deadline) while other pages of it are still being decoded
  on the pool or sit in the batcher / the engine with views into the caller's PDF bytes: every job therefore counts
  its outstanding pages, and ``future.drained`` is set - and the ``on_drained`` callbacks run - only when the last of
  them has been retired.  Callers close the document and unmap its memory then, never on the future alone.
"""
from __future__ import annotations

import collections
import logging
import threading
import time
from concurrent.futures import Future
from typing import Callable, Dict, List, Optional, Sequence

import torch

from ..pdf import reader
from .batcher import DynamicBatcher, page_cost_bytes, shape_key
from .document import DocumentPipeline
from .engine import OcrEngine, PageImage, PageResult

logger = logging.getLogger("distributed_llms_amd")


class _Job:
    __slots__ = ("key", "expected", "pages", "future", "mu", "failed", "on_progress", "t0", "refs", "drained", "_on_drained")

    def __init__(self, key: str, expected: int, on_progress: Optional[Callable[[int], None]]):
        self.key = key
        self.expected = expected
        self.pages: Dict[int, Optional[PageResult]] = {}
        self.future: Future = Future()
        self.mu = threading.Lock()
        self.failed = False
        self.on_progress = on_progress
        self.t0 = time.perf_counter()
        self.refs = expected               # pages not yet retired: loading, queued in the batcher or inside the engine
        self.drained = threading.Event()
        self._on_drained: List[Callable[[], None]] = []
        self.future.drained = self.drained      # what a caller waits for before it frees the document
        self.future.job_key = key
        if expected <= 0:
            self.drained.set()

    def retire(self, n: int = 1) -> bool:
        """``n`` pages of this job no longer touch the document; True when that was the last of them."""
        with self.mu:
            self.refs -= n
            if self.refs > 0 or self.drained.is_set():
                return False
            self.drained.set()
            cbs, self._on_drained = self._on_drained, []
        for cb in cbs:
            try:
                cb()
            except Exception:  # noqa: BLE001 - a failing cleanup must not take the calling (engine / decode) thread down
                logger.exception("drain callback failed")
        return True

    def add_drained_callback(self, cb: Callable[[], None]) -> None:
        with self.mu:
            if not self.drained.is_set():
                self._on_drained.append(cb)
                return
        cb()

    def done(self, index: int, result: Optional[PageResult]) -> None:
        with self.mu:
            if self.failed:
                return
            self.pages[index] = result
            finished = len(self.pages) >= self.expected
        if self.on_progress is not None:
            try:
                self.on_progress(1)
            except Exception:  # noqa: BLE001 - progress reporting must never break extraction
                pass
        if finished and not self.future.done():
            self.future.set_result(self.pages)

    def fail(self, exc: BaseException) -> None:
        with self.mu:
            if self.failed:
                return
            self.failed = True
        if not self.future.done():
            self.future.set_exception(exc)


class EngineService:
    def __init__(self, pipeline: DocumentPipeline, max_wait_ms: float = 3.0, hbm_budget_bytes: int = 0,
                 max_pending_pages: Optional[int] = None):
        self.pipeline = pipeline
        self.engine: OcrEngine = pipeline.engine
        self.max_wait_ms = max_wait_ms
        self.batcher = DynamicBatcher(self.engine.cfg.max_batch_pages, max_wait_ms, hbm_budget_bytes)
        self._jobs: Dict[str, _Job] = {}
        self._mu = threading.Lock()
        self._seq = 0
        self.recent_batch_sizes = collections.deque(maxlen=64)
        self._outstanding = 0            # pages submitted and not yet handed to the engine (queued or still decoding)
        self._in_engine: Dict[str, int] = {}     # job key -> pages handed to the engine and not yet returned
        self._slots = threading.Semaphore(max_pending_pages or 8 * self.engine.cfg.max_batch_pages)
        self._closed = False
        self.batches = 0
        self.pages_in_batches = 0
        self.last_progress = time.time()
        self.fault_hook: Optional[Callable[[], None]] = None      # tests: called by the engine thread before every batch
        self._thread = threading.Thread(target=self._run, daemon=True, name="dlamd-engine")
        self._thread.start()

    # ------------------------------------------------------------------ submission
    def submit(self, doc: reader.PdfDocument, page_ids: Sequence[int], priors=None,
               on_progress: Optional[Callable[[int], None]] = None) -> Future:
        """Queue pages of an open document; resolves to {page index: PageResult, or None for a page without an
        image}.  The caller keeps ``doc`` (and the bytes under it) alive until ``future.drained`` is set - the future
        itself can resolve earlier, with an error, while other pages of the document are still in flight."""
        with self._mu:
            if self._closed:
                raise RuntimeError("engine service is closed")
            self._seq += 1
            job = _Job(f"j{self._seq}", len(page_ids), on_progress)
            self._jobs[job.key] = job
            self._outstanding += len(page_ids)
        if not page_ids:
            job.future.set_result({})
            self._forget(job)
            return job.future
        for i in page_ids:
            self.pipeline.pool.submit(self._load_and_push, job, doc, i, None if priors is None else priors[i])
        return job.future

    def _forget(self, job: _Job) -> None:
        with self._mu:
            self._jobs.pop(job.key, None)
            self._in_engine.pop(job.key, None)

    def _retire(self, job: _Job, n: int = 1) -> None:
        if job.retire(n):
            self._forget(job)

    def on_drained(self, future: Future, cb: Callable[[], None]) -> None:
        """Run ``cb`` once no page of the job behind ``future`` is loading, queued or inside the engine (at once when
        that is already so).  It runs on whichever thread retires the last page: keep it short."""
        with self._mu:
            job = self._jobs.get(getattr(future, "job_key", ""))
        if job is None:
            cb()
        else:
            job.add_drained_callback(cb)

    def cancel(self, future: Future, reason: str = "cancelled") -> None:
        """Give up on a submitted job (deadline, client gone): its future fails, pages still waiting are dropped at
        the next batch pop and pages inside the engine finish unobserved."""
        with self._mu:
            job = self._jobs.get(getattr(future, "job_key", ""))
        if job is not None:
            job.fail(RuntimeError(reason))

    def _load_and_push(self, job: _Job, doc: reader.PdfDocument, i: int, prior) -> None:
        if job.failed:
            with self._mu:
                self._outstanding -= 1
            self._retire(job)
            return
        self._slots.acquire()
        pushed = False
        try:
            page = self.pipeline._load_page(doc, i, job.key, prior)
            if page is None:
                job.done(i, None)
                return
            H, W = page.hw
            _, _, Hd, Wd = self.engine.det_geometry(H, W)
            self.batcher.push(shape_key(Hd, Wd), page, page_cost_bytes(H, W, Hd, Wd))
            pushed = True
        except BaseException as e:  # noqa: BLE001 - one bad page fails its document, not the service
            job.fail(e if isinstance(e, Exception) else RuntimeError(str(e)))
        finally:
            if not pushed:
                self._slots.release()
                with self._mu:
                    self._outstanding -= 1
                self._retire(job)        # last: the native reader is no longer inside this page

    # ------------------------------------------------------------------ engine thread
    def _next_batch(self, block: bool):
        if self.fault_hook is not None:
            self.fault_hook()
        items = self._pop_balanced(block)
        if not items:
            if self._closed and self.batcher.pending() == 0:
                return OcrEngine.STREAM_END
            return None
        for _ in items:
            self._slots.release()
        live: List[PageImage] = []
        dropped: List[_Job] = []
        with self._mu:
            for p in items:
                job = self._jobs.get(p.doc_id)
                if job is not None and not job.failed:
                    live.append(p)
                    self._in_engine[job.key] = self._in_engine.get(job.key, 0) + 1
                elif job is not None:
                    dropped.append(job)
        n_popped = len(items)
        del items
        for job in dropped:
            self._retire(job)
        self.batches += 1
        self.pages_in_batches += len(live)
        self.recent_batch_sizes.append(n_popped)
        return live or None

    def _pop_balanced(self, block: bool):
        """Pop the next batch, sized so that the pages on their way (queued or still decoding) split into equal batches:
        100 pages of one document at a limit of 36 leave as 34 + 33 + 33 rather than 36 + 36 + 28 (a short last batch
        runs every layer at a worse tile fill).  A batch is taken as soon as that many pages wait; otherwise the
        batcher's own rule applies (full bucket, or the oldest page's deadline)."""
        B = self.batcher.max_items
        while True:
            with self._mu:
                out = self._outstanding
            cap = B
            if out > B:
                nb = -(-out // B)
                cap = -(-out // nb)
            if cap < B and self.batcher.pending() >= cap:
                _, items = self.batcher.pop(timeout_ms=0.0, flush=True, cap=cap)
            else:
                _, items = self.batcher.pop(timeout_ms=(2.0 if cap < B else -1.0) if block else 0.0, flush=False, cap=cap)
            if items:
                with self._mu:
                    self._outstanding -= len(items)
                return items
            if not block or cap >= B or (self._closed and self.batcher.pending() == 0):
                return items

    def _run(self) -> None:
        while True:
            try:
                for results in self.engine.run_stream(self._next_batch):
                    self.last_progress = time.time()
                    for pr in results:
                        with self._mu:
                            job = self._jobs.get(pr.doc_id)
                            if job is not None:
                                self._in_engine[job.key] = self._in_engine.get(job.key, 1) - 1
                        if job is not None:
                            job.done(pr.page_index, pr)
                            self._retire(job)
                return
            except BaseException as e:  # noqa: BLE001 - fail what was in flight, keep serving
                logger.error(f"engine loop error: {type(e).__name__}: {e}")
                with self._mu:
                    jobs = list(self._jobs.values())
                    lost, self._in_engine = self._in_engine, {}
                for job in jobs:
                    job.fail(RuntimeError(f"engine failure: {type(e).__name__}: {e}"))
                # the pipeline's batches went down with the generator: their pages hold no view any more
                for job in jobs:
                    if lost.get(job.key, 0) > 0:
                        self._retire(job, lost[job.key])
                if self._closed:
                    return

    # ------------------------------------------------------------------ lifecycle / stats
    def load(self) -> int:
        """Pages this service still owes: decoding, queued in the batcher or inside the engine."""
        with self._mu:
            return max(0, self._outstanding) + sum(v for v in self._in_engine.values() if v > 0)

    def stats(self) -> Dict[str, float]:
        s = dict(self.batcher.stats())
        s["engine_batches"] = self.batches
        s["mean_batch_pages"] = round(self.pages_in_batches / self.batches, 2) if self.batches else 0.0
        with self._mu:
            s["documents_in_flight"] = len(self._jobs)
        # cumulative seconds per pipeline stage (host decode wait, detector submit / wait, recogniser, finish) and the
        # HBM this process holds: exported per worker through GET /metrics
        s["stage_seconds"] = {**self.pipeline.timer.summary(), **self.engine.timer.summary()}
        if self.engine.on_gpu and getattr(self.engine, "device", None) is not None:
            s["hbm_allocated_bytes"] = int(torch.cuda.memory_allocated(self.engine.device))
            s["hbm_reserved_bytes"] = int(torch.cuda.memory_reserved(self.engine.device))
        return s

    def close(self, timeout: float = 10.0) -> None:
        with self._mu:
            self._closed = True
        self.batcher.close()
        self._thread.join(timeout=timeout)
