"""Engine stand-in with the host-side behaviour of the GPU engine and none of its GPU work.

Used to measure what the layers AROUND the engines cost - upload, shared-memory hand-off, native PDF parse, JPEG header
scans on the decode pool, dynamic batching, result packing, transport, assembly, stores - with N "GPUs" on a machine
that has none (bench/frontend_ceiling.py), and in CPU tests of the multi-process paths.  The stub
"GPU" is a clock: a batch of pages occupies it for ``len(batch) / pages_per_s`` seconds from the moment it is free,
and the engine thread sleeps until then - host work done meanwhile overlaps with it, as it does with a real device
that runs asynchronously.  Every page comes back with ``lines_per_page`` canned text lines of realistic length, so
result packing and assembly move what a real scan moves.

The reference's per-replica capacity is whatever ``ProcessPoolExecutor(os.cpu_count())`` gives one VM
(services/ocr_service/ocr_service.py:42); this is the tool that sizes the front end that replaces N such replicas.
"""
from __future__ import annotations

import time
from typing import Iterator, List

from .engine import EngineConfig, OcrEngine, PageResult, StageTimer, TextLine

_WORDS = ("the quick brown fox jumps over the lazy dog while seven wizards quietly box with jumping zebras and "
          "pack my red box with five dozen quality liquor jugs before noon").split()


class StubEngine:
    STREAM_END = OcrEngine.STREAM_END

    def __init__(self, pages_per_s: float = 1000.0, lines_per_page: int = 40, cfg: EngineConfig | None = None):
        self.cfg = cfg or EngineConfig()
        self.pages_per_s = float(pages_per_s)
        self.lines_per_page = int(lines_per_page)
        self.on_gpu = True            # page loaders hand over encoded JPEG streams, as they do for the GPU decoder
        self.device = None
        self.timer = StageTimer()
        self.pages_done = 0
        self.lines_done = 0
        self._lines = [" ".join(_WORDS[(k + j) % len(_WORDS)] for j in range(11)) for k in range(self.lines_per_page)]
        self._by_size = {}            # (H, W) -> the page's TextLine objects (read-only, shared by all pages of that size)
        self._free_at = 0.0           # when the stub device has finished everything handed to it so far

    def det_geometry(self, H: int, W: int):
        lim = self.cfg.det_limit_side
        s = min(1.0, lim / float(max(H, W)))
        oh, ow = max(32, int(round(H * s))), max(32, int(round(W * s)))
        return oh, ow, (oh + 31) // 32 * 32, (ow + 31) // 32 * 32

    def _results(self, batch) -> List[PageResult]:
        out = []
        for p in batch:
            H, W = p.hw
            pr = PageResult(p.doc_id, p.page_index, W, H, n_detected=self.lines_per_page)
            lines = self._by_size.get((H, W))
            if lines is None:
                step = 0.8 * H / max(1, self.lines_per_page)
                lines = self._by_size[(H, W)] = [
                    TextLine((0.1 * W, 0.1 * H + k * step, 0.9 * W, 0.1 * H + (k + 0.6) * step), text, 0.99, k // 8)
                    for k, text in enumerate(self._lines)]
            pr.lines = list(lines)
            out.append(pr)
        self.pages_done += len(out)
        self.lines_done += len(out) * self.lines_per_page
        return out

    def run_stream(self, next_batch) -> Iterator[List[PageResult]]:
        """Same contract as ``OcrEngine.run_stream``: up to ``pipeline_slots`` batches are in flight on the stub
        device; the source is only waited on when nothing is in flight, otherwise the oldest batch is finished."""
        import collections

        inflight = collections.deque()         # (finish time, results)
        depth = max(2, int(self.cfg.pipeline_slots))
        while True:
            batch = next_batch(not inflight)
            if batch is self.STREAM_END:
                break
            if batch:
                t0 = time.perf_counter()
                res = self._results(batch)
                del batch
                dur = len(res) / self.pages_per_s if self.pages_per_s > 0 else 0.0
                self._free_at = max(self._free_at, t0) + dur
                inflight.append((self._free_at, res))
                self.timer.add("stub_engine", dur)
                if len(inflight) < depth:
                    continue
            if inflight:
                fin, res = inflight.popleft()
                wait = fin - time.perf_counter()
                if wait > 0:
                    time.sleep(wait)
                yield res
        while inflight:
            fin, res = inflight.popleft()
            wait = fin - time.perf_counter()
            if wait > 0:
                time.sleep(wait)
            yield res

    def process_batches(self, batches) -> Iterator[List[PageResult]]:
        for b in batches:
            if b:
                if self.pages_per_s > 0:
                    time.sleep(len(b) / self.pages_per_s)
                yield self._results(b)

    def warmup(self, *a, **k) -> None:
        pass
