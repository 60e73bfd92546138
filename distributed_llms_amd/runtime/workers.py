"""GPU worker pool: data-parallel page sharding across the GPUs of one node.

The reference scales by running N stateless copies of the whole service behind a load
balancer, each extracting documents in a ``ProcessPoolExecutor(os.cpu_count())``
(services/ocr_service/ocr_service.py:42, 274-278; terraform/main.tf:176-294).  Here one
front end owns N worker *processes*, one per MI355X:

* a document's PDF bytes are placed in POSIX shared memory ONCE; workers map it and parse it
  with the native reader (no per-page pickling across the process boundary);
* its pages are split into contiguous shards handed to the least-loaded workers; each worker
  batches the pages it holds - across documents - with the native dynamic batcher and runs
  them through its ``OcrEngine``; per-page results come back on a queue and are gathered in page
  order by the front end (inference needs no collectives: GPUs stay independent);
* every worker heartbeats; the supervisor re-queues the shards of a dead worker onto the
  survivors and respawns it (the reference's pool would raise BrokenProcessPool forever, 5.3);
* ``FAULT_INJECT`` (settings.fault_inject) hooks make those paths testable:
  ``crash_after_tasks=N`` / ``slow_page_ms=M``.
"""
from __future__ import annotations

import logging
import multiprocessing as mp
import os
import queue
import threading
import time
import traceback
from concurrent.futures import Future
from dataclasses import dataclass, field
from multiprocessing import shared_memory
from typing import Dict, List, Optional, Tuple

import numpy as np

logger = logging.getLogger("distributed_llms_amd")


# ----------------------------------------------------------------------------------------- worker side
def _parse_faults(spec: str) -> Dict[str, float]:
    out: Dict[str, float] = {}
    for part in (spec or "").split(","):
        if "=" in part:
            k, v = part.split("=", 1)
            try:
                out[k.strip()] = float(v)
            except ValueError:
                pass
    return out


def build_engine(device: str, engine_spec: Dict) -> "object":
    """Construct the per-worker OCR engine.
    ``engine_spec``: {'model': 'full'|'tiny', 'recognizer': 'crnn'|'trocr', 'cfg': {...}, 'weights_dir': ...}."""
    import torch

    from ..models.crnn import CRNN
    from ..models.dbnet import DBNet
    from .engine import EngineConfig, OcrEngine

    cfg = EngineConfig(**(engine_spec.get("cfg") or {}))
    torch.manual_seed(int(engine_spec.get("seed", 1234)))
    if engine_spec.get("model", "full") == "tiny":
        det = DBNet(widths=(16, 32, 32, 64), stem=8, inner=32)
        rec = CRNN(hidden=64, widths=(16, 32, 32, 32, 64, 64))
    else:
        det, rec = DBNet(), CRNN()
    kind = engine_spec.get("recognizer", "crnn")
    if kind == "trocr":
        from ..models.trocr import TrOCR, TrOCRConfig

        tiny = engine_spec.get("model", "full") == "tiny"
        rec = TrOCR(TrOCRConfig(image_size=64, enc_dim=128, enc_layers=1, enc_heads=2, enc_mlp=256, dec_dim=128, dec_layers=1,
                                dec_heads=2, dec_ffn=256, vocab=272, max_len=16) if tiny else None)
    wd = engine_spec.get("weights_dir")
    if kind == "trocr" and wd and os.path.exists(os.path.join(str(wd), "trocr_config.json")) and engine_spec.get("model", "full") == "full":
        # a converted public checkpoint (tools/download_models.py) brings its own geometry and decoder variant
        import json

        from ..models.trocr import TrOCR, TrOCRConfig

        rec = TrOCR(TrOCRConfig(**json.load(open(os.path.join(str(wd), "trocr_config.json")))))
    if wd and str(wd).lower() != "none" and engine_spec.get("model", "full") == "full":
        from ..utils.weights import load_model_weights

        load_model_weights(det, os.path.join(wd, "dbnet.safetensors"), strict=False)
        rec_file = os.path.join(wd, "trocr.safetensors" if kind == "trocr" else "crnn.safetensors")
        if kind != "trocr" or os.path.exists(rec_file):
            load_model_weights(rec, rec_file, strict=False)
    formula = None
    fpath = os.path.join(wd, "formula.safetensors") if wd and str(wd).lower() != "none" else None
    if fpath and os.path.exists(fpath) and engine_spec.get("model", "full") == "full":
        from ..models.trocr import TrOCR, formula_config
        from ..utils.weights import load_model_weights

        formula = TrOCR(formula_config())
        load_model_weights(formula, fpath)
    return OcrEngine(device, det, rec, cfg, formula_rec=formula)


def worker_main(worker_id: int, device: str, engine_spec: Dict, task_q, result_q, fault_spec: str = "",
                max_wait_ms: float = 20.0, hbm_budget_gb: float = 200.0) -> None:
    """Entry point of one GPU worker process: a continuously running engine loop (runtime/service.py) fed with the
    shards the front end sends.  Shards of different documents share page batches; each shard is answered as soon
    as its last page is done."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        import numpy as np
        import torch

        from ..pdf import reader
        from .document import DocumentPipeline
        from .service import EngineService

        if device.startswith("cuda"):
            torch.cuda.set_device(torch.device(device))
        engine = build_engine(device, engine_spec)
        pipe = DocumentPipeline(engine)
        service = EngineService(pipe, max_wait_ms=max_wait_ms, hbm_budget_bytes=int(hbm_budget_gb * (1 << 30)))
        faults = _parse_faults(fault_spec)
        result_q.put(("ready", worker_id, os.getpid()))
    except Exception:
        result_q.put(("fatal", worker_id, traceback.format_exc()))
        return

    stop = threading.Event()
    state = {"tasks_done": 0, "tasks_seen": 0}
    mu = threading.Lock()

    def beat():
        # Liveness follows PROGRESS, not the mere existence of this thread: the beat carries the number of pages the
        # engine thread has finished.  A worker whose engine is stuck in a GPU call keeps beating (the call releases
        # the GIL) but its counter stands still, and the front end declares it hung.
        while not stop.wait(0.5):
            try:
                result_q.put(("hb", worker_id, time.time(), engine.pages_done + state["tasks_done"]))
            except Exception:
                return

    threading.Thread(target=beat, daemon=True, name="dlamd-heartbeat").start()

    if faults.get("slow_page_ms"):
        load_page = pipe._load_page

        def slow_load(*a, **k):
            time.sleep(faults["slow_page_ms"] / 1000.0)
            return load_page(*a, **k)

        pipe._load_page = slow_load

    def finish(job_id, shard_id, shm, doc, fut):
        err = None
        pages = []
        try:
            got = fut.result()
            for i in sorted(got):
                pr = got[i]
                if pr is not None:
                    pr.doc_id = job_id
                pages.append((i, pr))
        except BaseException as e:  # noqa: BLE001 - reported to the front end, the worker lives on
            err = f"{type(e).__name__}: {e}"
        finally:
            try:
                doc.close()
            finally:
                shm.close()
        if err is not None:
            result_q.put(("error", worker_id, job_id, shard_id, err))
        else:
            result_q.put(("done", worker_id, job_id, shard_id, pages, service.stats()))
        with mu:
            state["tasks_done"] += 1
            done = state["tasks_done"]
        if faults.get("crash_after_tasks") and done >= int(faults["crash_after_tasks"]) and worker_id == 0:
            os._exit(17)  # fault injection: die without cleanup, like a GPU fault would

    while True:
        task = task_q.get()
        if task is None:
            break
        job_id, shard_id, shm_name, nbytes, page_ids = task
        state["tasks_seen"] += 1
        if faults.get("hang_after_tasks") and state["tasks_seen"] > int(faults["hang_after_tasks"]) and worker_id == 0:
            # fault injection: the engine thread stops taking batches (what a wedged GPU call looks like from outside)
            service.fault_hook = lambda: time.sleep(3600.0)
        shm = None
        try:
            shm = shared_memory.SharedMemory(name=shm_name)
            # parse in place: the native reader works on the shared-memory mapping itself, nothing is copied
            doc = reader.PdfDocument(np.frombuffer(shm.buf, dtype=np.uint8, count=nbytes))
        except Exception as e:  # unreadable document: fail the shard, keep the worker alive
            if shm is not None:
                shm.close()
            result_q.put(("error", worker_id, job_id, shard_id, f"{type(e).__name__}: {e}"))
            continue
        fut = service.submit(doc, list(page_ids))
        fut.add_done_callback(lambda f, a=(job_id, shard_id, shm, doc): finish(*a, f))
    stop.set()
    service.close()
    pipe.close()


# ----------------------------------------------------------------------------------------- front-end side
@dataclass
class _Shard:
    shard_id: int
    page_ids: List[int]
    worker: int = -1
    done: bool = False
    attempts: int = 0


@dataclass
class _Job:
    job_id: str
    shm: shared_memory.SharedMemory
    nbytes: int
    shards: List[_Shard]
    future: Future
    deadline: float
    pages: Dict[int, object] = field(default_factory=dict)
    t0: float = field(default_factory=time.perf_counter)


class WorkerPool:
    """Front-end handle on the per-GPU worker processes."""

    def __init__(self, devices: List[str], engine_spec: Optional[Dict] = None, fault_inject: str = "",
                 max_wait_ms: float = 20.0, hbm_budget_gb: float = 200.0, min_pages_per_shard: int = 32,
                 doc_deadline_s: float = 600.0, heartbeat_timeout_s: float = 30.0, max_attempts: int = 3,
                 progress_timeout_s: float = 60.0, max_restarts: int = 5):
        self.devices = list(devices)
        self.engine_spec = engine_spec or {}
        self.fault_inject = fault_inject
        self.max_wait_ms = max_wait_ms
        self.hbm_budget_gb = hbm_budget_gb
        # a shard smaller than the engine's page batch would run the detector on half-empty batches
        self.min_pages_per_shard = max(1, min_pages_per_shard)
        self.doc_deadline_s = doc_deadline_s
        self.heartbeat_timeout_s = heartbeat_timeout_s
        self.progress_timeout_s = progress_timeout_s
        self.max_attempts = max_attempts
        self.max_restarts = max_restarts
        self.ctx = mp.get_context("spawn")     # HIP state must never be forked
        self.result_q = self.ctx.Queue()
        n = len(self.devices)
        self.task_qs: List = [None] * n
        self.procs: List = [None] * n
        self.last_beat: List[float] = [0.0] * n
        self.progress: List[int] = [-1] * n            # last progress counter a worker reported
        self.last_progress: List[float] = [0.0] * n    # when that counter last moved (or a shard came back)
        self.ready: List[bool] = [False] * n
        self.failures: List[int] = [0] * n             # consecutive deaths without a finished shard in between
        self.next_spawn: List[float] = [0.0] * n       # earliest time of the next respawn (exponential backoff)
        self.disabled: List[bool] = [False] * n        # gave up on this device after max_restarts
        self.inflight: List[Dict[Tuple[str, int], _Shard]] = [{} for _ in range(n)]
        self.pending: List[Tuple[str, int]] = []       # shards waiting for a worker to become ready
        self.jobs: Dict[str, _Job] = {}
        self.mu = threading.RLock()
        self._stop = threading.Event()
        self._threads: List[threading.Thread] = []
        self.restarts = 0
        self.pages_done = 0
        self.batch_stats: Dict[str, float] = {}
        self.worker_stats: Dict[int, Dict[str, object]] = {}     # last EngineService.stats() of every worker
        self._job_seq = 0

    # -- lifecycle
    def _spawn(self, wid: int, fault: str) -> None:
        q = self.ctx.Queue()
        p = self.ctx.Process(target=worker_main, name=f"dlamd-worker-{wid}", daemon=True,
                             args=(wid, self.devices[wid], self.engine_spec, q, self.result_q, fault,
                                   self.max_wait_ms, self.hbm_budget_gb))
        p.start()
        now = time.time()
        self.task_qs[wid], self.procs[wid] = q, p
        self.last_beat[wid], self.last_progress[wid], self.progress[wid], self.ready[wid] = now, now, -1, False

    def start(self, wait_ready_s: float = 600.0) -> "WorkerPool":
        for wid in range(len(self.devices)):
            self._spawn(wid, self.fault_inject)
        for fn, name in ((self._listen, "dlamd-results"), (self._supervise, "dlamd-supervisor")):
            t = threading.Thread(target=fn, daemon=True, name=name)
            t.start()
            self._threads.append(t)
        t_end = time.time() + wait_ready_s
        while time.time() < t_end:
            with self.mu:
                if all(self.ready):
                    return self
                if any(self.disabled):
                    break
            time.sleep(0.05)
        with self.mu:
            ok = all(self.ready)
        if not ok:
            self.stop()         # no supervisor may outlive a failed start: it would respawn dying workers forever
            raise RuntimeError("GPU workers failed to start (see log for the worker traceback)")
        return self

    def stop(self) -> None:
        self._stop.set()
        for q in self.task_qs:
            try:
                if q is not None:
                    q.put(None)
            except Exception:
                pass
        for p in self.procs:
            if p is None:
                continue
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        with self.mu:
            for job in list(self.jobs.values()):
                self._fail(job, "service shutting down")

    @property
    def num_workers(self) -> int:
        return len(self.devices)

    # -- submission
    def submit(self, pdf_bytes: bytes, n_pages: int, page_ids: Optional[List[int]] = None) -> Future:
        """Shard the pages of one PDF over the workers; resolves to {page_index: PageResult | None}."""
        fut: Future = Future()
        ids = list(range(n_pages)) if page_ids is None else list(page_ids)
        if not ids:
            fut.set_result({})
            return fut
        shm = shared_memory.SharedMemory(create=True, size=max(1, len(pdf_bytes)))
        if len(pdf_bytes):
            # numpy's copy loop runs without the interpreter lock (a memoryview slice assignment holds it: ~30 ms per
            # 120 MB document, in a front end whose event loop and other submitting threads need that lock)
            np.copyto(np.frombuffer(shm.buf, dtype=np.uint8, count=len(pdf_bytes)),
                      np.frombuffer(pdf_bytes, dtype=np.uint8))
        with self.mu:
            self._job_seq += 1
            job_id = f"j{os.getpid()}_{self._job_seq}"
            nw = max(1, sum(1 for w in range(self.num_workers) if not self.disabled[w]))
            n_shards = max(1, min(nw, (len(ids) + self.min_pages_per_shard - 1) // self.min_pages_per_shard))
            per = (len(ids) + n_shards - 1) // n_shards
            shards = [_Shard(k, ids[k * per : (k + 1) * per]) for k in range(n_shards) if ids[k * per : (k + 1) * per]]
            job = _Job(job_id, shm, len(pdf_bytes), shards, fut, time.time() + self.doc_deadline_s)
            self.jobs[job_id] = job
            for s in shards:
                self._dispatch(job, s)
        return fut

    def _load(self, wid: int) -> int:
        return sum(len(s.page_ids) for s in self.inflight[wid].values())

    def _usable(self, w: int) -> bool:
        return self.ready[w] and not self.disabled[w] and self.procs[w] is not None and self.procs[w].is_alive()

    def _dispatch(self, job: _Job, s: _Shard, avoid: int = -1) -> None:
        """Only workers that have reported ready (and not been declared hung since) take work; with none available
        the shard waits in ``pending`` until the supervisor sees one come up."""
        cand = [w for w in range(self.num_workers) if self._usable(w) and w != avoid] or \
               [w for w in range(self.num_workers) if self._usable(w)]
        if not cand:
            if all(self.disabled):
                self._fail(job, "no GPU worker is alive")
            else:
                self.pending.append((job.job_id, s.shard_id))
            return
        w = min(cand, key=self._load)
        s.worker, s.attempts = w, s.attempts + 1
        self.inflight[w][(job.job_id, s.shard_id)] = s
        self.task_qs[w].put((job.job_id, s.shard_id, job.shm.name, job.nbytes, s.page_ids))

    def _fail(self, job: _Job, msg: str) -> None:
        self.jobs.pop(job.job_id, None)
        for w in range(len(self.inflight)):
            for key in [k for k in self.inflight[w] if k[0] == job.job_id]:
                self.inflight[w].pop(key, None)
        self.pending = [k for k in self.pending if k[0] != job.job_id]
        self._release(job)
        if not job.future.done():
            job.future.set_exception(RuntimeError(msg))

    @staticmethod
    def _release(job: _Job) -> None:
        try:
            job.shm.close()
            job.shm.unlink()
        except Exception:
            pass

    # -- background threads
    def _listen(self) -> None:
        while not self._stop.is_set():
            try:
                msg = self.result_q.get(timeout=0.2)
            except queue.Empty:
                continue
            except (EOFError, OSError):
                return
            kind, wid = msg[0], msg[1]
            with self.mu:
                now = time.time()
                if kind == "hb":
                    self.last_beat[wid] = now
                    if len(msg) > 3 and msg[3] != self.progress[wid]:
                        self.progress[wid], self.last_progress[wid] = msg[3], now
                elif kind == "ready":
                    self.ready[wid] = True
                    self.last_beat[wid] = self.last_progress[wid] = now
                    logger.info(f"GPU worker {wid} ready on {self.devices[wid]} (pid {msg[2]})")
                elif kind == "fatal":
                    logger.error(f"GPU worker {wid} failed to start:\n{msg[2]}")
                elif kind in ("done", "error"):
                    job_id, shard_id = msg[2], msg[3]
                    self.last_beat[wid] = self.last_progress[wid] = now
                    self.failures[wid] = 0
                    s = self.inflight[wid].pop((job_id, shard_id), None)
                    job = self.jobs.get(job_id)
                    if job is None or s is None or s.done:
                        continue
                    if kind == "error":
                        self._fail(job, f"extraction failed: {msg[4]}")
                        continue
                    s.done = True
                    for i, pr in msg[4]:
                        job.pages[i] = pr
                    self.pages_done += len(msg[4])
                    self.worker_stats[wid] = dict(msg[5] or {})
                    for k, v in (msg[5] or {}).items():
                        if isinstance(v, (int, float)):
                            self.batch_stats[k] = v
                    if all(x.done for x in job.shards):
                        self.jobs.pop(job_id, None)
                        self._release(job)
                        if not job.future.done():
                            job.future.set_result(job.pages)

    def _supervise(self) -> None:
        while not self._stop.wait(0.25):
            now = time.time()
            with self.mu:
                for wid in range(self.num_workers):
                    if self.disabled[wid]:
                        continue
                    p = self.procs[wid]
                    if p is None:                      # waiting out the backoff of a dead worker
                        if now >= self.next_spawn[wid]:
                            self.restarts += 1
                            self._spawn(wid, "")       # a respawned worker runs without fault injection
                        continue
                    dead = not p.is_alive()
                    # hung: shards in flight but neither a heartbeat, nor any progress of its engine loop, for too long
                    silent = self.ready[wid] and bool(self.inflight[wid]) and now - self.last_beat[wid] > self.heartbeat_timeout_s
                    stuck = self.ready[wid] and bool(self.inflight[wid]) and now - self.last_progress[wid] > self.progress_timeout_s
                    if not (dead or silent or stuck):
                        continue
                    why = "died" if dead else ("stopped heartbeating" if silent else "made no progress")
                    logger.error(f"GPU worker {wid} {why} (exit code {p.exitcode}); re-queueing "
                                 f"{len(self.inflight[wid])} shard(s)")
                    if not dead:
                        p.terminate()
                        p.join(timeout=2)
                        if p.is_alive():
                            p.kill()
                    orphans = list(self.inflight[wid].items())
                    self.inflight[wid].clear()
                    self.ready[wid] = False
                    self.procs[wid] = None
                    self.failures[wid] += 1
                    if self.failures[wid] > self.max_restarts:
                        self.disabled[wid] = True
                        logger.error(f"GPU worker {wid} failed {self.failures[wid]} times in a row: giving up on {self.devices[wid]}")
                    else:
                        self.next_spawn[wid] = now + min(30.0, 0.5 * (2 ** (self.failures[wid] - 1)))
                    for (job_id, _), s in orphans:
                        job = self.jobs.get(job_id)
                        if job is None:
                            continue
                        if s.attempts >= self.max_attempts:
                            self._fail(job, f"shard {s.shard_id} failed on {s.attempts} workers")
                        else:
                            self._dispatch(job, s, avoid=wid)
                if self.pending and any(self._usable(w) for w in range(self.num_workers)):
                    waiting, self.pending = self.pending, []
                    for job_id, shard_id in waiting:
                        job = self.jobs.get(job_id)
                        if job is not None:
                            self._dispatch(job, next(x for x in job.shards if x.shard_id == shard_id))
                if self.pending and all(self.disabled):
                    for job_id, _ in list(self.pending):
                        job = self.jobs.get(job_id)
                        if job is not None:
                            self._fail(job, "no GPU worker is alive")
                for job in list(self.jobs.values()):
                    if now > job.deadline:
                        self._fail(job, f"document deadline of {self.doc_deadline_s:.0f} s exceeded")

    def describe(self) -> Dict[str, object]:
        with self.mu:
            return {"workers": self.num_workers, "alive": sum(p is not None and p.is_alive() for p in self.procs),
                    "ready": sum(self.ready), "disabled": sum(self.disabled),
                    "restarts": self.restarts, "jobs_in_flight": len(self.jobs), "pages_done": self.pages_done,
                    "queue_depth": [self._load(w) for w in range(self.num_workers)], "batcher": dict(self.batch_stats),
                    "worker_stats": {w: dict(v) for w, v in self.worker_stats.items()}}
