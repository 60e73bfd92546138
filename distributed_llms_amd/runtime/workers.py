"""GPU worker pool: data-parallel page sharding across the GPUs of one node.

The reference scales by running N stateless copies of the whole service behind a load
balancer, each extracting documents in a ``ProcessPoolExecutor(os.cpu_count())``
(services/ocr_service/ocr_service.py:42, 274-278; terraform/main.tf:176-294).  Here one
front end owns N worker *processes*, one per MI355X:

* a document's PDF bytes are placed in POSIX shared memory ONCE; workers map it and parse it
  with the native reader (no per-page pickling across the process boundary, cf. 2B-11);
* its pages are split into contiguous shards handed to the least-loaded workers; each worker
  batches the pages it holds - across documents - with the native dynamic batcher (N10) and runs
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
    if wd and str(wd).lower() != "none" and engine_spec.get("model", "full") == "full":
        from ..utils.weights import load_model_weights

        load_model_weights(det, os.path.join(wd, "dbnet.safetensors"), strict=False)
        rec_file = os.path.join(wd, "trocr.safetensors" if kind == "trocr" else "crnn.safetensors")
        if kind != "trocr" or os.path.exists(rec_file):
            load_model_weights(rec, rec_file, strict=False)
    return OcrEngine(device, det, rec, cfg)


def worker_main(worker_id: int, device: str, engine_spec: Dict, task_q, result_q, fault_spec: str = "",
                max_wait_ms: float = 20.0, hbm_budget_gb: float = 200.0) -> None:
    """Entry point of one GPU worker process."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        import numpy as np
        import torch

        from ..pdf import reader
        from .batcher import DynamicBatcher, page_cost_bytes, shape_key
        from .document import DocumentPipeline

        if device.startswith("cuda"):
            torch.cuda.set_device(torch.device(device))
        engine = build_engine(device, engine_spec)
        pipe = DocumentPipeline(engine)
        faults = _parse_faults(fault_spec)
        result_q.put(("ready", worker_id, os.getpid()))
    except Exception:
        result_q.put(("fatal", worker_id, traceback.format_exc()))
        return

    stop = threading.Event()

    def beat():
        while not stop.wait(0.5):
            try:
                result_q.put(("hb", worker_id, time.time()))
            except Exception:
                return

    threading.Thread(target=beat, daemon=True, name="dlamd-heartbeat").start()
    tasks_done = 0
    budget = int(hbm_budget_gb * (1 << 30))
    while True:
        task = task_q.get()
        if task is None:
            break
        round_tasks = [task]
        # gather whatever else is already waiting: pages of different documents share batches
        t_end = time.perf_counter() + max_wait_ms / 1000.0
        while len(round_tasks) < 64:
            try:
                t = task_q.get(timeout=max(0.0, t_end - time.perf_counter()))
            except queue.Empty:
                break
            if t is None:
                task_q.put(None)
                break
            round_tasks.append(t)
        shms, docs = [], {}
        results: Dict[Tuple[str, int], list] = {}
        errors: Dict[Tuple[str, int], str] = {}
        try:
            batcher = DynamicBatcher(engine.cfg.max_batch_pages, max_wait_ms, budget)
            total = 0
            futs = []
            for (job_id, shard_id, shm_name, nbytes, page_ids) in round_tasks:
                key = (job_id, shard_id)
                results[key] = []
                try:
                    shm = shared_memory.SharedMemory(name=shm_name)
                    shms.append(shm)
                    doc = reader.PdfDocument(bytes(shm.buf[:nbytes]))
                    docs[key] = doc
                    for i in page_ids:
                        futs.append((key, i, pipe.pool.submit(pipe._load_page, doc, i, f"{job_id}\x00{shard_id}", None)))
                        total += 1
                except Exception as e:  # unreadable document: fail the shard, keep the worker alive
                    errors[key] = f"{type(e).__name__}: {e}"

            def producer():
                for key, i, f in futs:
                    try:
                        p = f.result()
                        if faults.get("slow_page_ms"):
                            time.sleep(faults["slow_page_ms"] / 1000.0)
                        if p is None:
                            # no page image: report an empty page, the front end falls back to the text layer
                            results[key].append((i, None))
                            continue
                        H, W = p.hw
                        oh, ow, Hd, Wd = engine.det_geometry(H, W)
                        batcher.push(shape_key(Hd, Wd), p, page_cost_bytes(H, W, Hd, Wd))
                    except Exception as e:
                        errors[key] = f"page {i}: {type(e).__name__}: {e}"
                batcher.close()

            th = threading.Thread(target=producer, daemon=True, name="dlamd-producer")
            th.start()

            def batches():
                while True:
                    _, items = batcher.pop(timeout_ms=-1)
                    if not items:
                        return
                    yield items

            for batch_res in engine.process_batches(batches()):
                for pr in batch_res:
                    job_id, shard_id = pr.doc_id.split("\x00")
                    results[(job_id, int(shard_id))].append((pr.page_index, pr))
            th.join()
            stats = batcher.stats()
        except Exception:
            tb = traceback.format_exc()
            for key in results:
                errors.setdefault(key, tb)
            stats = {}
        finally:
            for d in docs.values():
                d.close()
            for s in shms:
                s.close()
        for (job_id, shard_id, *_rest) in round_tasks:
            key = (job_id, shard_id)
            if key in errors:
                result_q.put(("error", worker_id, job_id, shard_id, errors[key]))
            else:
                pages = []
                for i, pr in sorted(results[key], key=lambda t: t[0]):
                    if pr is None:
                        pages.append((i, None))
                    else:
                        pr.doc_id = job_id
                        pages.append((i, pr))
                result_q.put(("done", worker_id, job_id, shard_id, pages, stats))
            tasks_done += 1
            if faults.get("crash_after_tasks") and tasks_done >= int(faults["crash_after_tasks"]) and worker_id == 0:
                os._exit(17)  # fault injection: die without cleanup, like a GPU fault would
    stop.set()
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
                 max_wait_ms: float = 20.0, hbm_budget_gb: float = 200.0, min_pages_per_shard: int = 8,
                 doc_deadline_s: float = 600.0, heartbeat_timeout_s: float = 30.0, max_attempts: int = 3):
        self.devices = list(devices)
        self.engine_spec = engine_spec or {}
        self.fault_inject = fault_inject
        self.max_wait_ms = max_wait_ms
        self.hbm_budget_gb = hbm_budget_gb
        self.min_pages_per_shard = max(1, min_pages_per_shard)
        self.doc_deadline_s = doc_deadline_s
        self.heartbeat_timeout_s = heartbeat_timeout_s
        self.max_attempts = max_attempts
        self.ctx = mp.get_context("spawn")     # HIP state must never be forked
        self.result_q = self.ctx.Queue()
        self.task_qs: List = []
        self.procs: List = []
        self.last_beat: List[float] = []
        self.ready: List[bool] = []
        self.inflight: List[Dict[Tuple[str, int], _Shard]] = []
        self.jobs: Dict[str, _Job] = {}
        self.mu = threading.RLock()
        self._stop = threading.Event()
        self._threads: List[threading.Thread] = []
        self.restarts = 0
        self.pages_done = 0
        self.batch_stats: Dict[str, int] = {}
        self._job_seq = 0

    # -- lifecycle
    def _spawn(self, wid: int, fault: str) -> None:
        q = self.ctx.Queue()
        p = self.ctx.Process(target=worker_main, name=f"dlamd-worker-{wid}", daemon=True,
                             args=(wid, self.devices[wid], self.engine_spec, q, self.result_q, fault,
                                   self.max_wait_ms, self.hbm_budget_gb))
        p.start()
        if wid < len(self.procs):
            self.task_qs[wid], self.procs[wid] = q, p
            self.last_beat[wid], self.ready[wid] = time.time(), False
        else:
            self.task_qs.append(q)
            self.procs.append(p)
            self.last_beat.append(time.time())
            self.ready.append(False)
            self.inflight.append({})

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
            if any(not p.is_alive() for p in self.procs):
                break
            time.sleep(0.05)
        with self.mu:
            if not all(self.ready):
                raise RuntimeError("GPU workers failed to start (see log for the worker traceback)")
        return self

    def stop(self) -> None:
        self._stop.set()
        for q in self.task_qs:
            try:
                q.put(None)
            except Exception:
                pass
        for p in self.procs:
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
        shm.buf[: len(pdf_bytes)] = pdf_bytes
        with self.mu:
            self._job_seq += 1
            job_id = f"j{os.getpid()}_{self._job_seq}"
            nw = self.num_workers
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

    def _dispatch(self, job: _Job, s: _Shard, avoid: int = -1) -> None:
        alive = [w for w in range(self.num_workers) if self.procs[w].is_alive() and w != avoid] or \
                [w for w in range(self.num_workers) if self.procs[w].is_alive()]
        if not alive:
            self._fail(job, "no GPU worker is alive")
            return
        w = min(alive, key=self._load)
        s.worker, s.attempts = w, s.attempts + 1
        self.inflight[w][(job.job_id, s.shard_id)] = s
        self.task_qs[w].put((job.job_id, s.shard_id, job.shm.name, job.nbytes, s.page_ids))

    def _fail(self, job: _Job, msg: str) -> None:
        self.jobs.pop(job.job_id, None)
        for w in range(len(self.inflight)):
            for key in [k for k in self.inflight[w] if k[0] == job.job_id]:
                self.inflight[w].pop(key, None)
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
                if kind == "hb":
                    self.last_beat[wid] = time.time()
                elif kind == "ready":
                    self.ready[wid] = True
                    self.last_beat[wid] = time.time()
                    logger.info(f"GPU worker {wid} ready on {self.devices[wid]} (pid {msg[2]})")
                elif kind == "fatal":
                    logger.error(f"GPU worker {wid} failed to start:\n{msg[2]}")
                elif kind in ("done", "error"):
                    job_id, shard_id = msg[2], msg[3]
                    self.last_beat[wid] = time.time()
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
                    for k, v in (msg[5] or {}).items():
                        self.batch_stats[k] = self.batch_stats.get(k, 0) + int(v)
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
                    p = self.procs[wid]
                    dead = not p.is_alive()
                    hung = self.ready[wid] and self.inflight[wid] and now - self.last_beat[wid] > self.heartbeat_timeout_s
                    if not (dead or hung):
                        continue
                    logger.error(f"GPU worker {wid} {'died' if dead else 'stopped heartbeating'} "
                                 f"(exit code {p.exitcode}); re-queueing {len(self.inflight[wid])} shard(s)")
                    if hung:
                        p.terminate()
                    orphans = list(self.inflight[wid].values())
                    keys = list(self.inflight[wid].keys())
                    self.inflight[wid].clear()
                    self.restarts += 1
                    self._spawn(wid, "")   # a respawned worker runs without fault injection
                    for (job_id, _), s in zip(keys, orphans):
                        job = self.jobs.get(job_id)
                        if job is None:
                            continue
                        if s.attempts >= self.max_attempts:
                            self._fail(job, f"shard {s.shard_id} failed on {s.attempts} workers")
                        else:
                            self._dispatch(job, s, avoid=wid)
                for job in list(self.jobs.values()):
                    if now > job.deadline:
                        self._fail(job, f"document deadline of {self.doc_deadline_s:.0f} s exceeded")

    def describe(self) -> Dict[str, object]:
        with self.mu:
            return {"workers": self.num_workers, "alive": sum(p.is_alive() for p in self.procs),
                    "restarts": self.restarts, "jobs_in_flight": len(self.jobs), "pages_done": self.pages_done,
                    "queue_depth": [self._load(w) for w in range(self.num_workers)], "batcher": dict(self.batch_stats)}
