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
    if engine_spec.get("model") == "stub":
        # no model, no device: sleeps pages / pages_per_s per batch and answers canned lines (runtime/stub.py)
        from .stub import StubEngine

        return StubEngine(float(engine_spec.get("pages_per_s", 1000.0)), int(engine_spec.get("lines_per_page", 40)), cfg)
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

        det_file = "dbnet.safetensors"
        if cfg.inline_formulas and os.path.exists(os.path.join(wd, "dbnet_inline.safetensors")):
            # the detector whose formula channel also marks formulas inside text lines (train/page_trainer.py
            # --inline-share); the default one marks display formulas only
            det_file = "dbnet_inline.safetensors"
        load_model_weights(det, os.path.join(wd, det_file), strict=False)
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


class WorkerCore:
    """The engine side of one worker process, independent of how tasks arrive (``worker_main``: queues from a parent
    process; ``runtime/worker_server.py``: Unix sockets from several front ends).

    Tasks: ``("shard", job_id, shard_id, segment, offset, nbytes, page_ids, doc_name)`` and ``("cancel", job_id)``.
    Answers go through the ``reply`` callable of the task's sender: ``("done", wid, job_id, shard_id, packed bytes,
    stats)`` (runtime/wire.py) or ``("error", wid, job_id, shard_id, message)``."""

    def __init__(self, worker_id: int, device: str, engine_spec: Dict, fault_spec: str = "", max_wait_ms: float = 20.0,
                 hbm_budget_gb: float = 200.0, cpus: Optional[List[int]] = None):
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if cpus:
            # before any other thread exists: torch's pools, the decode pool and the engine thread inherit the mask
            from ..utils.affinity import pin_process

            pin_process(cpus)
        import torch

        from .document import DocumentPipeline
        from .service import EngineService

        self.worker_id = worker_id
        if device.startswith("cuda"):
            torch.cuda.set_device(torch.device(device))
        self.engine = build_engine(device, engine_spec)
        self.pipe = DocumentPipeline(self.engine)
        self.service = EngineService(self.pipe, max_wait_ms=max_wait_ms, hbm_budget_bytes=int(hbm_budget_gb * (1 << 30)))
        self.faults = _parse_faults(fault_spec)
        self.state = {"tasks_done": 0, "tasks_seen": 0}
        self.mu = threading.Lock()
        self.graveyard: List = []      # segments whose unmapping was refused (a view still alive somewhere); retried later
        self.live: Dict[Tuple[str, int], object] = {}      # (job, shard) -> future, for cancellation
        if self.faults.get("slow_page_ms"):
            load_page = self.pipe._load_page
            delay = self.faults["slow_page_ms"] / 1000.0

            def slow_load(*a, **k):
                time.sleep(delay)
                return load_page(*a, **k)

            self.pipe._load_page = slow_load

    def heartbeat(self) -> tuple:
        """Liveness follows PROGRESS, not the mere existence of a thread: the beat carries the number of pages the
        engine thread has finished.  A worker whose engine is stuck in a GPU call keeps beating (the call releases
        the GIL) but its counter stands still, and the front end declares it hung.  The last field is the number of
        pages the worker still owes to ALL its front ends: what they balance their dispatch on."""
        return ("hb", self.worker_id, time.time(), self.engine.pages_done + self.state["tasks_done"], self.service.load())

    def _release(self, seg, doc) -> None:
        """Runs once no page of the shard is loading, queued or inside the engine (``EngineService.on_drained``):
        only then may the native document and the mapping under it go away."""
        try:
            doc.close()
        except Exception:  # noqa: BLE001
            logger.exception("closing a document failed")
        with self.mu:
            self.graveyard.append(seg)
            pending, self.graveyard = self.graveyard, []
        for m in pending:
            try:
                m.close()
            except BufferError:            # exported pointers exist: some object still views the segment
                with self.mu:
                    self.graveyard.append(m)
            except Exception:  # noqa: BLE001
                pass

    def _finish(self, job_id: str, shard_id: int, reply, fut) -> None:
        """Done-callback of a shard's future: the answer goes out FIRST, before any cleanup can fail."""
        from . import wire

        try:
            got = fut.result()
            pages = []
            for i in sorted(got):
                pr = got[i]
                if pr is not None:
                    pr.doc_id = job_id
                pages.append((i, pr))
            blob = wire.pack_shard(pages)
            reply(("done", self.worker_id, job_id, shard_id, blob, self.service.stats(), self.service.load()))
        except BaseException as e:  # noqa: BLE001 - reported to the front end, the worker lives on
            try:
                reply(("error", self.worker_id, job_id, shard_id, f"{type(e).__name__}: {e}"))
            except Exception:  # noqa: BLE001 - the front end went away
                pass
        with self.mu:
            self.state["tasks_done"] += 1
            done = self.state["tasks_done"]
            self.live.pop((job_id, shard_id), None)
        if self.faults.get("crash_after_tasks") and done >= int(self.faults["crash_after_tasks"]) and self.worker_id == 0:
            os._exit(17)  # fault injection: die without cleanup, like a GPU fault would

    def handle(self, task: tuple, reply) -> None:
        from ..pdf import reader
        from .shm import Segment

        if task[0] == "cancel":
            # the front end gave up on this document (deadline, failure of another shard): stop spending GPU time on it
            with self.mu:
                futs = [f for (j, _), f in self.live.items() if j == task[1]]
            for f in futs:
                self.service.cancel(f, "cancelled by the front end")
            return
        _, job_id, shard_id, seg_name, offset, nbytes, page_ids, _doc_name = task
        self.state["tasks_seen"] += 1
        if (self.faults.get("hang_after_tasks") and self.state["tasks_seen"] > int(self.faults["hang_after_tasks"])
                and self.worker_id == 0):
            # fault injection: the engine thread stops taking batches (what a wedged GPU call looks like from outside)
            self.service.fault_hook = lambda: time.sleep(3600.0)
        seg = None
        try:
            seg = Segment.attach(seg_name)
            # parse in place: the native reader works on the shared-memory mapping itself, nothing is copied
            doc = reader.PdfDocument(np.frombuffer(seg.buf, dtype=np.uint8, count=nbytes, offset=offset))
        except Exception as e:  # unreadable document: fail the shard, keep the worker alive
            if seg is not None:
                try:
                    seg.close()
                except BufferError:
                    pass
            reply(("error", self.worker_id, job_id, shard_id, f"{type(e).__name__}: {e}"))
            return
        fut = self.service.submit(doc, list(page_ids))
        with self.mu:
            self.live[(job_id, shard_id)] = fut
        fut.add_done_callback(lambda f, a=(job_id, shard_id, reply): self._finish(*a, f))
        self.service.on_drained(fut, lambda a=(seg, doc): self._release(*a))

    def close(self) -> None:
        self.service.close()
        self.pipe.close()


def worker_main(worker_id: int, device: str, engine_spec: Dict, task_q, result_q, fault_spec: str = "",
                max_wait_ms: float = 20.0, hbm_budget_gb: float = 200.0, cpus: Optional[List[int]] = None) -> None:
    """Entry point of one GPU worker process owned by a ``WorkerPool``: a continuously running engine loop
    (runtime/service.py) fed with the shards the front end sends.  Shards of different documents share page batches;
    each shard is answered as soon as its last page is done."""
    try:
        core = WorkerCore(worker_id, device, engine_spec, fault_spec, max_wait_ms, hbm_budget_gb, cpus)
        result_q.put(("ready", worker_id, os.getpid()))
    except Exception:
        result_q.put(("fatal", worker_id, traceback.format_exc()))
        return

    stop = threading.Event()

    def beat():
        while not stop.wait(0.2):
            try:
                result_q.put(core.heartbeat())
            except Exception:
                return

    threading.Thread(target=beat, daemon=True, name="dlamd-heartbeat").start()
    while True:
        task = task_q.get()
        if task is None:
            break
        core.handle(task, result_q.put)
    stop.set()
    core.close()


# ----------------------------------------------------------------------------------------- front-end side
@dataclass
class ShmDocument:
    """A PDF that already lives in a shared-memory segment (``api/app.py`` receives uploads straight into one):
    ``nbytes`` bytes at ``offset``.  ``release()`` removes the segment; whoever created it calls that when the
    document is finished."""

    seg: object
    offset: int
    nbytes: int
    pool: object = None           # runtime.shm.SegmentPool the segment goes back to after a clean finish

    def array(self) -> np.ndarray:
        return np.frombuffer(self.seg.buf, dtype=np.uint8, count=self.nbytes, offset=self.offset)

    def view(self) -> memoryview:
        return memoryview(self.seg.buf)[self.offset : self.offset + self.nbytes]

    def __len__(self) -> int:
        return self.nbytes

    def release(self, reusable: bool = False) -> None:
        """``reusable``: the document finished cleanly and no worker reads the segment any more."""
        if self.pool is not None:
            self.pool.give_back(self.seg, reusable)
        else:
            self.seg.release()


class PoolResult:
    """What a ``WorkerPool.submit`` future resolves to: the packed shards of the document, readable as a mapping
    {page index: PageResult or None} that materialises pages on access."""

    def __init__(self, shards: List, missing: Optional[List[int]] = None):
        self.shards = shards
        self._where: Dict[int, Tuple[object, int]] = {}
        for sh in shards:
            for k, i in enumerate(sh.page_indices()):
                self._where[i] = (sh, k)
        for i in missing or []:
            self._where.setdefault(i, (None, -1))

    @property
    def n_lines(self) -> int:
        return sum(sh.n_lines for sh in self.shards)

    def __iter__(self):
        return iter(sorted(self._where))

    def __len__(self) -> int:
        return len(self._where)

    def __contains__(self, i) -> bool:
        return i in self._where

    def keys(self):
        return sorted(self._where)

    def get(self, i: int, default=None):
        hit = self._where.get(i)
        if hit is None or hit[0] is None:
            return default
        pr = hit[0].page_result(hit[1])
        return default if pr is None else pr

    def __getitem__(self, i: int):
        if i not in self._where:
            raise KeyError(i)
        return self.get(i)


@dataclass
class _Shard:
    shard_id: int
    page_ids: List[int]
    worker: int = -1
    done: bool = False
    attempts: int = 0
    result: object = None


@dataclass
class _Job:
    job_id: str
    seg: object                   # runtime.shm.Segment holding the PDF
    owns_seg: bool                # created here (copied from bytes) -> released here; else the caller's
    offset: int
    nbytes: int
    doc_name: str
    shards: List[_Shard]
    future: Future
    deadline: float
    t0: float = field(default_factory=time.perf_counter)


class _LocalLink:
    """A worker that is a child process of this pool, reached over ``multiprocessing`` queues."""

    def __init__(self, pool: "WorkerPool", wid: int):
        self.pool, self.wid = pool, wid
        self.q = None
        self.proc = None

    def start(self, fault: str) -> bool:
        pool = self.pool
        self.q = pool.ctx.Queue()
        self.proc = pool.ctx.Process(target=worker_main, name=f"dlamd-worker-{self.wid}", daemon=True,
                                     args=(self.wid, pool.devices[self.wid], pool.engine_spec, self.q, pool.result_q, fault,
                                           pool.max_wait_ms, pool.hbm_budget_gb, pool.worker_cpus(self.wid)))
        self.proc.start()
        return True

    def send(self, task) -> None:
        self.q.put(task)

    def alive(self) -> bool:
        return self.proc is not None and self.proc.is_alive()

    def exitcode(self):
        return None if self.proc is None else self.proc.exitcode

    def kill(self) -> None:
        p = self.proc
        if p is None:
            return
        if p.is_alive():
            p.terminate()
            p.join(timeout=2)
            if p.is_alive():
                p.kill()

    def stop(self) -> None:
        try:
            if self.q is not None:
                self.q.put(None)
        except Exception:  # noqa: BLE001
            pass

    def join(self, timeout: float) -> None:
        if self.proc is not None:
            self.proc.join(timeout=timeout)
            if self.proc.is_alive():
                self.proc.terminate()


class _SocketLink:
    """A worker server shared by the front ends of a node (runtime/worker_server.py), reached over its Unix socket.
    This side neither starts nor kills the process - the node supervisor (api/serve.py) and the server's own watchdog
    do; "restart" here means: connect again once the socket is back."""

    def __init__(self, pool: "WorkerPool", wid: int, path: str):
        self.pool, self.wid, self.path = pool, wid, path
        self.conn = None
        self.wlock = threading.Lock()
        self._alive = False

    def start(self, fault: str) -> bool:
        from multiprocessing.connection import Client

        try:
            conn = Client(self.path, family="AF_UNIX")
        except OSError:
            return False
        self.conn, self._alive = conn, True
        threading.Thread(target=self._read, args=(conn,), daemon=True, name=f"dlamd-link-{self.wid}").start()
        return True

    def _read(self, conn) -> None:
        try:
            while True:
                msg = conn.recv()
                # the server numbers itself; this pool knows it by its position in ``devices``
                self.pool.result_q.put((msg[0], self.wid) + tuple(msg[2:]))
        except Exception:  # noqa: BLE001 - EOF, a reset, a message cut off by the server's exit: the link is down
            pass
        finally:
            if self.conn is conn:
                self._alive = False

    def send(self, task) -> None:
        with self.wlock:
            try:
                self.conn.send(task)
            except (OSError, ValueError, AttributeError):
                self._alive = False

    def alive(self) -> bool:
        return self._alive

    def exitcode(self):
        return None

    def kill(self) -> None:
        self._alive = False
        try:
            if self.conn is not None:
                self.conn.close()
        except OSError:
            pass

    def stop(self) -> None:
        self.kill()

    def join(self, timeout: float) -> None:
        pass


class WorkerPool:
    """Front-end handle on the per-GPU workers.

    ``devices``: one entry per worker - a device string (``"cuda:3"``, ``"cpu"``) for a worker process this pool
    starts and owns, or ``"unix:/path/to/socket"`` for a worker server that several front ends share.  With shared
    workers every front end sees every GPU of the node: a document goes to the least-loaded workers whoever received
    it, ``home`` only breaks ties (so an idle node keeps a front end's documents on the GPU next to it)."""

    def __init__(self, devices: List[str], engine_spec: Optional[Dict] = None, fault_inject: str = "",
                 max_wait_ms: float = 20.0, hbm_budget_gb: float = 200.0, min_pages_per_shard: int = 32,
                 doc_deadline_s: float = 600.0, heartbeat_timeout_s: float = 30.0, max_attempts: int = 3,
                 progress_timeout_s: float = 60.0, max_restarts: int = 5, pin_cpus: bool = True, home: int = 0):
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
        self.pin_cpus = pin_cpus
        self.home = home
        self.remote = any(d.startswith("unix:") for d in self.devices)
        self.ctx = mp.get_context("spawn")     # HIP state must never be forked
        self.result_q = queue.Queue() if self.remote else self.ctx.Queue()
        n = len(self.devices)
        self.links: List = [_SocketLink(self, w, d[5:]) if d.startswith("unix:") else _LocalLink(self, w)
                            for w, d in enumerate(self.devices)]
        self.up: List[bool] = [False] * n              # a process / connection exists for this worker
        self.last_beat: List[float] = [0.0] * n
        self.progress: List[int] = [-1] * n            # last progress counter a worker reported
        self.last_progress: List[float] = [0.0] * n    # when that counter last moved (or a shard came back)
        self.reported_load: List[int] = [0] * n        # pages the worker owes to all its front ends (its last report)
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

    def worker_cpus(self, wid: int) -> Optional[List[int]]:
        """Cores worker ``wid`` pins itself to (utils/affinity.py); None: leave it to the scheduler (CPU test workers,
        or fewer cores than workers)."""
        if not self.pin_cpus or not self.devices[wid].startswith("cuda"):
            return None
        from ..utils.affinity import allowed_cpus, cpus_for_worker

        if len(allowed_cpus()) < 2 * len(self.devices):
            return None
        dev = self.devices[wid]
        idx = int(dev.split(":")[1]) if ":" in dev else wid
        return cpus_for_worker(idx, max(len(self.devices), idx + 1))

    # -- lifecycle
    def _spawn(self, wid: int, fault: str) -> bool:
        ok = self.links[wid].start(fault)
        now = time.time()
        self.up[wid] = ok
        self.last_beat[wid], self.last_progress[wid], self.progress[wid], self.ready[wid] = now, now, -1, False
        self.reported_load[wid] = 0
        return ok

    def start(self, wait_ready_s: float = 600.0) -> "WorkerPool":
        for wid in range(len(self.devices)):
            if not self._spawn(wid, self.fault_inject):
                self.next_spawn[wid] = time.time() + 0.25          # a worker server that is still loading its models
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
        for ln in self.links:
            ln.stop()
        for ln in self.links:
            ln.join(timeout=5)
        with self.mu:
            for job in list(self.jobs.values()):
                self._fail(job, "service shutting down")

    @property
    def num_workers(self) -> int:
        return len(self.devices)

    # -- submission
    def submit(self, pdf, n_pages: int, page_ids: Optional[List[int]] = None, doc_name: str = "") -> Future:
        """Shard the pages of one PDF over the workers; resolves to a ``PoolResult``.  ``pdf``: the file's bytes (copied
        into a shared-memory segment once) or a ``ShmDocument`` (already in one: nothing is copied, and the segment
        stays the caller's to release after the future is done)."""
        from .shm import Segment

        fut: Future = Future()
        ids = list(range(n_pages)) if page_ids is None else list(page_ids)
        if not ids:
            fut.set_result(PoolResult([]))
            return fut
        if isinstance(pdf, ShmDocument):
            seg, owns, offset, nbytes = pdf.seg, False, pdf.offset, pdf.nbytes
        else:
            nbytes, offset, owns = len(pdf), 0, True
            seg = Segment.create(nbytes)
            if nbytes:
                # numpy's copy loop runs without the interpreter lock (a memoryview slice assignment holds it: ~30 ms
                # per 120 MB document, in a front end whose event loop and other submitting threads need that lock)
                np.copyto(np.frombuffer(seg.buf, dtype=np.uint8, count=nbytes), np.frombuffer(pdf, dtype=np.uint8))
        with self.mu:
            self._job_seq += 1
            job_id = f"j{os.getpid()}_{self._job_seq}"
            nw = max(1, sum(1 for w in range(self.num_workers) if not self.disabled[w]))
            n_shards = max(1, min(nw, (len(ids) + self.min_pages_per_shard - 1) // self.min_pages_per_shard))
            per = (len(ids) + n_shards - 1) // n_shards
            shards = [_Shard(k, ids[k * per : (k + 1) * per]) for k in range(n_shards) if ids[k * per : (k + 1) * per]]
            job = _Job(job_id, seg, owns, offset, nbytes, doc_name, shards, fut, time.time() + self.doc_deadline_s)
            self.jobs[job_id] = job
            for s in shards:
                self._dispatch(job, s)
        return fut

    def _load(self, wid: int) -> int:
        """Pages worker ``wid`` still owes: what this front end has in flight there, or the worker's own count when
        other front ends keep it busier than that."""
        return max(sum(len(s.page_ids) for s in self.inflight[wid].values()), self.reported_load[wid] if self.remote else 0)

    def _usable(self, w: int) -> bool:
        return self.ready[w] and not self.disabled[w] and self.up[w] and self.links[w].alive()

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
        n = self.num_workers
        w = min(cand, key=lambda c: (self._load(c), (c - self.home) % n))
        s.worker, s.attempts = w, s.attempts + 1
        self.inflight[w][(job.job_id, s.shard_id)] = s
        self.reported_load[w] += len(s.page_ids) if self.remote else 0      # until the worker's next report says so itself
        self.links[w].send(("shard", job.job_id, s.shard_id, job.seg.name, job.offset, job.nbytes, s.page_ids, job.doc_name))

    def _fail(self, job: _Job, msg: str) -> None:
        self.jobs.pop(job.job_id, None)
        for w in range(len(self.inflight)):
            keys = [k for k in self.inflight[w] if k[0] == job.job_id]
            for key in keys:
                self.inflight[w].pop(key, None)
            if keys and self._usable(w):
                try:        # shards of this document still running there: no more GPU time for a document nobody waits for
                    self.links[w].send(("cancel", job.job_id))
                except Exception:  # noqa: BLE001
                    pass
        self.pending = [k for k in self.pending if k[0] != job.job_id]
        self._release(job)
        if not job.future.done():
            job.future.set_exception(RuntimeError(msg))

    @staticmethod
    def _release(job: _Job) -> None:
        if job.owns_seg:
            job.seg.release()

    # -- background threads
    def _listen(self) -> None:
        from .wire import PackedShard

        while not self._stop.is_set():
            try:
                msg = self.result_q.get(timeout=0.2)
            except queue.Empty:
                continue
            except (EOFError, OSError):
                return
            kind, wid = msg[0], msg[1]
            shard_result = None
            if kind == "done":
                try:
                    shard_result = PackedShard(msg[4])        # header parse only; outside the pool lock
                except Exception as e:  # noqa: BLE001
                    kind, msg = "error", ("error", wid, msg[2], msg[3], f"undecodable shard result: {e}")
            with self.mu:
                now = time.time()
                if kind == "hb":
                    self.last_beat[wid] = now
                    if len(msg) > 3 and msg[3] != self.progress[wid]:
                        self.progress[wid], self.last_progress[wid] = msg[3], now
                    if len(msg) > 4:
                        self.reported_load[wid] = int(msg[4])
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
                    s.done, s.result = True, shard_result
                    self.pages_done += shard_result.n_pages
                    self.worker_stats[wid] = dict(msg[5] or {})
                    if len(msg) > 6:
                        self.reported_load[wid] = int(msg[6])
                    for k, v in (msg[5] or {}).items():
                        if isinstance(v, (int, float)):
                            self.batch_stats[k] = v
                    if all(x.done for x in job.shards):
                        self.jobs.pop(job_id, None)
                        self._release(job)
                        if not job.future.done():
                            job.future.set_result(PoolResult([x.result for x in job.shards]))

    def _supervise(self) -> None:
        while not self._stop.wait(0.25):
            now = time.time()
            with self.mu:
                for wid in range(self.num_workers):
                    if self.disabled[wid]:
                        continue
                    ln = self.links[wid]
                    if not self.up[wid]:                   # waiting out the backoff of a dead worker
                        if now >= self.next_spawn[wid]:
                            if self._spawn(wid, ""):       # a respawned worker runs without fault injection
                                self.restarts += 1 if self.failures[wid] else 0
                            else:
                                self.next_spawn[wid] = now + 0.5
                        continue
                    dead = not ln.alive()
                    # hung: shards in flight but neither a heartbeat, nor any progress of its engine loop, for too long
                    silent = self.ready[wid] and bool(self.inflight[wid]) and now - self.last_beat[wid] > self.heartbeat_timeout_s
                    stuck = self.ready[wid] and bool(self.inflight[wid]) and now - self.last_progress[wid] > self.progress_timeout_s
                    if not (dead or silent or stuck):
                        continue
                    why = "died" if dead else ("stopped heartbeating" if silent else "made no progress")
                    logger.error(f"GPU worker {wid} {why} (exit code {ln.exitcode()}); re-queueing "
                                 f"{len(self.inflight[wid])} shard(s)")
                    if not dead:
                        ln.kill()
                    orphans = list(self.inflight[wid].items())
                    self.inflight[wid].clear()
                    self.ready[wid] = False
                    self.up[wid] = False
                    self.failures[wid] += 1
                    if self.failures[wid] > self.max_restarts and not self.remote:
                        self.disabled[wid] = True
                        logger.error(f"GPU worker {wid} failed {self.failures[wid]} times in a row: giving up on {self.devices[wid]}")
                    else:
                        self.next_spawn[wid] = now + min(30.0, 0.5 * (2 ** (min(self.failures[wid], 8) - 1)))
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
            return {"workers": self.num_workers, "alive": sum(self.up[w] and self.links[w].alive() for w in range(self.num_workers)),
                    "ready": sum(self.ready), "disabled": sum(self.disabled),
                    "restarts": self.restarts, "jobs_in_flight": len(self.jobs), "pages_done": self.pages_done,
                    "queue_depth": [self._load(w) for w in range(self.num_workers)], "batcher": dict(self.batch_stats),
                    "shared_workers": self.remote,
                    "worker_stats": {w: dict(v) for w, v in self.worker_stats.items()}}
