"""Typed settings object.

Same environment variable names and defaults as the reference
(services/ocr_service/ocr_service.py:32-34, 56-58, 610 and the never-imported
services/ocr_service/config.py:1-32), with two deliberate differences:

* ``MONGO_CONNECTION_STRING`` has **no default credential** (the reference
  embeds a live-looking Atlas password as its default, ocr_service.py:32).
  Unset means: use the in-memory result store (``OCR_STORE=memory``) or fail
  startup loudly when ``OCR_STORE=redis+mongo`` is requested explicitly.
* new ``OCR_*`` knobs for the GPU engine (number of GPUs, batching, models).

``.env`` files are read by a small built-in parser (python-dotenv is not a
dependency).
"""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass, field, fields
from pathlib import Path
from typing import Dict, Mapping, Optional

logger = logging.getLogger("distributed_llms_amd")

LOG_FORMAT = "%(asctime)s - %(levelname)s - %(message)s"  # ocr_service.py:28
SERVICE_VERSION = "1.0.0"  # ocr_service.py:568
STATUS_TTL_SECONDS = 86400  # ocr_service.py:72,77
PREVIEW_CHARS = 500  # ocr_service.py:377
IO_THREADS = 10  # ocr_service.py:44


def configure_logging(level: int = logging.INFO) -> None:
    """Root logger -> stderr in the reference's format (like ``logging.basicConfig``: nothing happens when the root
    logger already has handlers).  Records travel through a queue to one writer thread, so a request or job thread
    never waits for the stream lock or the write (four INFO lines per document, dozens of threads)."""
    import atexit
    import logging.handlers
    import queue

    root = logging.getLogger()
    if root.handlers:
        return
    stream = logging.StreamHandler()
    stream.setFormatter(logging.Formatter(LOG_FORMAT))
    q: "queue.SimpleQueue" = queue.SimpleQueue()
    listener = logging.handlers.QueueListener(q, stream)
    listener.start()
    atexit.register(listener.stop)
    root.addHandler(logging.handlers.QueueHandler(q))
    root.setLevel(level)


def load_dotenv(path: os.PathLike | str = ".env", override: bool = False, environ: Optional[dict] = None) -> Dict[str, str]:
    """Minimal ``KEY=VALUE`` loader (comments, blank lines, optional quotes, ``export`` prefix)."""
    env = os.environ if environ is None else environ
    out: Dict[str, str] = {}
    p = Path(path)
    if not p.is_file():
        return out
    for raw in p.read_text(encoding="utf-8").splitlines():
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        if line.startswith("export "):
            line = line[len("export "):].lstrip()
        if "=" not in line:
            continue
        key, val = line.split("=", 1)
        key, val = key.strip(), val.strip()
        if len(val) >= 2 and val[0] == val[-1] and val[0] in "\"'":
            val = val[1:-1]
        elif " #" in val:
            val = val.split(" #", 1)[0].rstrip()
        out[key] = val
        if override or key not in env:
            env[key] = val
    return out


def _get_int(env: Mapping[str, str], key: str, default: int) -> int:
    v = env.get(key)
    if v is None or v == "":
        return default
    try:
        return int(v)
    except ValueError as e:
        raise ValueError(f"environment variable {key}={v!r} is not an integer") from e


def _get_float(env: Mapping[str, str], key: str, default: float) -> float:
    v = env.get(key)
    if v is None or v == "":
        return default
    return float(v)


def _get_bool(env: Mapping[str, str], key: str, default: bool) -> bool:
    v = env.get(key)
    if v is None or v == "":
        return default
    return v.strip().lower() in ("1", "true", "yes", "on")


def _shipped_weights():
    """The weights this repository trains on its synthetic page family (train/page_trainer.py) ship under
    ``<repo>/weights``; they are the default when OCR_WEIGHTS_DIR is unset ("none" forces random init)."""
    import pathlib

    wd = pathlib.Path(__file__).resolve().parents[1] / "weights"
    return str(wd) if (wd / "dbnet.safetensors").exists() and (wd / "crnn.safetensors").exists() else None


@dataclass
class Settings:
    # --- reference variables -------------------------------------------------
    mongo_connection_string: Optional[str] = None
    mongo_db_name: str = "ocr_service"
    mongo_collection_name: str = "documents"
    redis_host: str = "localhost"
    redis_port: int = 6379
    redis_password: Optional[str] = None
    port: int = 8000
    # --- store selection ------------------------------------------------------
    store: str = "auto"              # auto | redis+mongo | redis+file | memory
    mongo_tls: Optional[bool] = None  # None -> TLS iff mongodb+srv:// (Atlas), as the reference always used TLS
    status_ttl: int = STATUS_TTL_SECONDS
    output_dir: str = "output"
    # --- engine -----------------------------------------------------------------
    num_gpus: int = -1               # -1: all visible; 0: CPU engine
    gpu_ids: Optional[str] = None    # "2,3": the devices this process drives (multi-front-end deployments)
    max_batch_pages: int = 36
    max_wait_ms: float = 20.0
    # CPython hands the GIL over every `switch interval`; at the default 5 ms the engine thread (hundreds of kernel
    # launches per batch) waits that long every time the HTTP thread holds it, which dominates small-batch latency
    gil_switch_ms: float = 0.5
    det_limit_side: int = 1344
    recognizer: str = "crnn"         # crnn | trocr
    dtype: str = "bf16"
    rec_height: int = 48
    weights_dir: Optional[str] = None
    doc_deadline_s: float = 600.0
    hbm_budget_gb: float = 200.0
    use_hip_graphs: bool = True
    inline_formulas: bool = False    # formulas inside text lines are cut out and read as LaTeX ($...$ in the line's text)
    node_id: str = field(default_factory=lambda: os.uname().nodename)
    fault_inject: str = ""
    force_pool: bool = False         # use the worker-process pool even with a single GPU
    worker_sockets: Optional[str] = None   # "path,path,...": Unix sockets of the node's shared worker servers (api/serve.py)
    home_worker: int = 0             # index into worker_sockets of the worker next to this front end (breaks load ties)
    engine_model: str = "full"       # full | stub (no model, no GPU: front-end capacity tests, bench/frontend_ceiling.py)
    stub_pages_per_s: float = 1000.0
    cpu_model: str = "full"          # CPU oracle path model size: full | tiny (tests)

    @classmethod
    def from_env(cls, env: Optional[Mapping[str, str]] = None, dotenv: Optional[str] = None) -> "Settings":
        if dotenv:
            load_dotenv(dotenv)
        e = os.environ if env is None else env
        s = cls(
            mongo_connection_string=e.get("MONGO_CONNECTION_STRING") or None,
            mongo_db_name=e.get("MONGO_DB_NAME", "ocr_service"),
            mongo_collection_name=e.get("MONGO_COLLECTION_NAME", "documents"),
            redis_host=e.get("REDIS_HOST", "localhost"),
            redis_port=_get_int(e, "REDIS_PORT", 6379),
            redis_password=e.get("REDIS_PASSWORD") or None,
            port=_get_int(e, "PORT", 8000),
            store=e.get("OCR_STORE", "auto"),
            mongo_tls=None if e.get("MONGO_TLS") in (None, "") else _get_bool(e, "MONGO_TLS", True),
            status_ttl=_get_int(e, "OCR_STATUS_TTL", STATUS_TTL_SECONDS),
            output_dir=e.get("OCR_OUTPUT_DIR", "output"),
            num_gpus=_get_int(e, "OCR_NUM_GPUS", -1),
            gpu_ids=e.get("OCR_GPU_IDS") or None,
            max_batch_pages=_get_int(e, "OCR_MAX_BATCH_PAGES", 36),
            max_wait_ms=_get_float(e, "OCR_MAX_WAIT_MS", 20.0),
            gil_switch_ms=_get_float(e, "OCR_GIL_SWITCH_MS", 0.5),
            det_limit_side=_get_int(e, "OCR_DET_LIMIT_SIDE", 1344),
            recognizer=e.get("OCR_RECOGNIZER", "crnn"),
            dtype=e.get("OCR_DTYPE", "bf16"),
            rec_height=_get_int(e, "OCR_REC_HEIGHT", 48),
            weights_dir=e.get("OCR_WEIGHTS_DIR") or _shipped_weights(),
            doc_deadline_s=_get_float(e, "OCR_DOC_DEADLINE_S", 600.0),
            hbm_budget_gb=_get_float(e, "OCR_HBM_BUDGET_GB", 200.0),
            use_hip_graphs=_get_bool(e, "OCR_USE_HIP_GRAPHS", True),
            inline_formulas=_get_bool(e, "OCR_INLINE_FORMULAS", False),
            fault_inject=e.get("FAULT_INJECT", ""),
            force_pool=_get_bool(e, "OCR_FORCE_POOL", False),
            worker_sockets=e.get("OCR_WORKER_SOCKETS") or None,
            home_worker=_get_int(e, "OCR_HOME_WORKER", 0),
            engine_model=e.get("OCR_ENGINE_MODEL", "full"),
            stub_pages_per_s=_get_float(e, "OCR_STUB_PAGES_PER_S", 1000.0),
            cpu_model=e.get("OCR_CPU_MODEL", "full"),
            **({"node_id": e["OCR_NODE_ID"]} if e.get("OCR_NODE_ID") else {}),
        )
        s.validate()
        return s

    def validate(self) -> None:
        if self.store not in ("auto", "redis+mongo", "redis+file", "memory"):
            raise ValueError(f"OCR_STORE must be auto|redis+mongo|redis+file|memory, got {self.store!r}")
        if self.recognizer not in ("crnn", "trocr"):
            raise ValueError(f"OCR_RECOGNIZER must be crnn|trocr, got {self.recognizer!r}")
        if self.dtype not in ("bf16",):
            raise ValueError("OCR_DTYPE: only bf16 is implemented on the MFMA path")
        if self.store == "redis+mongo" and not self.mongo_connection_string:
            raise ValueError(
                "OCR_STORE=redis+mongo requires MONGO_CONNECTION_STRING (no default credential is shipped)"
            )
        if not self.mongo_connection_string and self.store == "auto":
            logger.warning("MONGO_CONNECTION_STRING not set: results are kept in the in-memory store")

    @property
    def image_dir(self) -> str:
        return os.path.join(self.output_dir, "images")

    def resolved_store(self) -> str:
        if self.store != "auto":
            return self.store
        return "redis+mongo" if self.mongo_connection_string else "memory"

    def describe(self) -> Dict[str, object]:
        d = {f.name: getattr(self, f.name) for f in fields(self)}
        for k in ("mongo_connection_string", "redis_password"):
            if d.get(k):
                d[k] = "***"
        return d
