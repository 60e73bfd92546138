"""Prometheus metrics: documents, pages, per-stage latency, queue depth, batch fill,
per-GPU memory.  The reference exposes no application metrics at all (only host metrics through
the CloudWatch agent, terraform/cloudwatch-agent-config.json); ``GET /metrics`` is new."""
from __future__ import annotations

from prometheus_client import CollectorRegistry, Counter, Gauge, Histogram, generate_latest

REGISTRY = CollectorRegistry()

DOCUMENTS = Counter("ocr_documents_total", "Documents finished, by outcome", ["outcome"], registry=REGISTRY)
PAGES = Counter("ocr_pages_total", "Pages extracted", registry=REGISTRY)
PAGE_WARNINGS = Counter("ocr_page_warnings_total", "Pages flagged in a completed document, by reason "
                        "(damaged: image stream not decodable to the end; low_confidence: unknown script or unreadable scan)",
                        ["reason"], registry=REGISTRY)
REQUESTS = Counter("ocr_http_requests_total", "HTTP requests", ["route", "code"], registry=REGISTRY)
JOBS_IN_FLIGHT = Gauge("ocr_jobs_in_flight", "Background extraction jobs running", registry=REGISTRY)
QUEUE_DEPTH = Gauge("ocr_worker_queue_pages", "Pages queued per GPU worker", ["worker"], registry=REGISTRY)
WORKERS_ALIVE = Gauge("ocr_workers_alive", "GPU worker processes alive", registry=REGISTRY)
WORKER_RESTARTS = Gauge("ocr_worker_restarts", "GPU worker respawns since start", registry=REGISTRY)
BATCH_FILL = Gauge("ocr_batch_fill_ratio", "Full batches / all batches emitted by the dynamic batcher", registry=REGISTRY)
GPU_MEM = Gauge("ocr_gpu_memory_bytes", "HBM held by the engine process of each GPU worker", ["gpu", "kind"], registry=REGISTRY)
ENGINE_STAGE = Gauge("ocr_engine_stage_seconds_total", "Cumulative wall-clock seconds the engine loop spent per stage",
                     ["worker", "stage"], registry=REGISTRY)
STAGE_SECONDS = Histogram(
    "ocr_stage_seconds", "Latency per pipeline stage", ["stage"], registry=REGISTRY,
    buckets=(0.005, 0.01, 0.025, 0.05, 0.1, 0.25, 0.5, 1, 2.5, 5, 10, 30, 60, 120, 300, 600))


def _export_engine_stats(worker: str, st: dict) -> None:
    for stage, sec in (st.get("stage_seconds") or {}).items():
        ENGINE_STAGE.labels(worker, stage).set(sec)
    for kind in ("allocated", "reserved"):
        v = st.get(f"hbm_{kind}_bytes")
        if v is not None:
            GPU_MEM.labels(worker, kind).set(v)


def refresh_from_extractor(extractor) -> None:
    """Pull gauges from the extraction back end right before a scrape."""
    try:
        d = extractor.describe()
    except Exception:
        return
    if "alive" in d:
        WORKERS_ALIVE.set(d["alive"])
        WORKER_RESTARTS.set(d.get("restarts", 0))
        for i, q in enumerate(d.get("queue_depth", [])):
            QUEUE_DEPTH.labels(str(i)).set(q)
        for w, st in (d.get("worker_stats") or {}).items():
            _export_engine_stats(str(w), st)
    else:
        _export_engine_stats("0", d)      # in-process engine
    b = d.get("batcher") or d
    if b.get("batches"):
        BATCH_FILL.set(b.get("full_batches", 0) / b["batches"])


def render() -> bytes:
    return generate_latest(REGISTRY)
