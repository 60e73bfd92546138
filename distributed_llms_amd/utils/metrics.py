"""Prometheus metrics: documents, pages, per-stage latency, queue depth, batch fill,
per-GPU memory.  The reference exposes no application metrics at all (only host metrics through
the CloudWatch agent, terraform/cloudwatch-agent-config.json); ``GET /metrics`` is new."""
from __future__ import annotations

from prometheus_client import CollectorRegistry, Counter, Gauge, Histogram, generate_latest

REGISTRY = CollectorRegistry()

DOCUMENTS = Counter("ocr_documents_total", "Documents finished, by outcome", ["outcome"], registry=REGISTRY)
PAGES = Counter("ocr_pages_total", "Pages extracted", registry=REGISTRY)
REQUESTS = Counter("ocr_http_requests_total", "HTTP requests", ["route", "code"], registry=REGISTRY)
JOBS_IN_FLIGHT = Gauge("ocr_jobs_in_flight", "Background extraction jobs running", registry=REGISTRY)
QUEUE_DEPTH = Gauge("ocr_worker_queue_pages", "Pages queued per GPU worker", ["worker"], registry=REGISTRY)
WORKERS_ALIVE = Gauge("ocr_workers_alive", "GPU worker processes alive", registry=REGISTRY)
WORKER_RESTARTS = Gauge("ocr_worker_restarts", "GPU worker respawns since start", registry=REGISTRY)
BATCH_FILL = Gauge("ocr_batch_fill_ratio", "Full batches / all batches emitted by the dynamic batcher", registry=REGISTRY)
GPU_MEM = Gauge("ocr_gpu_memory_bytes", "HBM in use per GPU", ["gpu"], registry=REGISTRY)
STAGE_SECONDS = Histogram(
    "ocr_stage_seconds", "Latency per pipeline stage", ["stage"], registry=REGISTRY,
    buckets=(0.005, 0.01, 0.025, 0.05, 0.1, 0.25, 0.5, 1, 2.5, 5, 10, 30, 60, 120, 300, 600))


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
        b = d.get("batcher") or {}
        if b.get("batches"):
            BATCH_FILL.set(b.get("full_batches", 0) / b["batches"])
    try:
        import torch

        if torch.cuda.is_available() and d.get("backend") == "inline":
            GPU_MEM.labels("0").set(torch.cuda.memory_allocated(0))
    except Exception:
        pass


def render() -> bytes:
    return generate_latest(REGISTRY)
