#!/usr/bin/env python3
"""Service-level benchmark through real HTTP (BASELINE configs 1 and 2 as a client sees them).

Starts the ASGI app under uvicorn in this process (in-memory stores unless REDIS_* / MONGO_* are set), then
  1. config 1: N single-page born-digital PDFs through /extract -> /check-status -> /result: p50 / p95 latency;
  2. config 2: a 100-page 300-DPI scanned PDF through the same three calls: end-to-end latency and pages/s;
  3. a burst of small scans from concurrent clients (cross-document batching in the GPU workers).
"""
import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import httpx
import numpy as np
import uvicorn

from distributed_llms_amd.api.app import create_app
from distributed_llms_amd.api.multipart import encode_multipart
from distributed_llms_amd.config import Settings
from distributed_llms_amd.pdf import synth


def run_doc(client, base, pdf, doc_id, poll=0.002):
    body, ctype = encode_multipart({"document_id": doc_id}, {"file": (f"{doc_id}.pdf", pdf, "application/pdf")})
    t0 = time.perf_counter()
    r = client.post(f"{base}/extract", content=body, headers={"content-type": ctype})
    assert r.status_code == 200 and r.json() == {"status": "ok"}, r.text
    while True:
        s = client.get(f"{base}/check-status/{doc_id}").json()["status"]
        if s != "processing":
            break
        time.sleep(poll)
    res = client.get(f"{base}/result/{doc_id}").json()
    return time.perf_counter() - t0, s, len(res.get("text", ""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--port", type=int, default=8123)
    ap.add_argument("--text-docs", type=int, default=200)
    ap.add_argument("--scan-pages", type=int, default=100)
    ap.add_argument("--scan-docs", type=int, default=3)
    ap.add_argument("--burst-docs", type=int, default=32)
    ap.add_argument("--burst-pages", type=int, default=4)
    ap.add_argument("--num-gpus", type=int, default=-1)
    ap.add_argument("--force-pool", action="store_true", help="use the worker-process pool even with one GPU")
    args = ap.parse_args()

    settings = Settings.from_env()
    settings.num_gpus = args.num_gpus
    settings.output_dir = "/tmp/ocr_bench_out"
    settings.force_pool = args.force_pool
    app = create_app(settings)
    server = uvicorn.Server(uvicorn.Config(app, host="127.0.0.1", port=args.port, log_level="warning"))
    th = threading.Thread(target=server.run, daemon=True)
    th.start()
    base = f"http://127.0.0.1:{args.port}"
    client = httpx.Client(timeout=600.0)
    for _ in range(1200):
        try:
            if client.get(f"{base}/health").status_code in (200, 503):
                break
        except Exception:
            time.sleep(0.25)
    out = {}

    # ---- config 1: single-page text PDFs (plumbing only)
    lines = [f"Line {i}: the quick brown fox jumps over the lazy dog and keeps running for a while." for i in range(40)]
    pdf = synth.make_text_pdf([lines])
    lat = [run_doc(client, base, pdf, f"text-{i}")[0] for i in range(args.text_docs)]
    lat = np.array(lat[5:]) * 1e3
    out["config1_text_pdf"] = {"docs": len(lat), "p50_ms": round(float(np.percentile(lat, 50)), 2),
                               "p95_ms": round(float(np.percentile(lat, 95)), 2), "mean_ms": round(float(lat.mean()), 2)}

    # ---- config 2: 100-page scans
    rendered = [synth.render_page(i)[0] for i in range(10)]
    enc = [synth.encode_page_jpeg(p) for p in rendered]
    n = args.scan_pages
    scan = synth.make_scanned_pdf([rendered[i % 10] for i in range(n)], codec="jpeg", encoded=[enc[i % 10] for i in range(n)])
    res = [run_doc(client, base, scan, f"scan-{i}") for i in range(args.scan_docs)]
    best = min(r[0] for r in res[1:]) if len(res) > 1 else res[0][0]
    out["config2_scan_100p"] = {"pdf_mb": round(len(scan) / 1e6, 1), "latencies_s": [round(r[0], 3) for r in res],
                                "status": res[-1][1], "pages_per_s": round(n / best, 1),
                                "note": "random-init detector: the service path recognises whatever boxes it finds"}

    # ---- burst of small scans from concurrent clients
    small = synth.make_scanned_pdf([rendered[i % 10] for i in range(args.burst_pages)], codec="jpeg",
                                   encoded=[enc[i % 10] for i in range(args.burst_pages)])
    lats = [0.0] * args.burst_docs

    def one(i):
        with httpx.Client(timeout=600.0) as c:
            lats[i] = run_doc(c, base, small, f"burst-{i}")[0]

    t0 = time.perf_counter()
    ths = [threading.Thread(target=one, args=(i,)) for i in range(args.burst_docs)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    wall = time.perf_counter() - t0
    out["burst_small_scans"] = {"docs": args.burst_docs, "pages_each": args.burst_pages,
                                "pages_per_s": round(args.burst_docs * args.burst_pages / wall, 1),
                                "p50_s": round(float(np.percentile(lats, 50)), 3), "p95_s": round(float(np.percentile(lats, 95)), 3)}
    out["metrics_sample"] = [l for l in client.get(f"{base}/metrics").text.splitlines()
                             if l.startswith(("ocr_pages_total", "ocr_documents_total", "ocr_batch_fill", "ocr_workers_alive"))]
    print(json.dumps(out), flush=True)
    server.should_exit = True
    th.join(timeout=10)
    os._exit(0)


if __name__ == "__main__":
    main()
