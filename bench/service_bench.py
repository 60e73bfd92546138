#!/usr/bin/env python3
"""Service-level benchmark through real HTTP.

Starts the ASGI app under uvicorn as a child process (in-memory stores unless REDIS_* / MONGO_* are set), then
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

import http.client
import subprocess

import numpy as np

from distributed_llms_amd.api.multipart import encode_multipart
from distributed_llms_amd.pdf import synth


class Client:
    """Keep-alive stdlib HTTP client: the bench measures the service, so the client must cost next to nothing
    (httpx in the server's own process spent more time encoding the 124 MB upload than the service took)."""

    def __init__(self, port):
        self.port = port
        self.c = http.client.HTTPConnection("127.0.0.1", port, timeout=600)

    def request(self, method, path, body=None, headers=None):
        for attempt in (0, 1):
            try:
                self.c.request(method, path, body=body, headers=headers or {})
                r = self.c.getresponse()
                return r.status, r.read()
            except (http.client.HTTPException, ConnectionError, OSError):
                self.c.close()
                self.c = http.client.HTTPConnection("127.0.0.1", self.port, timeout=600)
                if attempt:
                    raise

    def get_json(self, path):
        st, data = self.request("GET", path)
        return st, json.loads(data)


def run_doc(client, pdf_body, ctype, doc_id, poll=0.002):
    """Upload, poll /check-status every ``poll`` seconds, fetch /result.  -> (latency, status, text, upload time)."""
    t0 = time.perf_counter()
    st, data = client.request("POST", "/extract", body=pdf_body, headers={"content-type": ctype})
    assert st == 200 and json.loads(data) == {"status": "ok"}, data
    t1 = time.perf_counter()
    while True:
        s = client.get_json(f"/check-status/{doc_id}")[1]["status"]
        if s != "processing":
            break
        time.sleep(poll)
    res = client.get_json(f"/result/{doc_id}")[1]
    return time.perf_counter() - t0, s, res.get("text", ""), t1 - t0


def body_for(pdf, doc_id):
    return encode_multipart({"document_id": doc_id}, {"file": (f"{doc_id}.pdf", pdf, "application/pdf")})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--port", type=int, default=8123)
    ap.add_argument("--text-docs", type=int, default=200)
    ap.add_argument("--scan-pages", type=int, default=100)
    ap.add_argument("--scan-docs", type=int, default=4)
    ap.add_argument("--scan-clients", type=int, default=3, help="concurrent clients of the sustained 100-page phase (0: skip)")
    ap.add_argument("--scan-rounds", type=int, default=4, help="documents each of them uploads, one after another")
    ap.add_argument("--burst-docs", type=int, default=32)
    ap.add_argument("--burst-pages", type=int, default=4)
    ap.add_argument("--burst-rounds", type=int, default=6, help="documents each concurrent client submits, one after another")
    ap.add_argument("--burst-poll-ms", type=float, default=10.0, help="status polling interval of the concurrent clients")
    ap.add_argument("--num-gpus", type=int, default=-1)
    ap.add_argument("--force-pool", action="store_true", help="use the worker-process pool even with one GPU")
    ap.add_argument("--frontends", type=int, default=2, help="--server serve: front-end processes on the port")
    ap.add_argument("--server", choices=["dlamd", "uvicorn", "serve"], default="dlamd",
                    help="HTTP server in front of the application: api/httpd.py (default) or uvicorn")
    args = ap.parse_args()

    # the service runs as its own process, exactly as deployed: uvicorn + the module-level ASGI app
    env = dict(os.environ, OCR_NUM_GPUS=str(args.num_gpus), OCR_OUTPUT_DIR="/tmp/ocr_bench_out",
               OCR_FORCE_POOL="1" if args.force_pool else "0", OCR_DOTENV="/nonexistent")
    if args.server == "serve":
        # the node launcher: front-end processes on one port, every one connected to the GPU worker server(s) over Unix
        # sockets; uploads received into shared memory, results returned packed (api/serve.py)
        env.pop("OCR_NUM_GPUS")
        env.pop("OCR_FORCE_POOL")
        cmd = [sys.executable, "-m", "distributed_llms_amd.api.serve", "--host", "127.0.0.1", "--port", str(args.port),
               "--frontends", str(args.frontends), "--embedded-redis"]
    elif args.server == "uvicorn":
        cmd = [sys.executable, "-m", "uvicorn", "distributed_llms_amd.api.asgi:app", "--host", "127.0.0.1",
               "--port", str(args.port), "--log-level", "warning"]
    else:
        cmd = [sys.executable, "-m", "distributed_llms_amd.api.httpd", "distributed_llms_amd.api.asgi:app", "--host", "127.0.0.1",
               "--port", str(args.port)]
    server = subprocess.Popen(cmd, env=env, cwd=str(Path(__file__).resolve().parents[1]))
    client = Client(args.port)
    for _ in range(1200):
        try:
            if client.request("GET", "/health")[0] in (200, 503):
                break
        except Exception:
            time.sleep(0.25)
    out = {}

    # ---- config 1: single-page text PDFs (plumbing only)
    lines = [f"Line {i}: the quick brown fox jumps over the lazy dog and keeps running for a while." for i in range(40)]
    pdf = synth.make_text_pdf([lines])
    lat = [run_doc(client, *body_for(pdf, f"text-{i}"), f"text-{i}")[0] for i in range(args.text_docs)]
    lat = np.array(lat[5:]) * 1e3
    out["config1_text_pdf"] = {"docs": len(lat), "p50_ms": round(float(np.percentile(lat, 50)), 2),
                               "p95_ms": round(float(np.percentile(lat, 95)), 2), "mean_ms": round(float(lat.mean()), 2)}

    # ---- config 2: 100-page scans
    pages_truth = [synth.render_page(i) for i in range(10)]
    rendered = [p for p, _ in pages_truth]
    enc = [synth.encode_page_jpeg(p, restart_blocks=0) for p in rendered]     # as scanners write them: no restart markers
    n = args.scan_pages
    scan = synth.make_scanned_pdf([rendered[i % 10] for i in range(n)], codec="jpeg", encoded=[enc[i % 10] for i in range(n)])
    res = [run_doc(client, *body_for(scan, f"scan-{i}"), f"scan-{i}") for i in range(args.scan_docs)]
    best = min(r[0] for r in res[1:]) if len(res) > 1 else res[0][0]
    from distributed_llms_amd.train.trainer import _edit_distance

    # accuracy of what the service returned: share of ground-truth lines found verbatim in the text, and the
    # character error rate of the first page's worth of text (edit distance is quadratic, so not on 230k chars)
    got = res[-1][2]
    norm = lambda t: " ".join(t.split())
    got_n = norm(got)
    truth_lines = [norm(t) for i in range(n) for t in pages_truth[i % 10][1].texts]
    line_hit = sum(1 for t in truth_lines if t in got_n) / max(len(truth_lines), 1)
    want0 = norm(" ".join(pages_truth[0][1].texts))
    cer = _edit_distance(got_n[: len(want0)], want0) / max(len(want0), 1)
    out["config2_scan_100p"] = {"pdf_mb": round(len(scan) / 1e6, 1), "latencies_s": [round(r[0], 3) for r in res],
                                "upload_s": [round(r[3], 3) for r in res], "status": res[-1][1],
                                "pages_per_s": round(n / best, 1), "text_chars": len(got),
                                "truth_lines_found_verbatim": round(line_hit, 4), "cer_first_page": round(cer, 5)}

    # ---- config 2, sustained: several clients upload 100-page scans at the same time, so one document's upload
    # overlaps another's recognition (a lone document pays upload + engine time back to back)
    if args.scan_clients > 0:
        sb = [[body_for(scan, f"cscan-{i}-{r}") for r in range(args.scan_rounds)] for i in range(args.scan_clients)]
        slat = []

        def big(i):
            c = Client(args.port)
            for r in range(args.scan_rounds):
                slat.append(run_doc(c, *sb[i][r], f"cscan-{i}-{r}", poll=0.005)[0])

        t0 = time.perf_counter()
        ths = [threading.Thread(target=big, args=(i,)) for i in range(args.scan_clients)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        wall = time.perf_counter() - t0
        out["config2_concurrent_scans"] = {"clients": args.scan_clients, "docs_per_client": args.scan_rounds, "pages_each": n,
                                           "pages_per_s": round(args.scan_clients * args.scan_rounds * n / wall, 1),
                                           "p50_s": round(float(np.percentile(slat, 50)), 3)}
        del sb

    # ---- burst of small scans from concurrent clients
    small = synth.make_scanned_pdf([rendered[i % 10] for i in range(args.burst_pages)], codec="jpeg",
                                   encoded=[enc[i % 10] for i in range(args.burst_pages)])
    lats = [0.0] * (args.burst_docs * args.burst_rounds)

    bodies = [[body_for(small, f"burst-{i}-{r}") for r in range(args.burst_rounds)] for i in range(args.burst_docs)]

    def one(i):
        c = Client(args.port)
        for r in range(args.burst_rounds):
            lats[i * args.burst_rounds + r] = run_doc(c, *bodies[i][r], f"burst-{i}-{r}", poll=args.burst_poll_ms / 1000.0)[0]

    t0 = time.perf_counter()
    ths = [threading.Thread(target=one, args=(i,)) for i in range(args.burst_docs)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    wall = time.perf_counter() - t0
    out["burst_small_scans"] = {"clients": args.burst_docs, "docs_per_client": args.burst_rounds, "pages_each": args.burst_pages,
                                "poll_ms": args.burst_poll_ms,
                                "pages_per_s": round(args.burst_docs * args.burst_rounds * args.burst_pages / wall, 1),
                                "p50_s": round(float(np.percentile(lats, 50)), 3), "p95_s": round(float(np.percentile(lats, 95)), 3)}
    out["metrics_sample"] = [l for l in client.request("GET", "/metrics")[1].decode().splitlines()
                             if l.startswith(("ocr_pages_total", "ocr_documents_total", "ocr_batch_fill", "ocr_workers_alive"))]
    print(json.dumps(out), flush=True)
    server.terminate()
    try:
        server.wait(timeout=15)
    except subprocess.TimeoutExpired:
        server.kill()


if __name__ == "__main__":
    main()
