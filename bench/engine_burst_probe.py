#!/usr/bin/env python3
"""Probe: the burst workload of bench/service_bench.py (32 closed-loop clients x 6 four-page scans) straight into the
in-process extractor, without HTTP - separates what the engine loop sustains on small documents from what the
front end costs.  Prints the engine's stage times and batch sizes."""
import json
import os
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from distributed_llms_amd.config import Settings
from distributed_llms_amd.pdf import synth
from distributed_llms_amd.runtime.extractor import make_extractor


def main():
    clients = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    pages_each = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    os.environ["OCR_DOTENV"] = "/nonexistent"
    ex = make_extractor(Settings.from_env())
    rendered = [synth.render_page(i, size=synth.A4_300DPI, font_px=42, margin=220) for i in range(8)]
    enc = [synth.encode_page_jpeg(r[0]) for r in rendered]
    pdfs = []
    for k in range(8):
        idx = [(k + j) % 8 for j in range(pages_each)]
        pdfs.append(synth.make_scanned_pdf([rendered[i][0] for i in idx], codec="jpeg", encoded=[enc[i] for i in idx]))
    for k in range(3):
        ex.extract_document(pdfs[k], f"warm{k}")
    lat = []
    lock = threading.Lock()

    def client(c):
        for r in range(rounds):
            t0 = time.perf_counter()
            res = ex.extract_document(pdfs[(c + r) % 8], f"c{c}r{r}")
            assert res.error is None and len(res.pages) == pages_each
            with lock:
                lat.append(time.perf_counter() - t0)

    before = ex.describe()
    ths = [threading.Thread(target=client, args=(c,)) for c in range(clients)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    after = ex.describe()
    lat.sort()
    out = {"clients": clients, "rounds": rounds, "pages_each": pages_each, "pages_per_s": round(clients * rounds * pages_each / dt, 1),
           "p50_s": round(lat[len(lat) // 2], 3), "p95_s": round(lat[int(len(lat) * 0.95)], 3)}
    for k in ("engine_batches", "mean_batch_pages", "stage_seconds"):
        if k in after:
            a, b = after[k], before.get(k)
            out[k] = ({s: round(a[s] - (b or {}).get(s, 0.0), 3) for s in a} if isinstance(a, dict)
                      else (a - b if isinstance(b, (int, float)) else a))
    print(json.dumps(out), flush=True)
    ex.close()


if __name__ == "__main__":
    main()
