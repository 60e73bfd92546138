"""API contract tests against the ASGI app with
in-memory stores.  The CPU extractor runs the real pipeline: text PDFs through the native reader,
scanned PDFs through tiny DBNet/CRNN models on the CPU oracle path."""
import json
import time

import pytest
from starlette.testclient import TestClient

from distributed_llms_amd.api.app import create_app
from distributed_llms_amd.api.multipart import encode_multipart
from distributed_llms_amd.config import Settings
from distributed_llms_amd.pdf import synth
from distributed_llms_amd.store.results import MemoryResultStore
from distributed_llms_amd.store.status import MemoryStatusStore

LINES = ["The quick brown fox jumps over the lazy dog near the river bank today.",
         "A second line of perfectly ordinary born-digital text follows the first one.", "",
         "Another paragraph starts here and it is long enough to be counted as text."]


class StubExtractor:
    def __init__(self, result=None, delay=0.0):
        self.result = result or {"status": "completed", "result": "stub text", "pages": 1}
        self.delay = delay
        self.calls = 0

    def extract(self, pdf_bytes, document_id, document_hash=None):
        self.calls += 1
        if self.delay:
            time.sleep(self.delay)
        return dict(self.result)

    def describe(self):
        return {"backend": "stub"}

    def close(self):
        pass


def _cpu_extractor():
    from distributed_llms_amd.runtime.document import DocumentPipeline
    from distributed_llms_amd.runtime.extractor import InlineExtractor
    from distributed_llms_amd.runtime.workers import build_engine

    eng = build_engine("cpu", {"model": "tiny", "cfg": dict(det_limit_side=160, rec_max_width=256,
                                                             rec_width_quantum=64, max_batch_pages=2)})
    return InlineExtractor(DocumentPipeline(eng, decode_threads=2))


@pytest.fixture()
def ctx(tmp_path):
    settings = Settings(store="memory", output_dir=str(tmp_path / "out"), num_gpus=0, node_id="test-node")
    status, results = MemoryStatusStore(), MemoryResultStore()
    return settings, status, results


def _post(client, pdf, doc_id, filename="doc.pdf", fields=None):
    f = {"document_id": doc_id} if doc_id is not None else {}
    f.update(fields or {})
    body, ctype = encode_multipart(f, {"file": (filename, pdf, "application/pdf")})
    return client.post("/extract", content=body, headers={"content-type": ctype})


def _wait(client, doc_id, timeout=60.0):
    t_end = time.time() + timeout
    while time.time() < t_end:
        r = client.get(f"/check-status/{doc_id}")
        if r.status_code == 200 and r.json()["status"] != "processing":
            return r.json()["status"]
        time.sleep(0.02)
    raise AssertionError("job did not finish")


def test_extract_text_pdf_end_to_end(ctx):
    settings, status, results = ctx
    app = create_app(settings, status, results, _cpu_extractor())
    pdf = synth.make_text_pdf([LINES, ["Page two has a single line of text that is long enough."]])
    with TestClient(app) as c:
        r = _post(c, pdf, "doc-1")
        assert r.status_code == 200 and r.json() == {"status": "ok"}
        assert _wait(c, "doc-1") == "completed"
        r = c.get("/result/doc-1")
        assert r.status_code == 200
        body = r.json()
        assert set(body) == {"document_id", "text", "status"} and body["status"] == "completed"
        assert body["text"].startswith("The quick brown fox") and "Page two has a single line" in body["text"]
        assert "\n\n" in body["text"]  # content-list items are joined with blank lines
    # Mongo "checkpoint format": exactly six fields with the reference's types
    doc = results.find("doc-1")
    assert set(doc) == {"document_id", "combined_text", "status", "processed_at", "word_count", "character_count"}
    assert doc["status"] == "completed" and doc["word_count"] == len(doc["combined_text"].split())
    assert doc["character_count"] == len(doc["combined_text"])
    # final Redis value holds the full text, the hash, no preview; hash key points at the id
    st = status.get("doc-1")
    assert st["status"] == "completed" and st["result"] == doc["combined_text"] and "result_preview" not in st
    assert status.document_for_hash(st["document_hash"]) == "doc-1"


def test_extract_scanned_pdf_goes_through_ocr_branch(ctx):
    settings, status, results = ctx
    app = create_app(settings, status, results, _cpu_extractor())
    pg, truth = synth.render_page(5, size=(248, 350), font_px=10, margin=16)
    pdf = synth.make_scanned_pdf([pg], codec="flate", dpi=30)
    with TestClient(app) as c:
        assert _post(c, pdf, "scan-1").json() == {"status": "ok"}
        assert _wait(c, "scan-1") in ("completed", "error")
        r = c.get("/result/scan-1")
        assert r.status_code == 200 and r.json()["status"] in ("completed", "failed")


def test_rejects_non_pdf_and_empty_id_and_missing_fields(ctx):
    settings, status, results = ctx
    app = create_app(settings, status, results, StubExtractor())
    with TestClient(app) as c:
        r = _post(c, b"hello", "x", filename="notes.txt")
        assert r.status_code == 400 and r.json() == {"detail": "Only PDF files are accepted"}
        r = _post(c, b"%PDF-1.4", "", filename="a.PDF")
        assert r.status_code == 400 and r.json() == {"detail": "document_id is required"}
        r = _post(c, b"%PDF-1.4", None)
        assert r.status_code == 422
        r = c.post("/extract", content=b"garbage", headers={"content-type": "multipart/form-data; boundary=zz"})
        assert r.status_code == 422
        assert _post(c, b"%PDF-1.4", "up", filename="UPPER.PDF").status_code == 200  # suffix check is case-insensitive


def test_duplicate_id_returns_raw_status_and_does_not_reprocess(ctx):
    settings, status, results = ctx
    ex = StubExtractor(delay=0.3)
    app = create_app(settings, status, results, ex)
    with TestClient(app) as c:
        assert _post(c, b"%PDF-1.4 x", "dup").json() == {"status": "ok"}
        again = _post(c, b"%PDF-1.4 x", "dup").json()
        assert again["status"] in ("queued", "processing")      # raw lifecycle word leaks, as in the reference
        assert _wait(c, "dup") == "completed"
        assert _post(c, b"%PDF-1.4 x", "dup").json() == {"status": "completed"}
        assert ex.calls == 1


def test_failed_job_status_vocabulary_and_error_file(ctx, tmp_path):
    settings, status, results = ctx
    app = create_app(settings, status, results, StubExtractor({"status": "failed", "error": "boom"}))
    with TestClient(app) as c:
        _post(c, b"%PDF-1.4", "bad")
        assert _wait(c, "bad") == "error"                        # public word
        r = c.get("/result/bad")
        assert r.status_code == 200 and r.json() == {
            "document_id": "bad", "status": "failed", "message": "Document processing status: failed"}
        assert _post(c, b"%PDF-1.4", "bad").json() == {"status": "failed"}   # cannot be retried until the TTL lapses
    err = json.loads((tmp_path / "out" / "error_bad.json").read_text())
    assert err["document_id"] == "bad" and err["status"] == "failed" and err["error"] == "boom"
    assert results.find("bad") is None                           # failed documents never reach the result store


def test_real_parse_failure_becomes_failed_status(ctx):
    settings, status, results = ctx
    app = create_app(settings, status, results, _cpu_extractor())
    with TestClient(app) as c:
        _post(c, b"this is not a pdf", "garbled")
        assert _wait(c, "garbled") == "error"
        assert "cannot parse PDF" in status.get("garbled")["error"]


def test_unknown_ids(ctx):
    settings, status, results = ctx
    app = create_app(settings, status, results, StubExtractor())
    with TestClient(app) as c:
        r = c.get("/check-status/nope")
        assert r.status_code == 404 and r.json() == {"status": "unknown", "error": "No record found for document ID: nope"}
        r = c.get("/result/nope")
        assert r.status_code == 404 and r.json() == {"error": "No record found for document ID: nope"}


def test_result_survives_expired_status_key(ctx):
    settings, status, results = ctx
    app = create_app(settings, status, results, StubExtractor())
    with TestClient(app) as c:
        _post(c, b"%PDF-1.4", "old")
        assert _wait(c, "old") == "completed"
        # "completed" is written twice (preview, then the full text): wait for the second write, or it lands after
        # the delete below and revives the key
        t_end = time.time() + 30
        while "result_preview" in (status.get("old") or {}) and time.time() < t_end:
            time.sleep(0.01)
        status.delete("old")                                      # TTL lapsed; the Mongo row never expires
        assert c.get("/check-status/old").status_code == 404
        r = c.get("/result/old")
        assert r.status_code == 200 and r.json() == {"document_id": "old", "text": "stub text", "status": "completed"}


def test_health_states(ctx):
    settings, status, results = ctx
    app = create_app(settings, status, results, StubExtractor())
    with TestClient(app) as c:
        r = c.get("/health")
        assert r.status_code == 200
        b = r.json()
        assert b["status"] == "healthy" and b["version"] == "1.0.0" and "timestamp" in b
        assert b["dependencies"] == {"redis": "healthy", "mongodb": "healthy"}
        results.healthy = False
        r = c.get("/health")
        assert r.status_code == 503 and r.json()["status"] == "unhealthy" and r.json()["dependencies"]["mongodb"] == "unhealthy"
        results.healthy = True
        status.healthy = False
        r = c.get("/health")
        assert r.status_code == 503 and r.json()["dependencies"]["redis"] == "unhealthy"
        # /result with the status store down -> 500 with the reference's error body
        r = c.get("/result/x")
        assert r.status_code == 500 and r.json()["error"].startswith("Error retrieving result:")
        status.healthy = True
        results.ping = lambda: {"ok": 0}
        r = c.get("/health")
        assert r.status_code == 200 and r.json()["status"] == "degraded"


def test_write_order_result_row_exists_before_completed(ctx):
    """The result store is written before any status says 'completed'."""
    settings, status, results = ctx
    seen = []
    orig = status.update

    def spy(document_id, data):
        seen.append((data["status"], results.find(document_id) is not None, "result_preview" in data, "result" in data))
        return orig(document_id, data)

    status.update = spy
    app = create_app(settings, status, results, StubExtractor())
    with TestClient(app) as c:
        _post(c, b"%PDF-1.4", "ord")
        assert _wait(c, "ord") == "completed"
    assert [s[0] for s in seen] == ["processing", "completed", "completed"]
    assert seen[1] == ("completed", True, True, False) and seen[2] == ("completed", True, False, True)


def test_metrics_and_cors(ctx):
    settings, status, results = ctx
    app = create_app(settings, status, results, StubExtractor())
    with TestClient(app) as c:
        _post(c, b"%PDF-1.4", "m1")
        _wait(c, "m1")
        text = c.get("/metrics").text
        assert "ocr_documents_total" in text and "ocr_stage_seconds_bucket" in text
        r = c.options("/extract", headers={"origin": "http://example.com", "access-control-request-method": "POST"})
        assert r.headers.get("access-control-allow-origin") in ("*", "http://example.com")


def test_orphan_sweep_on_startup(ctx):
    settings, status, results = ctx
    status.update("stuck", {"status": "processing", "owner": "test-node"})
    status.update("other", {"status": "processing", "owner": "someone-else"})
    app = create_app(settings, status, results, StubExtractor())
    with TestClient(app) as c:
        assert c.get("/check-status/stuck").json() == {"status": "error"}
        assert c.get("/check-status/other").json() == {"status": "processing"}


# ------------------------------------------------------------------ streaming multipart reader
def test_streaming_form_matches_one_shot_parser_at_any_chunking():
    import hashlib
    import random
    from concurrent.futures import ThreadPoolExecutor

    from distributed_llms_amd.api.multipart import StreamingForm, encode_multipart, parse_multipart

    rnd = random.Random(5)
    data = bytes(rnd.getrandbits(8) for _ in range(70_000)) + b"\r\n--not-the-boundary\r\n" + bytes(3000)
    body, ctype = encode_multipart({"document_id": "doc-1", "note": "x" * 10}, {"file": ("a.pdf", data, "application/pdf")})
    want = parse_multipart(body, ctype)
    pool = ThreadPoolExecutor(2)
    for known_length in (True, False):
        for _ in range(6):
            form = StreamingForm(ctype, 1 << 20, len(body) if known_length else None, hash_pool=pool)
            form.HASH_PIECE = 4096
            pos = 0
            while pos < len(body):
                step = rnd.choice([1, 7, 100, 5000, 40_000])
                form.feed(body[pos : pos + step])
                pos += step
            parts = form.finish()
            assert [(p.name, p.filename, bytes(p.data)) for p in parts] == [(p.name, p.filename, p.data) for p in want]
            assert form.sha256_hex() == hashlib.sha256(data).hexdigest()


def test_streaming_form_can_leave_the_hash_thread_running():
    """``finish(defer_hash=True)`` returns while the last pieces are still being hashed; the future carries the digest."""
    import hashlib
    import threading
    from concurrent.futures import ThreadPoolExecutor

    from distributed_llms_amd.api.multipart import StreamingForm, encode_multipart

    data = bytes(range(256)) * 400
    body, ctype = encode_multipart({"document_id": "doc-1"}, {"file": ("a.pdf", data, "application/pdf")})
    pool = ThreadPoolExecutor(2)
    gate = threading.Event()
    pool.submit(gate.wait)                      # both hash workers are held up: the digest cannot be ready at finish()
    pool.submit(gate.wait)
    form = StreamingForm(ctype, 1 << 20, len(body), hash_pool=pool)
    form.HASH_PIECE = 4096
    for k in range(0, len(body), 10_000):
        form.feed(body[k : k + 10_000])
    parts = form.finish(defer_hash=True)
    fut = form.sha256_pending()
    assert fut is not None and not fut.done()
    assert bytes(next(p for p in parts if p.is_file).data) == data
    gate.set()
    assert fut.result(timeout=10) == hashlib.sha256(data).hexdigest()
    assert form.sha256_hex() == hashlib.sha256(data).hexdigest()
    # without a streamed hash there is nothing to wait for
    form2 = StreamingForm(ctype, 1 << 20, None, hash_pool=pool)
    form2.feed(body)
    form2.finish(defer_hash=True)
    assert form2.sha256_pending() is None and form2.sha256_hex() == hashlib.sha256(data).hexdigest()


def test_job_started_with_a_pending_digest_records_it(ctx):
    """A job handed a digest future starts extraction at once and writes the digest with its status records."""
    import asyncio
    from concurrent.futures import Future

    from distributed_llms_amd.jobs.manager import JobManager
    from distributed_llms_amd.store.results import ErrorSink

    settings, status, results = ctx
    jobs = JobManager(status, results, ErrorSink(settings.output_dir), StubExtractor(), node_id="test-node")
    fut: Future = Future()

    async def run():
        assert await jobs.claim("late-hash", None, "a.pdf") is None
        task = jobs.start(b"%PDF-1.4 late digest", "late-hash", None, hash_future=fut)
        await asyncio.sleep(0.05)
        assert jobs.status.get("late-hash")["status"] == "queued"      # 'processing' waits for the digest
        fut.set_result("ab" * 32)
        return await task

    out = asyncio.run(run())
    assert out["status"] == "completed"
    rec = jobs.status.get("late-hash")
    assert rec["status"] == "completed" and rec["document_hash"] == "ab" * 32


def test_upload_cap_holds_without_content_length(ctx, monkeypatch):
    """A chunked upload (no Content-Length) is cut off at the cap while it streams, not buffered to the end."""
    import distributed_llms_amd.api.app as appmod
    from distributed_llms_amd.api.multipart import encode_multipart

    monkeypatch.setattr(appmod, "MAX_UPLOAD_BYTES", 50_000)
    body, ctype = encode_multipart({"document_id": "big"}, {"file": ("big.pdf", b"%PDF-" + bytes(200_000), "application/pdf")})

    def gen():
        for k in range(0, len(body), 8192):
            yield body[k : k + 8192]

    settings, status, results = ctx
    app = create_app(settings, status, results, _cpu_extractor())
    with TestClient(app) as c:
        r = c.post("/extract", content=gen(), headers={"content-type": ctype})
        assert r.status_code == 413
        assert c.get("/check-status/big").status_code == 404


def test_metrics_export_engine_stage_timers_and_worker_memory(ctx):
    """GET /metrics carries the engine loop's per-stage seconds (in-process engine) and, for a GPU pool, the last
    stage timers and HBM figures every worker reported."""
    from distributed_llms_amd.utils import metrics

    settings, status, results = ctx
    app = create_app(settings, status, results, _cpu_extractor())
    pg, _ = synth.render_page(5, size=(248, 350), font_px=10, margin=16)
    pdf = synth.make_scanned_pdf([pg], codec="flate", dpi=30)
    with TestClient(app) as c:
        assert _post(c, pdf, "scan-m").json() == {"status": "ok"}
        _wait(c, "scan-m")
        text = c.get("/metrics").text
    assert 'ocr_engine_stage_seconds_total{stage="wait_det",worker="0"}' in text
    assert 'ocr_engine_stage_seconds_total{stage="submit_det",worker="0"}' in text

    class PoolLike:
        def describe(self):
            return {"backend": "gpu-pool", "alive": 2, "restarts": 1, "queue_depth": [3, 0],
                    "batcher": {"batches": 4, "full_batches": 3},
                    "worker_stats": {0: {"stage_seconds": {"wait_det": 1.5}, "hbm_allocated_bytes": 7 << 30,
                                         "hbm_reserved_bytes": 9 << 30},
                                     1: {"stage_seconds": {"wait_det": 0.25}, "hbm_allocated_bytes": 6 << 30}}}

    metrics.refresh_from_extractor(PoolLike())
    text = metrics.render().decode()
    assert 'ocr_engine_stage_seconds_total{stage="wait_det",worker="1"} 0.25' in text
    assert 'ocr_gpu_memory_bytes{gpu="0",kind="reserved"} 9.663676416e+09' in text
    assert 'ocr_gpu_memory_bytes{gpu="1",kind="allocated"}' in text
    assert "ocr_batch_fill_ratio 0.75" in text


def test_page_warnings_reach_the_status_record_and_the_metrics(ctx):
    """A completed document whose pages were damaged or read with low confidence: the status record lists the warnings and
    ``ocr_page_warnings_total`` counts them by reason."""
    from distributed_llms_amd.utils import metrics

    settings, status, results = ctx
    warn = ["page 2: page image is damaged: 10 of 900 JPEG blocks could be decoded; no text is reported for this page",
            "page 3: text read with low confidence (median 0.71 over 18 lines): the page may be in a script other than Latin, "
            "Cyrillic or Greek, or too degraded to read"]
    before = {r: metrics.PAGE_WARNINGS.labels(r)._value.get() for r in ("damaged", "low_confidence")}
    app = create_app(settings, status, results, StubExtractor({"status": "completed", "result": "some text", "pages": 3,
                                                                "warnings": warn}))
    with TestClient(app) as c:
        _post(c, b"%PDF-1.4 warn", "w1")
        assert _wait(c, "w1") == "completed"
        t_end = time.time() + 10.0          # the record turns "completed" with the stored result, the job's own update follows
        while "warnings" not in status.get("w1") and time.time() < t_end:
            time.sleep(0.02)
        assert status.get("w1")["warnings"] == warn
        assert c.get("/result/w1").json() == {"document_id": "w1", "text": "some text", "status": "completed", "warnings": warn}
        text = c.get("/metrics").text
        assert 'ocr_page_warnings_total{reason="low_confidence"}' in text and 'ocr_page_warnings_total{reason="damaged"}' in text
    assert metrics.PAGE_WARNINGS.labels("damaged")._value.get() == before["damaged"] + 1
    assert metrics.PAGE_WARNINGS.labels("low_confidence")._value.get() == before["low_confidence"] + 1
