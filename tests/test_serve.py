"""Node-level launcher: several front-end processes on one port, shared stores (api/serve.py)."""
import http.client
import json
import os
import time

import pytest

from distributed_llms_amd.api import serve
from distributed_llms_amd.api.multipart import encode_multipart
from distributed_llms_amd.pdf import synth
from distributed_llms_amd.store.results import FileResultStore, make_result_document


def test_gpu_slices():
    assert serve.split_gpus(8, 4) == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert serve.split_gpus(8, 3) == [[0, 1, 2], [3, 4, 5], [6, 7]]
    assert serve.split_gpus(1, 3) == [[0], [], []]
    assert serve.split_gpus(0, 2) == [[], []]


def test_file_result_store_roundtrip(tmp_path):
    st = FileResultStore(str(tmp_path / "r"))
    st.connect()
    assert st.ping() == {"ok": 1} and st.find("nope") is None
    doc = make_result_document("a/b c", "hello world")
    st.upsert(doc)
    got = st.find("a/b c")
    assert got["combined_text"] == "hello world" and got["word_count"] == 2 and got["status"] == "completed"
    assert got["processed_at"] == doc["processed_at"]          # datetime restored
    st.upsert({"document_id": "a/b c", "combined_text": "x"})
    assert st.find("a/b c")["combined_text"] == "x" and st.find("a/b c")["word_count"] == 2


def _get(port, path):
    c = http.client.HTTPConnection("127.0.0.1", port, timeout=60)
    c.request("GET", path)
    r = c.getresponse()
    body = r.read()
    hdr = dict((k.lower(), v) for k, v in r.getheaders())
    c.close()
    return r.status, body, hdr


@pytest.mark.timeout(180)
def test_two_front_ends_share_one_port_and_the_stores(tmp_path, monkeypatch):
    monkeypatch.setenv("OCR_OUTPUT_DIR", str(tmp_path / "out"))
    monkeypatch.setenv("OCR_CPU_MODEL", "tiny")
    monkeypatch.setenv("OCR_DOTENV", "/nonexistent")
    monkeypatch.delenv("REDIS_HOST", raising=False)
    monkeypatch.delenv("MONGO_CONNECTION_STRING", raising=False)
    sup = serve.Supervisor(0, "127.0.0.1", 2, gpus=0).start()
    try:
        assert sup.redis is not None
        seen = set()
        t0 = time.time()
        while time.time() - t0 < 120 and len(seen) < 2:          # both children come up and take connections
            try:
                st, body, hdr = _get(sup.port, "/health")
                if st == 200:
                    assert json.loads(body)["status"] == "healthy"
                    seen.add(hdr.get("x-ocr-frontend"))
            except OSError:
                time.sleep(0.2)
        assert len(seen) == 2, seen
        pdf = synth.make_text_pdf([["The launcher test page.", "Second line of text here for the classifier to count."] * 4])
        ids = [f"doc-{i}" for i in range(6)]
        for d in ids:
            body, ctype = encode_multipart({"document_id": d}, {"file": (f"{d}.pdf", pdf, "application/pdf")})
            c = http.client.HTTPConnection("127.0.0.1", sup.port, timeout=60)
            c.request("POST", "/extract", body=body, headers={"content-type": ctype})
            r = c.getresponse()
            assert r.status == 200 and json.loads(r.read()) == {"status": "ok"}
            c.close()
        # any front end answers for any document (fresh connection per poll -> the kernel picks the process)
        for d in ids:
            for _ in range(400):
                st, body, _ = _get(sup.port, f"/check-status/{d}")
                if st == 200 and json.loads(body)["status"] == "completed":
                    break
                time.sleep(0.02)
            for _ in range(4):
                st, body, _ = _get(sup.port, f"/result/{d}")
                assert st == 200 and "launcher test page" in json.loads(body)["text"]
        # a front end that dies is replaced
        victim = sup.children[0]
        victim.kill()
        victim.wait(timeout=10)
        sup.poll_once()
        assert sup.children[0].pid != victim.pid and sup.restarts == 1
    finally:
        sup.shutdown()


def test_gpu_count_never_touches_the_gpu_runtime(tmp_path, monkeypatch):
    from distributed_llms_amd.api import serve

    for var in ("OCR_NUM_GPUS", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    # a fake KFD topology: one CPU node (no SIMDs) and three GPU nodes
    for k, simd in enumerate((0, 1024, 1024, 1024)):
        d = tmp_path / str(k)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {16 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    assert serve.count_kfd_gpus(str(tmp_path)) == 3
    assert serve.count_kfd_gpus(str(tmp_path / "missing")) == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    assert serve.count_gpus() == 2
    monkeypatch.setenv("OCR_NUM_GPUS", "5")
    assert serve.count_gpus() == 5
    import sys
    assert "torch" not in serve.count_gpus.__code__.co_names
