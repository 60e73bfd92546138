"""Worker pool on CPU: page sharding over 2 worker processes, ordered gather, dead-worker recovery."""
import os
import time

import numpy as np
import pytest

from distributed_llms_amd.pdf import synth
from distributed_llms_amd.runtime.batcher import DynamicBatcher, page_cost_bytes, shape_key
from distributed_llms_amd.runtime.extractor import PoolExtractor
from distributed_llms_amd.runtime.workers import WorkerPool

SPEC = {"model": "tiny", "cfg": dict(det_limit_side=160, rec_max_width=256, rec_width_quantum=64, max_batch_pages=2)}


def _scan_pdf(n_pages, codec="flate"):
    pages = [synth.render_page(i, size=(248, 350), font_px=10, margin=16)[0] for i in range(n_pages)]
    return synth.make_scanned_pdf(pages, codec=codec, dpi=30)


def test_batcher_buckets_budget_and_deadline():
    b = DynamicBatcher(max_items=4, max_wait_ms=40.0, budget_bytes=10 * page_cost_bytes(350, 248, 160, 128))
    ka, kb = shape_key(1344, 960), shape_key(640, 480)
    for i in range(5):
        b.push(ka, ("a", i), page_cost_bytes(350, 248, 160, 128))
    b.push(kb, ("b", 0), 1)
    key, items = b.pop(0)
    assert key == ka and items == [("a", i) for i in range(4)]           # a full bucket is ready at once
    assert b.pop(0) == (None, [])                                          # the rest must wait for the deadline
    t0 = time.perf_counter()
    key, items = b.pop(500)
    assert items and 0.02 < time.perf_counter() - t0 < 0.3                 # released by max_wait, not by timeout
    got = {items[0]}
    got.add(b.pop(500)[1][0])
    assert got == {("a", 4), ("b", 0)}
    # byte budget splits a bucket even below max_items
    b2 = DynamicBatcher(max_items=8, max_wait_ms=1.0, budget_bytes=250)
    for i in range(4):
        b2.push(1, i, 100)
    assert b2.pop(100, flush=True)[1] == [0, 1] and b2.pop(100, flush=True)[1] == [2, 3]
    b2.close()
    assert b2.pop(100) == (None, [])
    st = b.stats()
    assert st["pushed"] == 6 and st["popped"] == 6 and st["full_batches"] == 1 and st["deadline_batches"] == 2


@pytest.fixture(scope="module")
def pool():
    p = WorkerPool(["cpu", "cpu"], SPEC, max_wait_ms=5.0, min_pages_per_shard=2, doc_deadline_s=120.0).start(300.0)
    yield p
    p.stop()


def test_pages_are_sharded_and_gathered_in_order(pool):
    pdf = _scan_pdf(5)
    ex = PoolExtractor(pool)
    doc = ex.extract_document(pdf, "d1")
    assert doc.mode == "ocr" and [p.page_index for p in doc.pages] == [0, 1, 2, 3, 4]
    assert pool.describe()["pages_done"] >= 5
    # both workers took part: 5 pages / min 2 per shard -> 2 shards
    again = ex.extract_document(pdf, "d2")
    assert again.text == doc.text                                          # deterministic across shardings
    # several documents at once share worker batches
    futs = [pool.submit(pdf, 5) for _ in range(3)]
    for f in futs:
        assert sorted(f.result(timeout=120)) == [0, 1, 2, 3, 4]


def test_text_documents_never_touch_the_workers(pool):
    before = pool.describe()["pages_done"]
    lines = ["This born-digital page has more than enough characters to be classified as text." for _ in range(4)]
    doc = PoolExtractor(pool).extract_document(synth.make_text_pdf([lines]), "t1")
    assert doc.mode == "txt" and "born-digital" in doc.text
    assert pool.describe()["pages_done"] == before


def test_unreadable_document_fails_cleanly(pool):
    res = PoolExtractor(pool).extract(b"%PDF-1.4 garbage without objects", "bad")
    assert res["status"] == "failed" and res["error"]


def test_dead_worker_is_respawned_and_its_shards_requeued():
    p = WorkerPool(["cpu", "cpu"], SPEC, fault_inject="crash_after_tasks=1", max_wait_ms=5.0,
                   min_pages_per_shard=2, doc_deadline_s=180.0).start(300.0)
    try:
        pdf = _scan_pdf(4)
        first = p.submit(pdf, 4).result(timeout=180)       # worker 0 finishes its shard, then dies
        assert sorted(first) == [0, 1, 2, 3]
        second = p.submit(pdf, 4).result(timeout=180)      # its next shard is re-queued / it is respawned
        assert sorted(second) == [0, 1, 2, 3]
        t_end = time.time() + 60
        while p.describe()["restarts"] < 1 and time.time() < t_end:
            time.sleep(0.1)
        d = p.describe()
        assert d["restarts"] >= 1
    finally:
        p.stop()


def test_hung_worker_is_detected_by_missing_progress_and_replaced():
    """Worker 0's engine loop wedges after its first shard (its heartbeat thread keeps beating, as it would with the
    main thread stuck in a GPU call).  The pool notices that the progress counter stands still, kills and respawns
    the worker, and the shard finishes on the other one."""
    p = WorkerPool(["cpu", "cpu"], SPEC, fault_inject="hang_after_tasks=1", max_wait_ms=5.0, min_pages_per_shard=2,
                   doc_deadline_s=120.0, progress_timeout_s=3.0).start(300.0)
    try:
        pdf = _scan_pdf(4)
        assert sorted(p.submit(pdf, 4).result(timeout=120)) == [0, 1, 2, 3]
        t0 = time.time()
        assert sorted(p.submit(pdf, 4).result(timeout=120)) == [0, 1, 2, 3]     # worker 0 hangs on this one
        assert time.time() - t0 < 60, "the hang must be caught by the progress timeout, not the document deadline"
        t_end = time.time() + 60
        while p.describe()["restarts"] < 1 and time.time() < t_end:
            time.sleep(0.1)
        assert p.describe()["restarts"] >= 1
        assert sorted(p.submit(pdf, 4).result(timeout=120)) == [0, 1, 2, 3]
    finally:
        p.stop()


def test_worker_that_cannot_start_is_not_respawned_forever():
    """A worker that dies at initialisation (here: an engine spec it cannot build) is retried with exponential
    backoff a bounded number of times; start() then fails AND stops the supervisor."""
    bad = {"model": "tiny", "cfg": dict(no_such_engine_option=1)}
    p = WorkerPool(["cpu"], bad, max_restarts=2)
    t0 = time.time()
    with pytest.raises(RuntimeError):
        p.start(120.0)
    assert time.time() - t0 < 110
    assert p.describe()["disabled"] == 1 and p.restarts <= 2
    assert p._stop.is_set()
    n = p.restarts
    time.sleep(1.5)
    assert p.restarts == n, "no respawn after start() failed"


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_gpu_worker_processes_shard_a_scanned_document():
    """Two worker PROCESSES on the GPU (both on cuda:0 on a one-GPU box - the 8-GPU node gives each its own device):
    shared-memory hand-off of the PDF, page shards, ordered gather; full-size models with the shipped weights, and
    the gathered text must equal what the in-process engine reads."""
    from pathlib import Path

    from distributed_llms_amd.runtime.document import DocumentPipeline
    from distributed_llms_amd.runtime.workers import build_engine

    wd = Path(__file__).resolve().parents[1] / "weights"
    spec = {"model": "full", "cfg": dict(max_batch_pages=4), "weights_dir": str(wd) if wd.exists() else None}
    rendered = [synth.render_page(500 + i)[0] for i in range(3)]
    pages = [rendered[i % 3] for i in range(12)]
    pdf = synth.make_scanned_pdf(pages, codec="jpeg")
    pool = WorkerPool(["cuda:0", "cuda:0"], spec, max_wait_ms=5.0, min_pages_per_shard=4, doc_deadline_s=240.0).start(240.0)
    try:
        doc = PoolExtractor(pool).extract_document(pdf, "g1")
        assert doc.mode == "ocr" and [p.page_index for p in doc.pages] == list(range(12))
        d = pool.describe()
        assert d["pages_done"] >= 12 and d["alive"] == 2
    finally:
        pool.stop()
    ref = DocumentPipeline(build_engine("cuda:0", spec)).process_pdf(pdf, "ref")
    assert [len(p.lines) for p in doc.pages] == [len(p.lines) for p in ref.pages]
    assert doc.text == ref.text


def test_inline_formula_setting_selects_the_detector_trained_for_it(tmp_path, monkeypatch):
    """OCR_INLINE_FORMULAS reaches the engine configuration, and an engine built with it loads ``dbnet_inline.safetensors``
    (when the weights directory has one) instead of ``dbnet.safetensors``."""
    from distributed_llms_amd.config import Settings
    from distributed_llms_amd.runtime import workers
    from distributed_llms_amd.utils import weights as W

    assert Settings.from_env(env={}).inline_formulas is False
    assert Settings.from_env(env={"OCR_INLINE_FORMULAS": "true"}).inline_formulas is True
    for name in ("dbnet.safetensors", "dbnet_inline.safetensors", "crnn.safetensors"):
        (tmp_path / name).write_bytes(b"")
    loaded = []
    monkeypatch.setattr(W, "load_model_weights", lambda model, path, strict=True: loaded.append(os.path.basename(path)))
    for flag, want in ((False, "dbnet.safetensors"), (True, "dbnet_inline.safetensors")):
        loaded.clear()
        eng = workers.build_engine("cpu", {"model": "full", "weights_dir": str(tmp_path), "cfg": dict(inline_formulas=flag)})
        assert eng.cfg.inline_formulas is flag and loaded[0] == want and "crnn.safetensors" in loaded
    (tmp_path / "dbnet_inline.safetensors").unlink()
    loaded.clear()
    workers.build_engine("cpu", {"model": "full", "weights_dir": str(tmp_path), "cfg": dict(inline_formulas=True)})
    assert loaded[0] == "dbnet.safetensors"
