"""The layers between the socket and the engines, on CPU with stub engines: document lifetime when one page is bad,
packed shard results, shared worker servers reached by several front ends, load-aware dispatch, CPU placement."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from distributed_llms_amd.pdf import synth
from distributed_llms_amd.runtime import wire
from distributed_llms_amd.runtime.document import DocumentPipeline, assemble_document, assemble_packed
from distributed_llms_amd.runtime.engine import Figure, PageResult, TextLine
from distributed_llms_amd.runtime.extractor import InlineExtractor, PoolExtractor
from distributed_llms_amd.runtime.shm import Segment
from distributed_llms_amd.runtime.stub import StubEngine
from distributed_llms_amd.runtime.workers import ShmDocument, WorkerPool
from distributed_llms_amd.utils import affinity

STUB = {"model": "stub", "pages_per_s": 400.0, "lines_per_page": 12, "cfg": dict(max_batch_pages=4)}


def _jpeg_pdf(n_pages, bad_last=False):
    pages = [synth.render_page(i, size=(248, 350), font_px=10, margin=16)[0] for i in range(min(n_pages, 3))]
    pdf = synth.make_scanned_pdf([pages[i % len(pages)] for i in range(n_pages)], codec="jpeg", dpi=30)
    if bad_last:
        k = pdf.rfind(b"/DCTDecode")
        assert k > 0
        pdf = pdf[:k] + b"/JPXDecode" + pdf[k + 10:]         # same length: the last page's image is now a codec we do not read
    return pdf


def test_one_unreadable_page_fails_the_document_and_nothing_else():
    """40 JPEG pages, the last one's filter rewritten to JPXDecode.  The stub engine takes encoded streams like the
    GPU engine does, so queued pages hold views into the shared-memory segment while the failure is reported: the
    error must reach the front end, the worker must stay alive (no restart), and the next document must work."""
    pool = WorkerPool(["cpu"], STUB, max_wait_ms=2.0, doc_deadline_s=60.0, progress_timeout_s=20.0).start(120.0)
    try:
        ex = PoolExtractor(pool)
        res = ex.extract(_jpeg_pdf(40, bad_last=True), "mixed")
        assert res["status"] == "failed" and "unsupported image encoding" in res["error"]
        good = ex.extract_document(_jpeg_pdf(12), "good")
        assert len(good.pages) == 12 and all(len(p.lines) == 12 for p in good.pages)
        again = ex.extract(_jpeg_pdf(9, bad_last=True), "mixed2")
        assert again["status"] == "failed"
        assert len(ex.extract_document(_jpeg_pdf(5), "good2").pages) == 5
        d = pool.describe()
        assert d["restarts"] == 0 and d["alive"] == 1
    finally:
        pool.stop()
    leftovers = [n for n in os.listdir("/dev/shm") if n.startswith(f"dlamd-{os.getpid()}-")]
    assert not leftovers, leftovers


def test_inline_extractor_keeps_the_document_open_until_its_pages_drained():
    ext = InlineExtractor(DocumentPipeline(StubEngine(200.0, 8)), max_wait_ms=2.0, doc_deadline_s=30.0)
    try:
        for _ in range(3):
            res = ext.extract(_jpeg_pdf(40, bad_last=True), "mixed")
            assert res["status"] == "failed" and "unsupported image encoding" in res["error"]
        ok = ext.extract(_jpeg_pdf(6), "fine")
        assert ok["status"] == "completed" and ok["pages"] == 6
        assert ext.service.load() == 0
    finally:
        ext.close()


def test_cancelled_document_stops_using_the_engine():
    from distributed_llms_amd.pdf import reader

    ext = InlineExtractor(DocumentPipeline(StubEngine(50.0, 4)), max_wait_ms=1.0)
    try:
        pdf = _jpeg_pdf(150)
        doc = reader.PdfDocument(pdf)
        fut = ext.service.submit(doc, list(range(150)))
        time.sleep(0.1)
        ext.service.cancel(fut, "deadline")
        with pytest.raises(RuntimeError):
            fut.result(timeout=5)
        assert fut.drained.wait(10.0)
        done = ext.pipeline.engine.pages_done
        # at 50 pages/s the document needs 3 s; the two batches the engine already held go through, the rest is dropped
        assert done <= 2 * ext.pipeline.engine.cfg.max_batch_pages < 150
        doc.close()
        time.sleep(0.3)
        assert ext.pipeline.engine.pages_done == done
    finally:
        ext.close()


def _page(i):
    pr = PageResult("j", i, 2480, 3508, n_detected=3)
    y = 100
    for k in range(5):
        pr.lines.append(TextLine((100, y, 900, y + 40), f"line {k} of page {i} é€", 0.9, block=k // 3))
        y += 50 if k != 2 else 200
    pr.lines.append(TextLine((100, y, 600, y + 80), "x^2", 0.8, 2, 0.0, "equation"))
    pr.figures.append(Figure((10, 10, 200, 200), 2, np.zeros((20, 20), np.uint8)))
    return pr


class _Writer:
    def __init__(self):
        self.files = {}

    def write(self, path, data):
        self.files[path] = bytes(data)


def test_packed_shard_round_trip_matches_the_object_path():
    blob = wire.pack_shard([(0, _page(0)), (1, None), (2, _page(2))])
    sh = wire.PackedShard(blob)
    assert sh.n_pages == 3 and sh.n_lines == 12 and not sh.present(1)
    w1, w2 = _Writer(), _Writer()
    a = assemble_document("doc 1", "ocr", [_page(0), _page(2)], None, w1)
    b = assemble_packed("doc 1", "ocr", [sh], {}, 3, None, w2)
    assert a.text == b.text and a.content_list == b.content_list and a.markdown == b.markdown
    assert w1.files.keys() == w2.files.keys() and all(v[:4] == b"\x89PNG" for v in w2.files.values())
    assert len(b.pages) == 2 and [ln.text for ln in b.pages[1].lines] == [ln.text for ln in a.pages[1].lines]
    assert b.pages[0].lines[5].kind == "equation" and b.pages[0].lines[0].box == (100.0, 100.0, 900.0, 140.0)
    with pytest.raises(ValueError):
        wire.PackedShard(b"nope" + blob[4:])


def test_upload_in_shared_memory_is_parsed_in_place_and_not_copied():
    pdf = _jpeg_pdf(10)
    seg = Segment.create(len(pdf) + 64)
    seg.buf[17 : 17 + len(pdf)] = pdf                    # as received: multipart framing around the file part
    pool = WorkerPool(["cpu"], STUB, max_wait_ms=2.0).start(120.0)
    try:
        before = set(os.listdir("/dev/shm"))
        doc = PoolExtractor(pool).extract_document(ShmDocument(seg, 17, len(pdf)), "shm")
        assert len(doc.pages) == 10
        assert set(os.listdir("/dev/shm")) == before         # no second segment was made for it
        assert os.path.exists(os.path.join("/dev/shm", seg.name))   # and the caller's one is the caller's to remove
    finally:
        pool.stop()
        seg.release()
    assert not os.path.exists(os.path.join("/dev/shm", seg.name))


def _start_server(tmp, wid, spec, extra=()):
    sock = os.path.join(tmp, f"w{wid}.sock")
    p = subprocess.Popen([sys.executable, "-m", "distributed_llms_amd.runtime.worker_server", "--id", str(wid), "--device", "cpu",
                          "--socket", sock, "--spec", json.dumps(spec), *extra],
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    return p, sock


def test_front_ends_share_worker_servers_and_balance_on_reported_load(tmp_path):
    """Two worker servers, two front-end pools.  Front end B's documents land on the worker that front end A is NOT
    keeping busy (the load a worker reports covers all its front ends); a killed server's shards finish elsewhere."""
    slow = dict(STUB, pages_per_s=60.0)
    procs = [_start_server(str(tmp_path), w, slow) for w in range(2)]
    socks = ["unix:" + s for _, s in procs]
    a = b = None
    try:
        a = WorkerPool(socks, min_pages_per_shard=64, home=0, doc_deadline_s=60.0).start(120.0)
        b = WorkerPool(socks, min_pages_per_shard=64, home=0, doc_deadline_s=60.0).start(120.0)
        pdf = _jpeg_pdf(30)
        fa = a.submit(pdf, 30)                          # 0.5 s of work on worker 0 (home of both front ends)
        time.sleep(0.45)                                # > one heartbeat: B has heard of it
        fb = b.submit(pdf, 30)
        with b.mu:
            placed = [w for w in range(2) if b.inflight[w]]
        assert placed == [1], "B must avoid the worker A keeps busy"
        assert len(fa.result(timeout=60)) == 30 and len(fb.result(timeout=60)) == 30
        assert fb.result().n_lines == 30 * 12
        # kill worker 0 under a document that runs there; its shard is re-dispatched to worker 1
        f = a.submit(pdf, 30)
        time.sleep(0.1)
        procs[0][0].kill()
        assert len(f.result(timeout=60)) == 30
        assert a.describe()["alive"] == 1
    finally:
        for pool in (a, b):
            if pool is not None:
                pool.stop()
        for p, _ in procs:
            p.kill()
            p.wait()


def test_worker_server_exits_when_its_engine_stops_making_progress(tmp_path):
    p, sock = _start_server(str(tmp_path), 0, STUB, ("--fault", "hang_after_tasks=1", "--progress-timeout", "1.5"))
    pool = None
    try:
        pool = WorkerPool(["unix:" + sock], doc_deadline_s=5.0, max_attempts=1).start(120.0)
        # the hook bites between batches: the first document gets through, the second wedges after its first batch
        assert len(pool.submit(_jpeg_pdf(4), 4).result(timeout=30)) == 4
        f = pool.submit(_jpeg_pdf(16), 16)
        with pytest.raises(RuntimeError):
            f.result(timeout=30)
        assert p.wait(timeout=30) == 19
    finally:
        if pool is not None:
            pool.stop()
        p.kill()
        p.wait()


def test_cpu_placement_prefers_the_gpu_local_cores():
    assert affinity.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    lists = [list(range(0, 64))] * 4 + [list(range(64, 128))] * 4           # two sockets, four GPUs each
    allowed = list(range(128))
    got = [affinity.cpus_for_worker(i, 8, allowed, lists) for i in range(8)]
    assert got[0] == list(range(0, 16)) and got[3] == list(range(48, 64)) and got[4] == list(range(64, 80))
    assert sorted(c for g in got for c in g) == allowed                       # disjoint and complete
    # no sysfs view: an even split of what the process may use
    got = [affinity.cpus_for_worker(i, 4, list(range(8)), []) for i in range(4)]
    assert got == [[0, 1], [2, 3], [4, 5], [6, 7]]
    # fewer cores than workers: everybody gets everything rather than nothing
    assert affinity.cpus_for_worker(5, 8, [0, 1], []) == [0, 1]
