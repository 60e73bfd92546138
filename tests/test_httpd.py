"""The in-repo HTTP/1.1 server (api/httpd.py) in front of the real application and of small ASGI apps: framing
(Content-Length, chunked, 100-continue, pipelining, keep-alive), error answers, and the same API behaviour a client
gets through uvicorn."""
import asyncio
import hashlib
import http.client
import json
import socket
import threading
import time

import pytest

from distributed_llms_amd.api import httpd
from distributed_llms_amd.api.app import create_app
from distributed_llms_amd.api.multipart import encode_multipart
from distributed_llms_amd.config import Settings
from distributed_llms_amd.store.results import MemoryResultStore
from distributed_llms_amd.store.status import MemoryStatusStore


class EchoExtractor:
    """Reports what it received, so a test can tell that every byte of an upload arrived intact."""

    def extract(self, pdf_bytes, document_id, document_hash=None):
        return {"status": "completed", "result": f"{len(pdf_bytes)} {hashlib.sha256(pdf_bytes).hexdigest()}", "pages": 1}

    def describe(self):
        return {"backend": "echo"}

    def close(self):
        pass


class Running:
    def __init__(self, app):
        self.server = httpd.Server(app)
        self.ready = threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        async def main():
            ev = asyncio.Event()

            async def flag():
                await ev.wait()
                self.ready.set()

            asyncio.get_running_loop().create_task(flag())
            await self.server.serve("127.0.0.1", 0, ready=ev)

        asyncio.run(main())

    def __enter__(self):
        self.thread.start()
        assert self.ready.wait(20)
        self.port = self.server.port
        return self

    def __exit__(self, *a):
        self.server.should_exit = True
        self.thread.join(20)


@pytest.fixture()
def service(tmp_path):
    st, rs = MemoryStatusStore(), MemoryResultStore()
    app = create_app(Settings(store="memory", output_dir=str(tmp_path), num_gpus=0, node_id="t"), st, rs, EchoExtractor())
    with Running(app) as r:
        yield r, st, rs


def _wait_done(c, doc_id):
    for _ in range(500):
        c.request("GET", f"/check-status/{doc_id}")
        r = c.getresponse()
        s = json.loads(r.read())["status"]
        if s != "processing":
            return s
        time.sleep(0.01)
    raise AssertionError("not finished")


def test_api_flow_over_keep_alive_connection(service):
    r, st, rs = service
    c = http.client.HTTPConnection("127.0.0.1", r.port, timeout=30)
    pdf = b"%PDF-1.4\n" + bytes(range(256)) * 9000          # 2.3 MB: several body chunks
    body, ct = encode_multipart({"document_id": "d1"}, {"file": ("a.pdf", pdf, "application/pdf")})
    c.request("POST", "/extract", body=body, headers={"content-type": ct})
    resp = c.getresponse()
    assert resp.status == 200 and json.loads(resp.read()) == {"status": "ok"}
    assert resp.getheader("content-type") == "application/json" and resp.getheader("server") == "dlamd"
    assert _wait_done(c, "d1") == "completed"
    c.request("GET", "/result/d1")
    resp = c.getresponse()
    out = json.loads(resp.read())
    assert out == {"document_id": "d1", "text": f"{len(pdf)} {hashlib.sha256(pdf).hexdigest()}", "status": "completed"}
    # unknown id: the polling fast path gives the route's 404 body
    c.request("GET", "/check-status/nope")
    resp = c.getresponse()
    assert resp.status == 404 and json.loads(resp.read()) == {"status": "unknown", "error": "No record found for document ID: nope"}
    # a browser poll (Origin header) goes through the framework and gets its CORS headers
    c.request("GET", "/check-status/d1", headers={"origin": "http://example.com"})
    resp = c.getresponse()
    assert resp.status == 200 and json.loads(resp.read()) == {"status": "completed"}
    assert resp.getheader("access-control-allow-origin") in ("*", "http://example.com")
    c.request("GET", "/health")
    resp = c.getresponse()
    assert resp.status == 200 and json.loads(resp.read())["status"] == "healthy"
    c.request("GET", "/metrics")
    resp = c.getresponse()
    assert resp.status == 200 and b'ocr_http_requests_total{code="404",route="check-status"} ' in resp.read()
    c.close()


def test_chunked_upload_expect_continue_and_pipelining(service):
    r, st, rs = service
    pdf = b"%PDF-1.4\n" + b"x" * 300_000
    body, ct = encode_multipart({"document_id": "c1"}, {"file": ("a.pdf", pdf, "application/pdf")})
    s = socket.create_connection(("127.0.0.1", r.port))
    s.sendall(b"POST /extract HTTP/1.1\r\nhost: x\r\ncontent-type: " + ct.encode() +
              b"\r\ntransfer-encoding: chunked\r\nexpect: 100-continue\r\n\r\n")
    f = s.makefile("rb")
    assert f.readline() == b"HTTP/1.1 100 Continue\r\n" and f.readline() == b"\r\n"
    for k in range(0, len(body), 70_001):                       # odd chunk size: framing straddles TCP segments
        piece = body[k : k + 70_001]
        s.sendall(b"%x;ext=1\r\n" % len(piece) + piece + b"\r\n")
    s.sendall(b"0\r\nx-trailer: 1\r\n\r\n")
    # two pipelined requests in ONE segment right behind the upload
    s.sendall(b"GET /check-status/nope HTTP/1.1\r\nhost: x\r\n\r\nGET /check-status/c1 HTTP/1.1\r\nhost: x\r\n\r\n")

    def read_response():
        status = f.readline()
        hdr = {}
        while True:
            ln = f.readline()
            if ln == b"\r\n":
                break
            k, _, v = ln.decode().partition(":")
            hdr[k.lower()] = v.strip()
        return int(status.split()[1]), hdr, f.read(int(hdr["content-length"]))

    code, _, data = read_response()
    assert code == 200 and json.loads(data) == {"status": "ok"}
    code, _, data = read_response()
    assert code == 404
    code, _, data = read_response()
    assert code == 200 and json.loads(data)["status"] in ("processing", "completed")
    s.close()
    c = http.client.HTTPConnection("127.0.0.1", r.port, timeout=30)
    assert _wait_done(c, "c1") == "completed"
    assert rs.find("c1")["combined_text"] == f"{len(pdf)} {hashlib.sha256(pdf).hexdigest()}"


def test_malformed_requests_get_an_error_and_the_server_lives_on(service):
    r, _, _ = service
    for raw, want in ((b"NOT-HTTP\r\n\r\n", b"400"), (b"GET / HTTP/1.1\r\nbad header\r\n\r\n", b"400"),
                      (b"GET / HTTP/1.1\r\nx: " + b"a" * 70_000 + b"\r\n\r\n", b"431"),
                      (b"POST /extract HTTP/1.1\r\ncontent-length: abc\r\n\r\n", b"400")):
        s = socket.create_connection(("127.0.0.1", r.port))
        s.sendall(raw)
        got = s.recv(4096)
        assert got.startswith(b"HTTP/1.1 " + want), got[:60]
        s.close()
    # a client that drops the connection in the middle of an upload
    s = socket.create_connection(("127.0.0.1", r.port))
    s.sendall(b"POST /extract HTTP/1.1\r\ncontent-type: multipart/form-data; boundary=x\r\ncontent-length: 1000000\r\n\r\n--x\r\n")
    s.close()
    c = http.client.HTTPConnection("127.0.0.1", r.port, timeout=10)
    c.request("GET", "/health")
    assert c.getresponse().status == 200


def test_streaming_response_without_length_is_chunked_and_head_has_no_body():
    async def app(scope, receive, send):
        if scope["type"] == "lifespan":
            msg = await receive()
            await send({"type": "lifespan.startup.complete"})
            await receive()
            await send({"type": "lifespan.shutdown.complete"})
            return
        assert (await receive())["more_body"] is False
        await send({"type": "http.response.start", "status": 200, "headers": [(b"content-type", b"text/plain")]})
        for k in range(3):
            await send({"type": "http.response.body", "body": b"part%d;" % k, "more_body": True})
        await send({"type": "http.response.body", "body": b"", "more_body": False})

    with Running(app) as r:
        c = http.client.HTTPConnection("127.0.0.1", r.port, timeout=10)
        c.request("GET", "/x?y=1")
        resp = c.getresponse()
        assert resp.getheader("transfer-encoding") == "chunked" and resp.read() == b"part0;part1;part2;"
        c.request("HEAD", "/x")
        resp = c.getresponse()
        assert resp.status == 200 and resp.read() == b""
        c.request("GET", "/again")                          # the connection survived both
        assert c.getresponse().read() == b"part0;part1;part2;"


def test_scope_fields_match_what_the_framework_expects():
    seen = {}

    async def app(scope, receive, send):
        if scope["type"] != "http":
            return
        seen.update(scope)
        body = b""
        while True:
            m = await receive()
            body += bytes(m["body"])
            if not m["more_body"]:
                break
        seen["body"] = body
        await send({"type": "http.response.start", "status": 204, "headers": []})
        await send({"type": "http.response.body", "body": b""})

    with Running(app) as r:
        c = http.client.HTTPConnection("127.0.0.1", r.port, timeout=10)
        c.request("PUT", "/a%20b/%C3%A9?q=1&r=%20", body=b"hello", headers={"X-Thing": "Val"})
        assert c.getresponse().status == 204
    assert seen["method"] == "PUT" and seen["path"] == "/a b/é" and seen["raw_path"] == b"/a%20b/%C3%A9"
    assert seen["query_string"] == b"q=1&r=%20" and seen["http_version"] == "1.1" and seen["scheme"] == "http"
    assert (b"x-thing", b"Val") in seen["headers"] and seen["body"] == b"hello"
    assert seen["client"][0] == "127.0.0.1" and seen["server"][1] == r.port


def test_ambiguous_request_framing_is_rejected(service):
    """Where two parsers in a row (a load balancer, then this server) could disagree about where a request ends, the
    request is refused: Content-Length with Transfer-Encoding, repeated Content-Length, codings other than exactly
    'chunked', chunk sizes that int(x, 16) would accept but the grammar does not."""
    r, _, _ = service
    base = b"POST /extract HTTP/1.1\r\nhost: x\r\n"
    bad_heads = [
        base + b"content-length: 5\r\ntransfer-encoding: chunked\r\n\r\n",
        base + b"content-length: 5\r\ncontent-length: 5\r\n\r\nhello",
        base + b"content-length: 5\r\ncontent-length: 6\r\n\r\nhello",
        base + b"transfer-encoding: xchunked\r\n\r\n",
        base + b"transfer-encoding: gzip, chunked\r\n\r\n",
        base + b"transfer-encoding: chunked\r\ntransfer-encoding: chunked\r\n\r\n",
        base + b"content-length: 99999999999999999999\r\n\r\n",
    ]
    for raw in bad_heads:
        s = socket.create_connection(("127.0.0.1", r.port))
        s.sendall(raw)
        got = s.recv(4096)
        assert got.startswith(b"HTTP/1.1 400"), (raw[-60:], got[:40])
        s.close()
    for size in (b"-1", b"+5", b"0x10", b"1_0", b"", b"11111111111111111", b"5 5"):
        s = socket.create_connection(("127.0.0.1", r.port))
        s.sendall(base + b"content-type: multipart/form-data; boundary=x\r\ntransfer-encoding: chunked\r\n\r\n" + size + b"\r\nhello\r\n0\r\n\r\n")
        got = s.recv(4096)
        assert got.startswith(b"HTTP/1.1 400"), (size, got[:40])
        s.close()
    c = http.client.HTTPConnection("127.0.0.1", r.port, timeout=10)
    c.request("GET", "/health")
    assert c.getresponse().status == 200


def test_pipelined_bytes_behind_a_running_request_are_bounded():
    """A client that keeps writing while its first request is still being answered is stopped by back-pressure: the
    server holds a bounded number of its bytes, not everything it cares to send."""
    release = threading.Event()
    protos = []

    async def app(scope, receive, send):
        if scope["type"] != "http":
            return
        await receive()
        while not release.is_set():
            await asyncio.sleep(0.01)
        await send({"type": "http.response.start", "status": 200, "headers": [(b"content-length", b"2")]})
        await send({"type": "http.response.body", "body": b"ok"})

    with Running(app) as r:
        s = socket.create_connection(("127.0.0.1", r.port))
        s.sendall(b"GET /slow HTTP/1.1\r\nhost: x\r\n\r\n")
        s.setblocking(False)
        junk = b"GET /again HTTP/1.1\r\nhost: x\r\nx-pad: " + b"a" * 900 + b"\r\n\r\n"
        sent, t_end = 0, time.time() + 3.0
        while time.time() < t_end and sent < 64 << 20:
            try:
                sent += s.send(junk)
            except BlockingIOError:
                time.sleep(0.01)
        protos = list(r.server.connections)
        assert protos and all(p.hlen <= httpd.MAX_PIPELINED + 2 * httpd.MAX_HEADER for p in protos)
        assert sent < 32 << 20, "the socket must have filled up: the server stopped reading"
        release.set()
        s.setblocking(True)
        s.settimeout(10)
        assert s.recv(4096).startswith(b"HTTP/1.1 200")
        s.close()


def test_body_sink_receives_the_upload_in_place():
    """An application that names the body's destination gets memoryview slices OF that buffer from receive():
    nothing is copied between the socket and the buffer."""
    got = {}

    async def app(scope, receive, send):
        if scope["type"] != "http":
            return
        n = scope["extensions"]["dlamd.body_sink"]["length"]
        buf = bytearray(n)
        assert scope["extensions"]["dlamd.body_sink"]["set"](buf)
        pos = in_place = 0
        while True:
            m = await receive()
            b = m["body"]
            if len(b):
                if isinstance(b, memoryview) and b.obj is buf:
                    in_place += len(b)
                else:
                    buf[pos : pos + len(b)] = b           # bytes that shared a segment with the request head
                pos += len(b)
            if not m["more_body"]:
                break
        got.update(buf=bytes(buf), in_place=in_place, n=n)
        await send({"type": "http.response.start", "status": 200, "headers": [(b"content-length", b"0")]})
        await send({"type": "http.response.body", "body": b""})

    payload = bytes(range(251)) * 20000                     # ~5 MB
    with Running(app) as r:
        c = http.client.HTTPConnection("127.0.0.1", r.port, timeout=30)
        c.request("POST", "/up", body=payload)
        assert c.getresponse().status == 200
    assert got["buf"] == payload and got["in_place"] > 0.9 * len(payload)


def test_upload_received_into_shared_memory_reaches_the_extractor_unchanged(tmp_path):
    """With a worker-pool back end the upload is received into a shared-memory segment; the extractor gets a
    ShmDocument that points at the file part inside it, and the segment is reused for the next upload."""
    import os

    from distributed_llms_amd.runtime.workers import ShmDocument

    seen = []

    class ShmEcho(EchoExtractor):
        wants_shm = True

        def extract(self, pdf_bytes, document_id, document_hash=None):
            assert isinstance(pdf_bytes, ShmDocument)
            seen.append(pdf_bytes.seg.name)
            raw = bytes(pdf_bytes.view())
            return {"status": "completed", "result": f"{len(raw)} {hashlib.sha256(raw).hexdigest()}", "pages": 1}

    st, rs = MemoryStatusStore(), MemoryResultStore()
    app = create_app(Settings(store="memory", output_dir=str(tmp_path), num_gpus=0, node_id="t"), st, rs, ShmEcho())
    with Running(app) as r:
        c = http.client.HTTPConnection("127.0.0.1", r.port, timeout=30)
        for k in range(3):
            pdf = b"%PDF-1.4\n" + bytes([k]) * 700_000 + b"\r\n--not the boundary\r\n" + b"y" * 1000
            body, ct = encode_multipart({"document_id": f"s{k}"}, {"file": ("a.pdf", pdf, "application/pdf")})
            c.request("POST", "/extract", body=body, headers={"content-type": ct})
            resp = c.getresponse()
            assert resp.status == 200 and json.loads(resp.read()) == {"status": "ok"}
            assert _wait_done(c, f"s{k}") == "completed"
            assert rs.find(f"s{k}")["combined_text"] == f"{len(pdf)} {hashlib.sha256(pdf).hexdigest()}"
            assert st.get(f"s{k}")["document_hash"] == hashlib.sha256(pdf).hexdigest()
        assert len(set(seen)) == 1, "the segment of a finished upload serves the next one"
        assert os.path.exists("/dev/shm/" + seen[0])
    assert not os.path.exists("/dev/shm/" + seen[0]), "released at shutdown"
