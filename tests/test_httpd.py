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
