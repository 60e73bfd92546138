"""HTTP/1.1 front end for the ASGI application, built for this service's two traffic shapes.

The reference runs under uvicorn (services/ocr_service/Dockerfile:28).  With the pure-Python ``h11`` parser (the only
one installable here) that costs ~0.5 ms of interpreter time per request and ~50 us per 64 KB body event: a client
polling ``/check-status`` every 10 ms for each of 32 documents in flight keeps one core busy with parsing alone, and a
124 MB scanned PDF spends 0.1 s inside the HTTP layer before the first page can be decoded - on a GPU that reads the
100 pages in 0.13 s.  This server speaks the same ASGI 3 interface to the same application and does less:

* ``asyncio.BufferedProtocol``: the kernel copies request bodies straight into 1 MiB chunk buffers that are handed
  to the application as they fill (no intermediate ``bytes`` objects, no per-64-KB event objects);
* request line and headers are split with ``bytes`` methods (one pass, no state machine objects);
* keep-alive, ``Content-Length`` and chunked request bodies, ``Expect: 100-continue``, chunked responses when the
  application streams without a length, pipelined requests served in order, read back-pressure, idle timeout;
* the ASGI lifespan protocol, and a listening socket passed in as a file descriptor (``api/serve.py`` binds one
  port for several front-end processes);
* ``scope["extensions"]["dlamd.body_sink"]``: an application that knows where the body has to end up (``POST
  /extract`` receives a scan into a shared-memory segment the GPU workers map) hands that buffer over with
  ``sink["set"](buffer)``; from then on the kernel copies the socket's bytes straight to ``buffer[body offset:]`` and
  ``receive()`` delivers memoryview slices OF that buffer - the upload is written exactly once on its way from the
  network card to the PDF parser;
* request framing is strict where leniency is dangerous behind a proxy: ``Content-Length`` together with
  ``Transfer-Encoding``, repeated or differing ``Content-Length``, a transfer coding other than exactly ``chunked``
  and chunk sizes that are not plain hex digits are all answered 400, and bytes pipelined behind a request that is
  still being served are bounded (431) instead of buffered without limit.

Not supported (answered with an error or by closing): HTTP/2, WebSocket upgrades, TLS (terminate it in front, as the
reference's load balancer does, terraform/main.tf:176-230).  ``python -m distributed_llms_amd.api.serve --server
uvicorn`` keeps the stock server available.
"""
from __future__ import annotations

import asyncio
import logging
import socket
import time
from email.utils import formatdate
from typing import Dict, List, Optional, Tuple
from urllib.parse import unquote_to_bytes

logger = logging.getLogger("ocr.httpd")

CHUNK = 1 << 20                 # body chunk buffer
MAX_HEADER = 64 << 10
MAX_QUEUED = 8                  # body chunks waiting for the application before reading pauses
KEEPALIVE_S = 5.0
MAX_PIPELINED = MAX_HEADER      # bytes that may wait in the header buffer while a response is still being produced
_HEX = frozenset(b"0123456789abcdefABCDEF")
_REASONS = {200: b"OK", 201: b"Created", 204: b"No Content", 304: b"Not Modified", 400: b"Bad Request",
            404: b"Not Found", 405: b"Method Not Allowed", 413: b"Payload Too Large", 422: b"Unprocessable Entity",
            431: b"Request Header Fields Too Large", 500: b"Internal Server Error", 503: b"Service Unavailable",
            307: b"Temporary Redirect", 401: b"Unauthorized", 403: b"Forbidden", 100: b"Continue"}
_date_cache: List = [0, b""]


def _date() -> bytes:
    now = int(time.time())
    if _date_cache[0] != now:
        _date_cache[0], _date_cache[1] = now, formatdate(now, usegmt=True).encode()
    return _date_cache[1]


class _BadRequest(Exception):
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


def parse_head(head: bytes) -> Tuple[bytes, bytes, bytes, List[Tuple[bytes, bytes]]]:
    """Request line + header lines (without the final blank line) -> (method, target, version, headers)."""
    lines = head.split(b"\r\n")
    parts = lines[0].split(b" ")
    if len(parts) != 3 or not parts[2].startswith(b"HTTP/1."):
        raise _BadRequest(400, "malformed request line")
    method, target, version = parts
    if not method.isalpha() or not target:
        raise _BadRequest(400, "malformed request line")
    headers: List[Tuple[bytes, bytes]] = []
    for ln in lines[1:]:
        k, sep, v = ln.partition(b":")
        if not sep or not k or k != k.strip() or b" " in k:
            raise _BadRequest(400, "malformed header line")
        headers.append((k.lower(), v.strip(b" \t")))
    return method, target, version, headers


class _Request:
    __slots__ = ("scope", "remaining", "chunked", "expect", "keep_alive", "chunk_state", "body_done", "body_pos", "sink")


class HttpProtocol(asyncio.BufferedProtocol):
    def __init__(self, app, server: "Server"):
        self.app, self.server = app, server
        self.loop = asyncio.get_running_loop()
        self.transport: Optional[asyncio.Transport] = None
        self.hbuf = bytearray(MAX_HEADER + 4096)       # header / chunked-framing bytes
        self.hlen = 0
        self.cur: Optional[bytearray] = None           # body chunk being filled
        self.cur_len = 0
        self.req: Optional[_Request] = None
        self.queue: List = []                          # body pieces (memoryview / bytes) ready for receive()
        self.waiter: Optional[asyncio.Future] = None   # receive() blocked on the queue
        self.task: Optional[asyncio.Task] = None
        self.paused = False
        self.closed = False
        self.idle: Optional[asyncio.TimerHandle] = None
        self.client = self.sockname = None
        self.write_ok = asyncio.Event()
        self.write_ok.set()
        self._ended = False
        self._head: Optional[bytes] = None
        self._sink_head = self._sink_tail = 0          # body offsets of the last in-place piece queued for receive()

    # ------------------------------------------------------------------ transport callbacks
    def connection_made(self, transport):
        self.transport = transport
        self.client = transport.get_extra_info("peername")
        self.sockname = transport.get_extra_info("sockname")
        sock = transport.get_extra_info("socket")
        if sock is not None:
            try:
                sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            except OSError:
                pass
        self.server.connections.add(self)
        self._arm_idle()

    def connection_lost(self, exc):
        self.closed = True
        self.server.connections.discard(self)
        if self.idle:
            self.idle.cancel()
        self.write_ok.set()
        if self.waiter and not self.waiter.done():
            self.waiter.set_result(None)
        if self.task and not self.task.done() and self.req is not None and not self.req.body_done:
            pass          # the application sees http.disconnect from receive()

    def pause_writing(self):
        self.write_ok.clear()

    def resume_writing(self):
        self.write_ok.set()

    def get_buffer(self, sizehint):
        r = self.req
        if r is not None and not r.chunked and r.remaining > 0 and self.hlen == 0:
            if r.sink is not None and self.cur is None:
                # the application named the body's destination: the kernel writes there
                return r.sink[r.body_pos : r.body_pos + r.remaining]
            # fixed-length body: the kernel writes into the chunk that the application will receive
            if self.cur is None:
                self.cur = bytearray(min(CHUNK, r.remaining))
                self.cur_len = 0
            return memoryview(self.cur)[self.cur_len:]
        if self.hlen >= min(len(self.hbuf), MAX_PIPELINED + 1):
            if (r is None or r.body_done) and self.task is not None and not self.task.done() and self.hlen > MAX_PIPELINED:
                # a client that keeps sending while its previous request is still being answered: stop reading until
                # the response is out (_run resumes), never grow the buffer without bound
                if not self.paused:
                    self.paused = True
                    self.transport.pause_reading()
            if self.hlen >= len(self.hbuf):
                self.hbuf.extend(bytes(4096))
        return memoryview(self.hbuf)[self.hlen:]

    def buffer_updated(self, n):
        r = self.req
        if r is not None and not r.chunked and r.remaining > 0 and self.hlen == 0:
            if r.sink is not None and self.cur is None:
                piece = r.sink[r.body_pos : r.body_pos + n]
                r.body_pos += n
                r.remaining -= n
                q = self.queue
                if q and isinstance(q[-1], memoryview) and q[-1].obj is r.sink.obj and self._sink_tail == r.body_pos - n:
                    # the application has not taken the previous in-place piece yet: one longer piece, not two
                    a = self._sink_head
                    q[-1] = r.sink[a : r.body_pos]
                    self._sink_tail = r.body_pos
                    if self.waiter is not None and not self.waiter.done():
                        self.waiter.set_result(None)
                else:
                    self._sink_head, self._sink_tail = r.body_pos - n, r.body_pos
                    self._push(piece)
                if r.remaining == 0:
                    r.body_done = True
                    self._push(None)
                return
            self.cur_len += n
            r.remaining -= n
            r.body_pos += n
            if self.cur_len == len(self.cur) or r.remaining == 0:
                self._push(memoryview(self.cur)[: self.cur_len])
                self.cur = None
                if r.remaining == 0:
                    r.body_done = True
                    self._push(None)
            elif self.waiter is not None:
                # the application is waiting: hand over what has arrived instead of filling the chunk first
                self._push(memoryview(self.cur)[: self.cur_len])
                self.cur = None
            return
        self.hlen += n
        self._consume_hbuf()

    def eof_received(self):
        return False

    # ------------------------------------------------------------------ parsing
    def _consume_hbuf(self):
        """Bytes in hbuf[:hlen] are, depending on the state: the head of a new request, body bytes that arrived with
        the head, chunked-body framing, or pipelined requests waiting for the current one to finish."""
        while self.hlen and not self.closed:
            r = self.req
            if r is None:
                if self.task is not None and not self.task.done():
                    if self.hlen > MAX_PIPELINED + MAX_HEADER:
                        self._reject(431, "too many pipelined bytes")
                    return                      # previous response still being produced: leave the bytes queued
                end = self.hbuf.find(b"\r\n\r\n", 0, self.hlen)
                if end < 0:
                    if self.hlen > MAX_HEADER:
                        self._reject(431, "header section too large")
                    return
                head = bytes(self.hbuf[:end])
                self._drop(end + 4)
                try:
                    self._start_request(head)
                except _BadRequest as e:
                    self._reject(e.code, str(e))
                    return
                continue
            if r.body_done:
                return
            if r.chunked:
                if not self._decode_chunked(r):
                    return
                continue
            # fixed length: body bytes that shared a segment with the head
            take = min(self.hlen, r.remaining)
            if take:
                self._push(bytes(self.hbuf[:take]))
                self._drop(take)
                r.remaining -= take
                r.body_pos += take
            if r.remaining == 0:
                r.body_done = True
                self._push(None)
            return

    def _drop(self, n: int):
        if n >= self.hlen:
            self.hlen = 0
        else:
            self.hbuf[: self.hlen - n] = self.hbuf[n : self.hlen]
            self.hlen -= n

    def _decode_chunked(self, r: _Request) -> bool:
        """One step of chunked decoding over hbuf; False when more bytes are needed."""
        st = r.chunk_state
        if st[0] == "size":
            eol = self.hbuf.find(b"\r\n", 0, self.hlen)
            if eol < 0:
                if self.hlen > 4096:
                    self._reject(400, "chunk size line too long")
                return False
            digits = bytes(self.hbuf[:eol]).split(b";")[0].strip(b" \t")
            # plain hex only: int(x, 16) would also take '-1', '+5', '0x10' and '1_0'
            if not 1 <= len(digits) <= 16 or not _HEX.issuperset(digits):
                self._reject(400, "bad chunk size")
                return False
            size = int(digits, 16)
            self._drop(eol + 2)
            st[0], st[1] = ("data", size) if size else ("trailer", 0)
            return True
        if st[0] == "data":
            take = min(self.hlen, st[1])
            if take:
                self._push(bytes(self.hbuf[:take]))
                self._drop(take)
                st[1] -= take
            if st[1] == 0:
                st[0] = "crlf"
                return True
            return False
        if st[0] == "crlf":
            if self.hlen < 2:
                return False
            if bytes(self.hbuf[:2]) != b"\r\n":
                self._reject(400, "bad chunk terminator")
                return False
            self._drop(2)
            st[0] = "size"
            return True
        # trailer section: lines until an empty one
        eol = self.hbuf.find(b"\r\n", 0, self.hlen)
        if eol < 0:
            return False
        self._drop(eol + 2)
        if eol == 0:
            r.body_done = True
            self._push(None)
        return True

    def _start_request(self, head: bytes):
        method, target, version, headers = parse_head(head)
        path, _, query = target.partition(b"?")
        clen = None
        chunked = False
        expect = False
        conn = b""
        te = None
        for k, v in headers:
            if k == b"content-length":
                if not v.isdigit() or len(v) > 18:
                    raise _BadRequest(400, "bad content-length")
                if clen is not None:
                    # two Content-Length fields (equal or not): a proxy in front may have framed by the other one
                    raise _BadRequest(400, "repeated content-length")
                clen = int(v)
            elif k == b"transfer-encoding":
                if te is not None:
                    raise _BadRequest(400, "repeated transfer-encoding")
                te = v.lower()
            elif k == b"expect":
                expect = v.lower() == b"100-continue"
            elif k == b"connection":
                conn = v.lower()
            elif k == b"upgrade" and v.lower() == b"websocket":
                raise _BadRequest(400, "websocket upgrades are not served")
        if te is not None:
            if te != b"chunked":
                # 'gzip, chunked' needs a decoder this server does not have; 'xchunked' or 'chunked, identity' must not
                # be read as chunked framing
                raise _BadRequest(400, "unsupported transfer-encoding")
            if clen is not None:
                raise _BadRequest(400, "content-length together with transfer-encoding")
            chunked = True
        r = _Request()
        r.keep_alive = (version == b"HTTP/1.1" and b"close" not in conn) or (version == b"HTTP/1.0" and b"keep-alive" in conn)
        r.chunked = chunked
        r.remaining = 0 if chunked else (clen or 0)
        r.expect = expect
        r.chunk_state = ["size", 0]
        r.body_done = not chunked and r.remaining == 0
        r.body_pos = 0
        r.sink = None
        ext = {}
        if not chunked and r.remaining > 0:
            ext["dlamd.body_sink"] = {"set": lambda buf, r=r: self._set_sink(r, buf), "length": r.remaining}
        r.scope = {
            "type": "http", "asgi": {"version": "3.0", "spec_version": "2.3"}, "http_version": version[5:].decode(),
            "method": method.decode().upper(), "scheme": "http", "path": unquote_to_bytes(path).decode("utf-8", "replace"),
            "raw_path": path, "query_string": query, "root_path": "", "headers": headers,
            "client": tuple(self.client[:2]) if self.client else None,
            "server": tuple(self.sockname[:2]) if self.sockname else None, "state": self.server.state.copy(),
            "extensions": ext,
        }
        self.req = r
        self.queue = []
        self._ended = False
        if r.body_done:
            self.queue.append(None)
        if self.idle:
            self.idle.cancel()
            self.idle = None
        self.task = self.loop.create_task(self._run(r))

    def _set_sink(self, r: _Request, buf) -> bool:
        """The application's destination for the rest of this request's body: ``buf[k]`` is body byte ``k``.  Returns
        False when it cannot be used (wrong request, too small, not writable)."""
        if r is not self.req or r.chunked or r.sink is not None:
            return False
        mv = memoryview(buf)
        if mv.readonly or mv.ndim != 1 or mv.itemsize != 1 or len(mv) < r.body_pos + r.remaining:
            return False
        if self.cur is not None:
            # a chunk is half filled: hand over what it holds, the sink takes over from the next byte
            if self.cur_len:
                self._push(memoryview(self.cur)[: self.cur_len])
            self.cur = None
        r.sink = mv
        return True

    def _reject(self, code: int, msg: str):
        body = msg.encode()
        self.transport.write(b"HTTP/1.1 %d %s\r\ncontent-type: text/plain\r\ncontent-length: %d\r\nconnection: close\r\n\r\n"
                             % (code, _REASONS.get(code, b"Error"), len(body)) + body)
        self.transport.close()
        self.closed = True

    # ------------------------------------------------------------------ body hand-over
    def _push(self, piece):
        self.queue.append(piece)
        if self.waiter is not None and not self.waiter.done():
            self.waiter.set_result(None)
        if len(self.queue) > MAX_QUEUED and not self.paused:
            self.paused = True
            self.transport.pause_reading()

    async def _receive(self):
        """ASGI receive(): body pieces in order, then one message with more_body False; after that it blocks until the
        client goes away (a response that is still streaming listens here for the disconnect)."""
        r = self.req
        if r is not None and r.expect:
            r.expect = False
            self.transport.write(b"HTTP/1.1 100 Continue\r\n\r\n")
        while True:
            if self.queue:
                piece = self.queue[0]
                if piece is None:
                    if not self._ended:
                        self._ended = True
                        return {"type": "http.request", "body": b"", "more_body": False}
                else:
                    self.queue.pop(0)
                    more = not (self.queue and self.queue[0] is None)
                    if not more:
                        self._ended = True
                    if self.paused and len(self.queue) <= MAX_QUEUED // 2:
                        self.paused = False
                        self.transport.resume_reading()
                    return {"type": "http.request", "body": piece, "more_body": more}
            if self.closed:
                return {"type": "http.disconnect"}
            self.waiter = self.loop.create_future()
            try:
                await self.waiter
            finally:
                self.waiter = None

    # ------------------------------------------------------------------ one request
    async def _run(self, r: _Request):
        started = False
        done = False
        chunked_out = False
        keep = r.keep_alive
        head_only = r.scope["method"] == "HEAD"
        receive = self._receive

        async def send(msg):
            nonlocal started, done, chunked_out, keep
            if self.closed:
                return
            t = msg["type"]
            if t == "http.response.start":
                code = msg["status"]
                out = [b"HTTP/1.1 %d %s\r\n" % (code, _REASONS.get(code, b"Status"))]
                has_len = False
                for k, v in msg.get("headers", ()):
                    kl = k.lower()
                    if kl == b"content-length":
                        has_len = True
                    elif kl == b"connection" and b"close" in v.lower():
                        keep = False
                    out.append(k + b": " + v + b"\r\n")
                if not has_len and not head_only and code >= 200 and code not in (204, 304):
                    chunked_out = True
                    out.append(b"transfer-encoding: chunked\r\n")
                out.append(b"date: " + _date() + b"\r\nserver: dlamd\r\n")
                if not keep:
                    out.append(b"connection: close\r\n")
                out.append(b"\r\n")
                self._head = b"".join(out)
                started = True
            elif t == "http.response.body":
                body = msg.get("body", b"")
                more = msg.get("more_body", False)
                parts = []
                if self._head is not None:
                    parts.append(self._head)
                    self._head = None
                if head_only:
                    body = b""
                if chunked_out:
                    if body:
                        parts += [b"%x\r\n" % len(body), body, b"\r\n"]
                    if not more:
                        parts.append(b"0\r\n\r\n")
                elif body:
                    parts.append(body)
                if parts:
                    self.transport.write(b"".join(parts) if len(parts) > 1 else parts[0])
                if not self.write_ok.is_set():
                    await self.write_ok.wait()
                if not more:
                    done = True

        self._head = None
        try:
            await self.app(r.scope, receive, send)
        except asyncio.CancelledError:
            raise
        except Exception:  # noqa: BLE001
            logger.exception("application error")
            if not started and not self.closed:
                self.transport.write(b"HTTP/1.1 500 Internal Server Error\r\ncontent-type: text/plain\r\n"
                                     b"content-length: 21\r\nconnection: close\r\n\r\nInternal Server Error")
            keep = False
            done = True
        if self.closed:
            return
        if not done or not r.body_done or not keep or self.server.stopping:
            # unread request body / unfinished response: the connection cannot be reused
            self.transport.close()
            self.closed = True
            return
        self.req = None
        self.queue = []
        self.cur = None
        self._arm_idle()
        if self.paused:
            self.paused = False
            self.transport.resume_reading()
        if self.hlen:
            self.loop.call_soon(self._consume_hbuf)      # a pipelined request is already here

    def _arm_idle(self):
        if self.idle:
            self.idle.cancel()
        self.idle = self.loop.call_later(KEEPALIVE_S, self._idle_close)

    def _idle_close(self):
        if self.req is None and not self.closed and (self.task is None or self.task.done()):
            self.transport.close()
            self.closed = True


class Server:
    """``await Server(app).serve(host, port)`` or ``run(app, ...)``: lifespan, listen, graceful stop."""

    def __init__(self, app):
        self.app = app
        self.connections: set = set()
        self.state: Dict = {}
        self.stopping = False
        self.should_exit = False
        self._lifespan_q: Optional[asyncio.Queue] = None
        self._lifespan_task: Optional[asyncio.Task] = None
        self._startup_done: Optional[asyncio.Future] = None
        self._shutdown_done: Optional[asyncio.Future] = None
        self.port: Optional[int] = None

    # ------------------------------------------------------------------ lifespan
    async def _lifespan(self):
        loop = asyncio.get_running_loop()
        self._lifespan_q = asyncio.Queue()
        self._startup_done, self._shutdown_done = loop.create_future(), loop.create_future()

        async def receive():
            return await self._lifespan_q.get()

        async def send(msg):
            t = msg["type"]
            if t == "lifespan.startup.complete":
                self._startup_done.set_result(True)
            elif t == "lifespan.startup.failed":
                self._startup_done.set_result(False)
                logger.error("application startup failed: %s", msg.get("message", ""))
            elif t in ("lifespan.shutdown.complete", "lifespan.shutdown.failed"):
                self._shutdown_done.set_result(True)

        try:
            await self.app({"type": "lifespan", "asgi": {"version": "3.0", "spec_version": "2.0"}, "state": self.state},
                           receive, send)
        except Exception:  # noqa: BLE001 - an application without lifespan support
            logger.debug("lifespan not supported by the application", exc_info=True)
        finally:
            for f in (self._startup_done, self._shutdown_done):
                if not f.done():
                    f.set_result(True)

    async def serve(self, host: str = "127.0.0.1", port: int = 8000, fd: Optional[int] = None,
                    ready: Optional[asyncio.Event] = None) -> None:
        loop = asyncio.get_running_loop()
        self._lifespan_task = loop.create_task(self._lifespan())
        await asyncio.sleep(0)
        await self._lifespan_q.put({"type": "lifespan.startup"})
        if not await self._startup_done:
            return
        kw = {}
        if fd is not None:
            sock = socket.socket(fileno=fd)
            sock.setblocking(False)
            kw["sock"] = sock
        else:
            kw.update(host=host, port=port, reuse_address=True)
        srv = await loop.create_server(lambda: HttpProtocol(self.app, self), backlog=2048, **kw)
        self.port = srv.sockets[0].getsockname()[1]
        if ready is not None:
            ready.set()
        try:
            while not self.should_exit:
                await asyncio.sleep(0.1)
        finally:
            self.stopping = True
            srv.close()
            await srv.wait_closed()
            t_end = time.monotonic() + 10.0
            while time.monotonic() < t_end and any(c.task is not None and not c.task.done() for c in self.connections):
                await asyncio.sleep(0.05)
            for c in list(self.connections):
                if c.transport is not None:
                    c.transport.close()
            await self._lifespan_q.put({"type": "lifespan.shutdown"})
            try:
                await asyncio.wait_for(self._shutdown_done, 30.0)
            except asyncio.TimeoutError:
                logger.error("application shutdown did not finish")


def run(app, host: str = "127.0.0.1", port: int = 8000, fd: Optional[int] = None) -> None:
    """Blocking entry point; SIGINT / SIGTERM stop it gracefully."""
    import signal

    server = Server(app)

    async def main():
        loop = asyncio.get_running_loop()
        for sig in (signal.SIGINT, signal.SIGTERM):
            try:
                loop.add_signal_handler(sig, lambda: setattr(server, "should_exit", True))
            except (NotImplementedError, RuntimeError):
                pass
        await server.serve(host, port, fd)

    asyncio.run(main())


def _load(spec: str):
    import importlib

    mod, _, attr = spec.partition(":")
    return getattr(importlib.import_module(mod), attr or "app")


def main(argv=None) -> int:
    """``python -m distributed_llms_amd.api.httpd distributed_llms_amd.api.asgi:app --port 8000``"""
    import argparse

    ap = argparse.ArgumentParser(description="HTTP/1.1 server for an ASGI application")
    ap.add_argument("app", nargs="?", default="distributed_llms_amd.api.asgi:app")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--fd", type=int, default=None, help="serve on this already-listening socket")
    args = ap.parse_args(argv)
    run(_load(args.app), args.host, args.port, args.fd)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
