"""Pure-Python RESP2 (Redis wire protocol) client and an in-process fake server.

The reference talks to Redis through redis-py (services/ocr_service/
ocr_service.py:55-60; requirements.txt:239).  redis-py is not a dependency
here: the handful of commands the service needs (SETEX, SET NX EX, GET,
EXISTS, DEL, TTL, PING, AUTH, SCAN) are spoken directly, wire-compatible with
a real Redis 7.  ``FakeRedisServer`` implements the same commands over a real
TCP socket so the client, the TTL behaviour and the status state machine can be
tested without a redis-server binary.
"""
from __future__ import annotations

import fnmatch
import socket
import socketserver
import threading
import time
from typing import Dict, Iterable, List, Optional, Tuple, Union

RespValue = Union[None, int, str, bytes, list]


class RespError(Exception):
    """Server replied with an error (``-ERR ...``)."""


class RespConnectionError(ConnectionError):
    pass


def encode_command(*args: Union[str, bytes, int, float]) -> bytes:
    out = [b"*%d\r\n" % len(args)]
    for a in args:
        if isinstance(a, bytes):
            b = a
        elif isinstance(a, str):
            b = a.encode("utf-8")
        else:
            b = str(a).encode("ascii")
        out.append(b"$%d\r\n" % len(b))
        out.append(b)
        out.append(b"\r\n")
    return b"".join(out)


class _Reader:
    def __init__(self, sock: socket.socket):
        self.sock = sock
        self.buf = bytearray()

    def _fill(self) -> None:
        chunk = self.sock.recv(65536)
        if not chunk:
            raise RespConnectionError("connection closed by peer")
        self.buf += chunk

    def readline(self) -> bytes:
        while True:
            i = self.buf.find(b"\r\n")
            if i >= 0:
                line = bytes(self.buf[:i])
                del self.buf[: i + 2]
                return line
            self._fill()

    def readexact(self, n: int) -> bytes:
        while len(self.buf) < n + 2:
            self._fill()
        data = bytes(self.buf[:n])
        del self.buf[: n + 2]
        return data

    def read_reply(self) -> RespValue:
        line = self.readline()
        if not line:
            raise RespConnectionError("empty reply")
        t, rest = line[:1], line[1:]
        if t == b"+":
            return rest.decode("utf-8")
        if t == b"-":
            raise RespError(rest.decode("utf-8", "replace"))
        if t == b":":
            return int(rest)
        if t == b"$":
            n = int(rest)
            if n < 0:
                return None
            return self.readexact(n)
        if t == b"*":
            n = int(rest)
            if n < 0:
                return None
            return [self.read_reply() for _ in range(n)]
        raise RespConnectionError(f"unknown RESP type byte {t!r}")


class RespClient:
    """Small thread-safe Redis client (one socket, lock per command, auto-reconnect once)."""

    def __init__(self, host: str = "localhost", port: int = 6379, password: Optional[str] = None,
                 timeout: float = 5.0, decode_responses: bool = True):
        self.host, self.port, self.password = host, int(port), password
        self.timeout = timeout
        self.decode_responses = decode_responses
        self._sock: Optional[socket.socket] = None
        self._reader: Optional[_Reader] = None
        self._lock = threading.Lock()

    # -- connection ----------------------------------------------------------
    def _connect(self) -> None:
        s = socket.create_connection((self.host, self.port), timeout=self.timeout)
        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self._sock, self._reader = s, _Reader(s)
        if self.password:
            s.sendall(encode_command("AUTH", self.password))
            self._reader.read_reply()

    def close(self) -> None:
        with self._lock:
            self._drop()

    def _drop(self) -> None:
        if self._sock is not None:
            try:
                self._sock.close()
            except OSError:
                pass
        self._sock = self._reader = None

    def execute(self, *args) -> RespValue:
        payload = encode_command(*args)
        with self._lock:
            for attempt in (0, 1):
                try:
                    if self._sock is None:
                        self._connect()
                    self._sock.sendall(payload)
                    return self._decode(self._reader.read_reply())
                except RespError:
                    raise
                except (OSError, RespConnectionError) as e:
                    self._drop()
                    if attempt == 1:
                        raise RespConnectionError(f"redis {self.host}:{self.port}: {e}") from e
        raise RespConnectionError("unreachable")

    def _decode(self, v: RespValue) -> RespValue:
        if not self.decode_responses:
            return v
        if isinstance(v, bytes):
            return v.decode("utf-8", "replace")
        if isinstance(v, list):
            return [self._decode(x) for x in v]
        return v

    # -- commands (redis-py compatible names) -----------------------------------
    def ping(self) -> bool:
        return self.execute("PING") == "PONG"

    def get(self, key: str) -> Optional[str]:
        return self.execute("GET", key)

    def set(self, key: str, value, ex: Optional[int] = None, nx: bool = False) -> bool:
        args: List = ["SET", key, value]
        if ex is not None:
            args += ["EX", int(ex)]
        if nx:
            args.append("NX")
        return self.execute(*args) == "OK"

    def setex(self, key: str, seconds: int, value) -> bool:
        return self.execute("SETEX", key, int(seconds), value) == "OK"

    def exists(self, *keys: str) -> int:
        return int(self.execute("EXISTS", *keys))

    def delete(self, *keys: str) -> int:
        return int(self.execute("DEL", *keys))

    def ttl(self, key: str) -> int:
        return int(self.execute("TTL", key))

    def expire(self, key: str, seconds: int) -> bool:
        return int(self.execute("EXPIRE", key, int(seconds))) == 1

    def scan_iter(self, match: str = "*", count: int = 500) -> Iterable[str]:
        cursor = "0"
        while True:
            reply = self.execute("SCAN", cursor, "MATCH", match, "COUNT", count)
            cursor, keys = reply[0], reply[1]
            for k in keys:
                yield k
            if str(cursor) == "0":
                break

    def flushall(self) -> bool:
        return self.execute("FLUSHALL") == "OK"


# ---------------------------------------------------------------------------
# In-process fake server (tests / dev without a redis binary)
# ---------------------------------------------------------------------------
class _FakeDB:
    def __init__(self, clock=time.monotonic):
        self.data: Dict[bytes, Tuple[bytes, Optional[float]]] = {}
        self.lock = threading.Lock()
        self.clock = clock

    def _live(self, k: bytes) -> Optional[bytes]:
        ent = self.data.get(k)
        if ent is None:
            return None
        v, exp = ent
        if exp is not None and self.clock() >= exp:
            del self.data[k]
            return None
        return v

    def handle(self, argv: List[bytes], authed: List[bool], password: Optional[str]):
        cmd = argv[0].upper().decode("ascii", "replace")
        if cmd == "AUTH":
            if password is None:
                return RespError("ERR Client sent AUTH, but no password is set")
            if argv[-1].decode() == password:
                authed[0] = True
                return "OK"
            return RespError("WRONGPASS invalid username-password pair or user is disabled.")
        if password is not None and not authed[0]:
            return RespError("NOAUTH Authentication required.")
        with self.lock:
            if cmd == "PING":
                return "PONG" if len(argv) == 1 else argv[1]
            if cmd == "GET":
                return self._live(argv[1])
            if cmd == "SETEX":
                self.data[argv[1]] = (argv[3], self.clock() + int(argv[2]))
                return "OK"
            if cmd == "SET":
                ex, nx, xx = None, False, False
                i = 3
                while i < len(argv):
                    o = argv[i].upper()
                    if o == b"EX":
                        ex = int(argv[i + 1]); i += 2
                    elif o == b"PX":
                        ex = int(argv[i + 1]) / 1000.0; i += 2
                    elif o == b"NX":
                        nx = True; i += 1
                    elif o == b"XX":
                        xx = True; i += 1
                    else:
                        return RespError("ERR syntax error")
                cur = self._live(argv[1])
                if (nx and cur is not None) or (xx and cur is None):
                    return None
                self.data[argv[1]] = (argv[2], None if ex is None else self.clock() + ex)
                return "OK"
            if cmd == "EXISTS":
                return sum(1 for k in argv[1:] if self._live(k) is not None)
            if cmd == "DEL":
                n = 0
                for k in argv[1:]:
                    if self._live(k) is not None:
                        del self.data[k]
                        n += 1
                return n
            if cmd == "TTL":
                if self._live(argv[1]) is None:
                    return -2
                exp = self.data[argv[1]][1]
                return -1 if exp is None else max(0, int(round(exp - self.clock())))
            if cmd == "EXPIRE":
                v = self._live(argv[1])
                if v is None:
                    return 0
                self.data[argv[1]] = (v, self.clock() + int(argv[2]))
                return 1
            if cmd in ("SCAN", "KEYS"):
                pattern = "*"
                if cmd == "KEYS":
                    pattern = argv[1].decode()
                else:
                    for i in range(2, len(argv) - 1):
                        if argv[i].upper() == b"MATCH":
                            pattern = argv[i + 1].decode()
                keys = [k for k in list(self.data) if self._live(k) is not None
                        and fnmatch.fnmatchcase(k.decode("utf-8", "replace"), pattern)]
                return keys if cmd == "KEYS" else [b"0", keys]
            if cmd == "FLUSHALL" or cmd == "FLUSHDB":
                self.data.clear()
                return "OK"
            if cmd == "DBSIZE":
                return sum(1 for k in list(self.data) if self._live(k) is not None)
            if cmd in ("CLIENT", "SELECT", "HELLO"):
                return "OK" if cmd != "HELLO" else RespError("ERR unknown command 'HELLO'")
        return RespError(f"ERR unknown command '{cmd}'")


def _encode_reply(v) -> bytes:
    if isinstance(v, RespError):
        return b"-" + str(v).encode() + b"\r\n"
    if v is None:
        return b"$-1\r\n"
    if isinstance(v, bool):
        return b":%d\r\n" % int(v)
    if isinstance(v, int):
        return b":%d\r\n" % v
    if isinstance(v, str):
        return b"+" + v.encode() + b"\r\n"
    if isinstance(v, bytes):
        return b"$%d\r\n" % len(v) + v + b"\r\n"
    if isinstance(v, list):
        return b"*%d\r\n" % len(v) + b"".join(_encode_reply(x) for x in v)
    raise TypeError(type(v))


class FakeRedisServer:
    """RESP2 server on 127.0.0.1:<ephemeral port>; use as a context manager."""

    def __init__(self, password: Optional[str] = None, clock=time.monotonic, port: int = 0):
        self.db = _FakeDB(clock)
        self.password = password
        outer = self

        class Handler(socketserver.BaseRequestHandler):
            def handle(self):
                reader = _Reader(self.request)
                authed = [False]
                try:
                    while True:
                        req = reader.read_reply()
                        if not isinstance(req, list) or not req:
                            self.request.sendall(b"-ERR protocol error\r\n")
                            continue
                        rep = outer.db.handle([bytes(x) for x in req], authed, outer.password)
                        self.request.sendall(_encode_reply(rep))
                except (RespConnectionError, OSError):
                    return

        class Server(socketserver.ThreadingTCPServer):
            allow_reuse_address = True
            daemon_threads = True

        self._server = Server(("127.0.0.1", port), Handler)
        self.host, self.port = self._server.server_address
        self._thread = threading.Thread(target=self._server.serve_forever, kwargs={"poll_interval": 0.05}, daemon=True)

    def start(self) -> "FakeRedisServer":
        self._thread.start()
        return self

    def stop(self) -> None:
        self._server.shutdown()
        self._server.server_close()

    def __enter__(self) -> "FakeRedisServer":
        return self.start()

    def __exit__(self, *exc) -> None:
        self.stop()
