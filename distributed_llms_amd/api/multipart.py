"""multipart/form-data parser (RFC 7578) for ``POST /extract``.

FastAPI's ``File``/``Form`` need the ``python-multipart`` package (the
reference depends on it, services/ocr_service/requirements.in); it is not a
dependency here.  Two entry points:

* ``parse_multipart(body, content_type)`` - single pass over a complete body;
* ``StreamingForm`` - fed with the request's chunks as they arrive: the upload cap is enforced while streaming
  (a chunked upload without Content-Length cannot buffer without limit), the body lands in ONE preallocated buffer,
  the SHA-256 of the ``file`` part is computed on a helper thread while the rest of the body is still on the wire,
  and the file's bytes are handed on as a zero-copy view of that buffer (the reference reads the whole upload, then
  hashes it: ocr_service.py:437-438).
"""
from __future__ import annotations

import collections
import re
import threading
from dataclasses import dataclass, field
from typing import Dict, List, Optional
from urllib.parse import unquote


class MultipartError(ValueError):
    pass


_memmem = None


def buffer_find(buf, needle: bytes, start: int = 0, end: Optional[int] = None) -> int:
    """``buf.find(needle, start, end)`` at library speed for any buffer.  ``bytes`` / ``bytearray`` have a fast search
    of their own; ``mmap.find`` (the shared-memory receive buffer) walks byte by byte at < 1 GB/s - 0.14 s on a 120 MB
    upload - so those go through libc's ``memmem``."""
    global _memmem
    if isinstance(buf, (bytes, bytearray)):
        return buf.find(needle, start, len(buf) if end is None else end)
    import ctypes

    n = len(buf) if end is None else min(end, len(buf))
    if start < 0 or n - start < len(needle):
        return -1
    if _memmem is None:
        f = ctypes.CDLL(None).memmem
        f.restype = ctypes.c_void_p
        f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        _memmem = f
    anchor = ctypes.c_char.from_buffer(buf)          # pins the buffer only for the duration of the call
    try:
        base = ctypes.addressof(anchor)
        hit = _memmem(base + start, n - start, needle, len(needle))
        return -1 if not hit else hit - base
    finally:
        del anchor


class UploadTooLarge(MultipartError):
    pass


@dataclass
class Part:
    name: str
    data: bytes
    filename: Optional[str] = None
    content_type: Optional[str] = None
    headers: Dict[str, str] = field(default_factory=dict)
    offset: int = -1          # where ``data`` starts in the request body (streamed bodies; -1: unknown)

    @property
    def is_file(self) -> bool:
        return self.filename is not None

    def text(self, encoding: str = "utf-8") -> str:
        return bytes(self.data).decode(encoding, "replace")


_PARAM_RE = re.compile(r';\s*([A-Za-z0-9_\-\*]+)\s*=\s*(?:"((?:[^"\\]|\\.)*)"|([^;\s]*))')


def parse_header_params(value: str) -> (str, Dict[str, str]):
    """``form-data; name="file"; filename="a.pdf"`` -> ("form-data", {...})."""
    main = value.split(";", 1)[0].strip().lower()
    params: Dict[str, str] = {}
    for m in _PARAM_RE.finditer(value):
        key = m.group(1).lower()
        val = m.group(2) if m.group(2) is not None else (m.group(3) or "")
        if m.group(2) is not None:
            val = re.sub(r"\\(.)", r"\1", val)
        if key.endswith("*"):
            # RFC 5987: charset'lang'percent-encoded
            key = key[:-1]
            try:
                charset, _, enc = val.split("'", 2)
                val = unquote(enc, encoding=charset or "utf-8")
            except ValueError:
                pass
        params[key] = val
    return main, params


def boundary_from_content_type(content_type: Optional[str]) -> bytes:
    if not content_type:
        raise MultipartError("missing Content-Type header")
    main, params = parse_header_params(content_type)
    if main != "multipart/form-data":
        raise MultipartError(f"expected multipart/form-data, got {main!r}")
    b = params.get("boundary")
    if not b:
        raise MultipartError("multipart boundary missing")
    if len(b) > 70:
        raise MultipartError("multipart boundary too long")
    return b.encode("latin-1")


def parse_multipart(body: bytes, content_type: Optional[str], max_parts: int = 64) -> List[Part]:
    boundary = boundary_from_content_type(content_type)
    delim = b"--" + boundary
    pos = body.find(delim)
    if pos < 0:
        raise MultipartError("opening boundary not found")
    parts: List[Part] = []
    n = len(body)
    while True:
        pos += len(delim)
        if body[pos:pos + 2] == b"--":
            break  # closing delimiter
        # skip transport padding + CRLF
        eol = body.find(b"\r\n", pos)
        if eol < 0:
            raise MultipartError("malformed boundary line")
        pos = eol + 2
        hdr_end = body.find(b"\r\n\r\n", pos)
        if hdr_end < 0:
            raise MultipartError("part headers not terminated")
        headers: Dict[str, str] = {}
        for line in body[pos:hdr_end].split(b"\r\n"):
            if not line:
                continue
            if b":" not in line:
                raise MultipartError("malformed part header")
            k, v = line.split(b":", 1)
            headers[k.decode("latin-1").strip().lower()] = v.decode("utf-8", "replace").strip()
        data_start = hdr_end + 4
        nxt = body.find(b"\r\n" + delim, data_start)
        if nxt < 0:
            raise MultipartError("closing boundary not found")
        data = body[data_start:nxt]
        disp = headers.get("content-disposition")
        if disp is None:
            raise MultipartError("part without Content-Disposition")
        kind, params = parse_header_params(disp)
        if kind != "form-data" or "name" not in params:
            raise MultipartError("Content-Disposition must be form-data with a name")
        parts.append(Part(name=params["name"], data=data, filename=params.get("filename"),
                          content_type=headers.get("content-type"), headers=headers))
        if len(parts) > max_parts:
            raise MultipartError("too many parts")
        pos = nxt + 2
        if pos >= n:
            raise MultipartError("truncated body")
    return parts


class StreamingForm:
    """Incremental multipart/form-data reader.  ``feed(chunk)`` for every body chunk, then ``finish()`` -> parts.

    File parts carry ``data`` as a memoryview into the receive buffer; ``sha256_hex(name)`` is the digest of the
    file part called ``name`` (hashed incrementally on ``hash_pool`` when one is given)."""

    HASH_PIECE = 4 << 20

    def __init__(self, content_type: Optional[str], max_bytes: int, content_length: Optional[int] = None,
                 hash_field: str = "file", hash_pool=None, max_parts: int = 64, buffer_factory=None):
        """``buffer_factory(nbytes)`` -> the receive buffer for a body of known length: anything writable with the
        buffer protocol, slice assignment and ``find`` (default ``bytearray``; ``api/app.py`` passes a shared-memory
        mapping).  It is called when the first body byte is about to arrive, not when the header announces a length."""
        import hashlib

        self.delim = b"--" + boundary_from_content_type(content_type)
        self.max_bytes = int(max_bytes)
        if content_length is not None and content_length > self.max_bytes:
            raise UploadTooLarge("upload too large")
        self.fixed = content_length is not None and content_length >= 0
        self.content_length = content_length if self.fixed else None
        self._factory = buffer_factory or bytearray
        self.buf = None if self.fixed else bytearray()
        self.n = 0                       # bytes received
        self.chunks: List[bytes] = []    # without Content-Length: collected, joined once at the end
        self.hash_field = hash_field
        self.hash_pool = hash_pool
        self.max_parts = max_parts
        self._sha = hashlib.sha256()
        self._hash_futs = []             # one future per drain task; all done <=> every submitted piece is hashed
        self._hash_queue = collections.deque()
        self._hash_lock = threading.Lock()
        self._hash_running = False
        # incremental scan state (only with a preallocated buffer; otherwise everything happens in finish())
        self._pos = -1                   # -1: opening delimiter not seen yet
        self._in_file = False
        self._data_start = 0
        self._hashed_to = 0
        self._scanned_to = 0             # every byte before this has been searched for the closing delimiter
        self._file_span = None           # (start, end) of the hashed file part once its closing delimiter is seen

    def buffer(self):
        """The receive buffer of a fixed-length body (allocated on first use): ``buffer()[k]`` is body byte ``k``.  A
        server that can write the socket's bytes there itself (``api/httpd.py``'s body sink) then feeds memoryview
        slices of this very buffer, which ``feed`` recognises and does not copy."""
        if not self.fixed:
            return None
        if self.buf is None:
            self.buf = self._factory(self.content_length)
            if len(self.buf) < self.content_length:
                raise MultipartError("receive buffer smaller than the announced body")
        return self.buf

    def feed(self, chunk: bytes) -> None:
        if not len(chunk):
            return
        if self.n + len(chunk) > self.max_bytes:
            raise UploadTooLarge("upload too large")
        if not self.fixed:
            self.chunks.append(bytes(chunk))
            self.n += len(chunk)
            return
        buf = self.buffer()
        if self.n + len(chunk) > self.content_length:
            raise MultipartError("body longer than its Content-Length")
        if isinstance(chunk, memoryview) and chunk.obj is buf:
            pass                          # received in place, at body offset self.n: nothing to copy
        else:
            buf[self.n : self.n + len(chunk)] = chunk
        self.n += len(chunk)
        self._scan()

    # -- incremental scan: find where the hashed file part starts and hash it as it arrives
    def _scan(self) -> None:
        if self._file_span is not None:
            return
        buf, n, delim = self.buf, self.n, self.delim
        while not self._in_file:
            if self._pos < 0:
                p = buffer_find(buf, delim, 0, n)
                if p < 0:
                    return
                self._pos = p
            p = self._pos + len(delim)
            if n < p + 2:
                return
            if buf[p : p + 2] == b"--":
                return                                   # closing delimiter: no file part to hash
            eol = buffer_find(buf, b"\r\n", p, n)
            hdr_end = buffer_find(buf, b"\r\n\r\n", p, n) if eol >= 0 else -1
            if hdr_end < 0:
                return
            data_start = hdr_end + 4
            # the same rule the final parser applies (Content-Disposition parameters), not a substring guess
            is_target = False
            for line in bytes(buf[eol + 2 : hdr_end]).split(b"\r\n"):
                k, sep, v = line.partition(b":")
                if sep and k.strip().lower() == b"content-disposition":
                    kind, params = parse_header_params(v.decode("utf-8", "replace"))
                    is_target = (kind == "form-data" and params.get("name") == self.hash_field
                                 and params.get("filename") is not None)
            if is_target:
                self._in_file, self._data_start, self._hashed_to = True, data_start, data_start
                break
            nxt = buffer_find(buf, b"\r\n" + delim, data_start, n)
            if nxt < 0:
                return
            self._pos = nxt + 2
        if self._in_file:
            tail = len(delim) + 2
            scan_from = max(self._data_start, self._scanned_to - tail)
            end = buffer_find(buf, b"\r\n" + delim, scan_from, n)
            self._scanned_to = n
            safe = end if end >= 0 else n - tail
            if end >= 0:
                self._file_span = (self._data_start, end)
            while safe - self._hashed_to >= (self.HASH_PIECE if end < 0 else 1):
                hi = safe if end >= 0 else self._hashed_to + self.HASH_PIECE
                self._hash_piece(self._hashed_to, hi)
                self._hashed_to = hi

    def _hash_piece(self, lo: int, hi: int) -> None:
        view = memoryview(self.buf)[lo:hi]       # the buffer is preallocated and never resized: the view stays valid
        if self.hash_pool is None:
            self._sha.update(view)
            return
        # pieces are hashed in order by at most ONE pool task per form: a piece either joins the running task's queue
        # or starts a new task (no pool worker ever blocks waiting for a predecessor)
        with self._hash_lock:
            self._hash_queue.append(view)
            if not self._hash_running:
                self._hash_running = True
                self._hash_futs.append(self.hash_pool.submit(self._hash_drain))

    def _hash_drain(self) -> None:
        while True:
            with self._hash_lock:
                if not self._hash_queue:
                    self._hash_running = False
                    return
                view = self._hash_queue.popleft()
            self._sha.update(view)

    def finish(self, defer_hash: bool = False) -> List[Part]:
        """Parse the complete body.  With ``defer_hash`` the call does not wait for the hash thread to catch up with
        the last pieces (10-30 ms behind the wire on a 120 MB upload): ``sha256_pending()`` then returns a future of
        the digest, and the caller can start work on the file meanwhile."""
        import hashlib

        if self.fixed:
            if self.n != self.content_length:
                raise MultipartError("truncated body")
            body = memoryview(self.buffer())[: self.content_length]
        else:
            body = memoryview(b"".join(self.chunks))
            self.chunks = []
        # the incremental scan has already found where the hashed file part ends: the final parse does not search
        # those bytes again (a second pass over a 120 MB upload is 30-50 ms on the event loop, after the last byte)
        parts = _parse_view(body, self.delim, self.max_parts, self._file_span if self.fixed else None,
                            raw=self.buf if self.fixed else None)
        self._digest = None
        self._digest_fut = None
        tgt = next((p for p in parts if p.name == self.hash_field and p.is_file), None)
        streamed = (self._file_span is not None and tgt is not None
                    and len(tgt.data) == self._file_span[1] - self._file_span[0]
                    and self._hashed_to == self._file_span[1])
        if streamed and defer_hash and self.hash_pool is not None and self._hash_futs:
            futs = list(self._hash_futs)

            def digest():
                for f in futs:
                    f.result()
                return self._sha.hexdigest()

            self._digest_fut = self.hash_pool.submit(digest)
            return parts
        for f in self._hash_futs:
            f.result()
        if streamed:
            self._digest = self._sha.hexdigest()
        elif tgt is not None:
            self._digest = hashlib.sha256(tgt.data).hexdigest()
        return parts

    def sha256_pending(self):
        """The digest's future after ``finish(defer_hash=True)`` left the hash thread running, else None."""
        return getattr(self, "_digest_fut", None)

    def sha256_hex(self) -> Optional[str]:
        fut = getattr(self, "_digest_fut", None)
        if fut is not None:
            return fut.result()
        return getattr(self, "_digest", None)


def _parse_view(body: memoryview, delim: bytes, max_parts: int, known_span: Optional[tuple] = None, raw=None) -> List[Part]:
    """``parse_multipart`` over a memoryview; file parts keep views (no copy), form fields become bytes.
    ``known_span`` = (data_start, end) of a part whose closing delimiter an earlier scan has located: it is checked
    in place instead of searched for.  ``raw``: the object under ``body`` when ``body`` starts at its first byte
    (bytearray or mmap; may be longer than the body)."""
    # searched in place when the view covers a whole bytes / bytearray / mmap object (a copy of a 120 MB upload
    # otherwise); a longer mapping (page-rounded shared memory) is searched up to the body's end
    if raw is None:
        raw = body.obj if isinstance(body.obj, (bytes, bytearray)) and len(body.obj) == len(body) else bytes(body)
    n = len(body)
    pos = buffer_find(raw, delim, 0, n)
    if pos < 0:
        raise MultipartError("opening boundary not found")
    parts: List[Part] = []
    while True:
        pos += len(delim)
        if raw[pos:pos + 2] == b"--":
            break
        eol = buffer_find(raw, b"\r\n", pos, n)
        if eol < 0:
            raise MultipartError("malformed boundary line")
        pos = eol + 2
        hdr_end = buffer_find(raw, b"\r\n\r\n", pos, n)
        if hdr_end < 0:
            raise MultipartError("part headers not terminated")
        headers: Dict[str, str] = {}
        for line in bytes(raw[pos:hdr_end]).split(b"\r\n"):
            if not line:
                continue
            if b":" not in line:
                raise MultipartError("malformed part header")
            k, v = line.split(b":", 1)
            headers[k.decode("latin-1").strip().lower()] = v.decode("utf-8", "replace").strip()
        data_start = hdr_end + 4
        closing = b"\r\n" + delim
        if (known_span is not None and known_span[0] == data_start and known_span[1] >= data_start
                and raw[known_span[1] : known_span[1] + len(closing)] == closing):
            nxt = known_span[1]       # the first occurrence after data_start, as the incremental scan established
        else:
            nxt = buffer_find(raw, closing, data_start, n)
        if nxt < 0:
            raise MultipartError("closing boundary not found")
        disp = headers.get("content-disposition")
        if disp is None:
            raise MultipartError("part without Content-Disposition")
        kind, params = parse_header_params(disp)
        if kind != "form-data" or "name" not in params:
            raise MultipartError("Content-Disposition must be form-data with a name")
        is_file = params.get("filename") is not None
        data = body[data_start:nxt] if is_file else bytes(raw[data_start:nxt])
        parts.append(Part(name=params["name"], data=data, filename=params.get("filename"),
                          content_type=headers.get("content-type"), headers=headers, offset=data_start))
        if len(parts) > max_parts:
            raise MultipartError("too many parts")
        pos = nxt + 2
        if pos >= n:
            raise MultipartError("truncated body")
    return parts


def encode_multipart(fields: Dict[str, str], files: Dict[str, tuple], boundary: str = "----dlamdFormBoundary7MA4YWxk"):
    """Client-side encoder (tests, bench clients).  files: name -> (filename, bytes, content_type)."""
    out = []
    for k, v in fields.items():
        out.append(f'--{boundary}\r\nContent-Disposition: form-data; name="{k}"\r\n\r\n'.encode() + str(v).encode() + b"\r\n")
    for k, (fn, data, ct) in files.items():
        out.append(
            f'--{boundary}\r\nContent-Disposition: form-data; name="{k}"; filename="{fn}"\r\n'
            f"Content-Type: {ct}\r\n\r\n".encode() + data + b"\r\n"
        )
    out.append(f"--{boundary}--\r\n".encode())
    return b"".join(out), f"multipart/form-data; boundary={boundary}"
