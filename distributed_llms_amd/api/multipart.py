"""multipart/form-data parser (RFC 7578) for ``POST /extract``.

FastAPI's ``File``/``Form`` need the ``python-multipart`` package (the
reference depends on it, services/ocr_service/requirements.in); it is not a
dependency here.  Uploads are whole PDFs that the reference also reads fully
into memory (ocr_service.py:437), so a single-pass parser over the complete
body with ``bytes.find`` is both simpler and faster than a streaming state
machine.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional
from urllib.parse import unquote


class MultipartError(ValueError):
    pass


@dataclass
class Part:
    name: str
    data: bytes
    filename: Optional[str] = None
    content_type: Optional[str] = None
    headers: Dict[str, str] = field(default_factory=dict)

    @property
    def is_file(self) -> bool:
        return self.filename is not None

    def text(self, encoding: str = "utf-8") -> str:
        return self.data.decode(encoding, "replace")


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
