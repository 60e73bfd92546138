"""Synthetic documents: rendered "scanned" pages, text lines for recognizer training, and PDF files.

There is no dataset and no network in the build environment, and the reference
ships no fixtures, so every workload is generated here:

* ``render_page``      - one A4 page at 300 DPI (2480 x 3508, 8-bit gray) with
  paragraphs of text, plus the ground-truth line boxes and strings;
* ``render_line``      - one text-line crop for CTC training (BASELINE config 5);
* ``make_scanned_pdf`` - a PDF whose pages are one full-page image XObject each
  (DCTDecode with restart markers, or FlateDecode) - the shape a scanner emits
  and the workload of BASELINE configs 2 and 3;
* ``make_text_pdf``    - a born-digital PDF with Type-1 Helvetica text
  (BASELINE config 1; exercises the C++ reader's text operators).

PIL (with FreeType and libjpeg-turbo) is the only dependency.
"""
from __future__ import annotations

import io
import zlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
from PIL import Image, ImageDraw, ImageFont

A4_300DPI = (2480, 3508)  # width, height in pixels

_WORDS = (
    "the of and to in is that for it as was with be by on not he this are or his from at which but have an "
    "had they you were their one all we can her has there been if more when will would who so no out up into "
    "page batch kernel wave tensor matrix memory stream graph device vector scalar cache tile fused compute "
    "document service status result extract upload health queue worker shard gather reduce scatter overlap "
    "Invoice Total Amount Date Number Reference Section Figure Table Chapter Summary Report Annex Appendix "
    "2024 2025 1,250.00 37% 4.5 No. 12 (a) [3] A-17 x=3 y<9 #42 $99 & / : ; ? !"
).split()

_FONT_CACHE = {}


def font(size: int, bold: bool = False) -> ImageFont.FreeTypeFont:
    key = (size, bold)
    f = _FONT_CACHE.get(key)
    if f is None:
        try:
            name = "DejaVuSans-Bold.ttf" if bold else "DejaVuSans.ttf"
            f = ImageFont.truetype(name, size)
        except OSError:
            f = ImageFont.load_default(size=size)
        _FONT_CACHE[key] = f
    return f


def random_text(rng: np.random.Generator, max_chars: int, min_words: int = 1) -> str:
    words: List[str] = []
    n = 0
    while True:
        w = _WORDS[int(rng.integers(len(_WORDS)))]
        if rng.random() < 0.08:
            w = w.capitalize()
        if n + len(w) + (1 if words else 0) > max_chars and len(words) >= min_words:
            break
        words.append(w)
        n += len(w) + 1
        if n >= max_chars:
            break
    return " ".join(words)[:max_chars].strip()


@dataclass
class PageTruth:
    boxes: np.ndarray            # [n, 4] float32 x0, y0, x1, y1 (pixels, reading order)
    texts: List[str] = field(default_factory=list)


def render_page(seed: int, size: Tuple[int, int] = A4_300DPI, font_px: int = 42, noise: float = 2.0,
                margin: int = 220) -> Tuple[np.ndarray, PageTruth]:
    """One synthetic scanned page: uint8 [H, W] (0 = ink, ~245 = paper) and its line truth."""
    W, H = size
    rng = np.random.default_rng(seed)
    img = Image.new("L", (W, H), 246)
    d = ImageDraw.Draw(img)
    f = font(font_px)
    fb = font(int(font_px * 1.35), bold=True)
    line_h = int(font_px * 1.55)
    y = margin
    boxes, texts = [], []
    avg_char = f.getlength("abcdefghij") / 10.0
    max_chars = max(8, int((W - 2 * margin) / avg_char))
    first = True
    while y + line_h < H - margin:
        if first or rng.random() < 0.07:
            # heading
            t = random_text(rng, max(8, max_chars // 3)).title()
            ff = fb
            first = False
            y += line_h // 2
        else:
            # paragraph line; the last line of a paragraph is short
            frac = 1.0 if rng.random() > 0.18 else float(rng.uniform(0.25, 0.8))
            t = random_text(rng, max(8, int(max_chars * frac)), min_words=2)
            ff = f
        if not t:
            continue
        x = margin + (0 if rng.random() > 0.1 else int(rng.integers(40, 160)))
        bb = d.textbbox((x, y), t, font=ff)
        if bb[2] > W - margin // 2:
            continue
        d.text((x, y), t, fill=int(rng.integers(10, 50)), font=ff)
        boxes.append([bb[0] - 4, bb[1] - 4, bb[2] + 4, bb[3] + 4])
        texts.append(t)
        y += line_h if ff is f else int(line_h * 1.5)
        if rng.random() < 0.12:
            y += line_h  # paragraph gap
    arr = np.asarray(img, dtype=np.uint8)
    if noise > 0:
        # scanner-like sensor noise on a coarse grid (keeps JPEG/Flate sizes realistic, cheap to make)
        nz = rng.normal(0.0, noise, size=(H // 4 + 1, W // 4 + 1)).astype(np.float32)
        nz = np.repeat(np.repeat(nz, 4, axis=0), 4, axis=1)[:H, :W]
        arr = np.clip(arr.astype(np.float32) + nz, 0, 255).astype(np.uint8)
    return arr, PageTruth(np.asarray(boxes, dtype=np.float32).reshape(-1, 4), texts)


def render_line(text: str, height: int = 48, font_px: int = 34, seed: int = 0, pad: int = 6,
                bold: bool = False) -> np.ndarray:
    """One text line as uint8 [height, w] (dark ink on light paper), tightly cropped."""
    rng = np.random.default_rng(seed)
    f = font(font_px, bold)
    w = int(f.getlength(text)) + 2 * pad + 2
    img = Image.new("L", (max(w, 8), height), int(rng.integers(225, 256)))
    d = ImageDraw.Draw(img)
    bb = d.textbbox((0, 0), text, font=f)
    y = (height - (bb[3] - bb[1])) // 2 - bb[1]
    d.text((pad, y), text, fill=int(rng.integers(0, 70)), font=f)
    return np.asarray(img, dtype=np.uint8)


# --------------------------------------------------------------------------------------- PDF writing
class _PdfWriter:
    """Minimal PDF 1.4 writer (classic xref table) or 1.5 (object streams + xref stream)."""

    def __init__(self) -> None:
        self.objs: List[bytes] = []  # object bodies, 1-based ids

    def add(self, body: bytes) -> int:
        self.objs.append(body)
        return len(self.objs)

    def reserve(self) -> int:
        self.objs.append(b"null")
        return len(self.objs)

    def set(self, oid: int, body: bytes) -> None:
        self.objs[oid - 1] = body

    @staticmethod
    def stream(dict_entries: bytes, data: bytes) -> bytes:
        return b"<< " + dict_entries + b" /Length " + str(len(data)).encode() + b" >>\nstream\n" + data + b"\nendstream"

    def finish(self, root: int, info: Optional[int] = None, xref_stream: bool = False) -> bytes:
        out = io.BytesIO()
        out.write(b"%PDF-1.5\n%\xe2\xe3\xcf\xd3\n" if xref_stream else b"%PDF-1.4\n%\xe2\xe3\xcf\xd3\n")
        offsets = [0] * (len(self.objs) + 1)
        for i, body in enumerate(self.objs, start=1):
            offsets[i] = out.tell()
            out.write(f"{i} 0 obj\n".encode() + body + b"\nendobj\n")
        n = len(self.objs) + 1
        if not xref_stream:
            xref_pos = out.tell()
            out.write(f"xref\n0 {n}\n".encode())
            out.write(b"0000000000 65535 f \n")
            for i in range(1, n):
                out.write(f"{offsets[i]:010d} 00000 n \n".encode())
            tr = f"trailer\n<< /Size {n} /Root {root} 0 R" + (f" /Info {info} 0 R" if info else "") + " >>\n"
            out.write(tr.encode())
            out.write(f"startxref\n{xref_pos}\n%%EOF\n".encode())
        else:
            xid = n
            xref_pos = out.tell()
            rows = bytearray()
            rows += bytes([0, 0, 0, 0, 0, 255])  # object 0: free
            for i in range(1, n):
                rows += bytes([1]) + offsets[i].to_bytes(4, "big") + bytes([0])
            rows += bytes([1]) + xref_pos.to_bytes(4, "big") + bytes([0])
            data = zlib.compress(bytes(rows))
            d = (f"/Type /XRef /Size {n + 1} /W [1 4 1] /Root {root} 0 R".encode()
                 + (f" /Info {info} 0 R".encode() if info else b"") + b" /Filter /FlateDecode")
            out.write(f"{xid} 0 obj\n".encode() + self.stream(d, data) + b"\nendobj\n")
            out.write(f"startxref\n{xref_pos}\n%%EOF\n".encode())
        return out.getvalue()


def encode_page_jpeg(page: np.ndarray, quality: int = 85, restart_blocks: int = 62) -> bytes:
    """Baseline JPEG with a restart marker every ``restart_blocks`` MCUs (0: none).

    Restart intervals make the entropy-coded segment decodable in parallel, which is
    what the GPU decoder (csrc/kernels/jpeg_decode.hip) uses; files without them are
    decoded on the host thread pool instead."""
    im = Image.fromarray(page)
    buf = io.BytesIO()
    kw = dict(format="JPEG", quality=quality, optimize=False, progressive=False)
    if restart_blocks > 0:
        kw["restart_marker_blocks"] = restart_blocks
    im.save(buf, **kw)
    return buf.getvalue()


def encode_page_ccitt_g4(page: np.ndarray, threshold: int = 128) -> Tuple[bytes, bool]:
    """Gray page -> (Group 4 fax strip, BlackIs1 flag for the PDF image) through PIL's libtiff writer (one strip)."""
    import io

    from PIL import Image

    gray = page if page.ndim == 2 else page.mean(axis=2)
    im = Image.fromarray(((gray >= threshold) * 255).astype(np.uint8)).convert("1")
    buf = io.BytesIO()
    im.save(buf, format="TIFF", compression="group4", tiffinfo={278: im.height})
    raw = buf.getvalue()
    t = Image.open(io.BytesIO(raw))
    offs, cnts = t.tag_v2[273], t.tag_v2[279]
    assert len(offs) == 1, "expected a single strip"
    # photometric 1 (BlackIsZero): the coder's "white" runs are the file's 0 bits = black pixels
    return raw[offs[0] : offs[0] + cnts[0]], int(t.tag_v2.get(262, 0)) == 1


def make_scanned_pdf(pages: Sequence[np.ndarray], codec: str = "jpeg", quality: int = 85, dpi: int = 300,
                     restart_blocks: int = 62, xref_stream: bool = False,
                     encoded: Optional[Sequence[bytes]] = None, rotate: int = 0) -> bytes:
    """PDF with one full-page image per page.  ``pages``: uint8 [H, W] (gray) or [H, W, 3] arrays.

    ``encoded`` may carry pre-encoded image streams (same order) to avoid re-encoding.
    ``rotate`` (0/90/180/270): what a scanner's "landscape feed" produces - the stored raster is turned
    counter-clockwise by that angle and the page carries ``/Rotate`` so that a viewer shows it upright."""
    assert rotate in (0, 90, 180, 270)
    if rotate:
        assert encoded is None
        pages = [np.ascontiguousarray(np.rot90(pg, k=rotate // 90)) for pg in pages]
    w = _PdfWriter()
    catalog = w.reserve()
    pages_id = w.reserve()
    kids = []
    for i, pg in enumerate(pages):
        H, W = pg.shape[:2]
        cs = b"/DeviceGray" if pg.ndim == 2 else b"/DeviceRGB"
        if codec == "jpeg":
            data = encoded[i] if encoded is not None else encode_page_jpeg(pg, quality, restart_blocks)
            flt = b"/DCTDecode"
        elif codec == "flate":
            data = encoded[i] if encoded is not None else zlib.compress(np.ascontiguousarray(pg).tobytes(), 1)
            flt = b"/FlateDecode"
        elif codec == "ccitt":
            # bilevel Group 4 fax data, what office scanners emit for black-and-white scans
            data, black_is_1 = encode_page_ccitt_g4(pg)
            img = w.add(w.stream(
                b"/Type /XObject /Subtype /Image /Width %d /Height %d /ColorSpace /DeviceGray /BitsPerComponent 1 "
                b"/Filter /CCITTFaxDecode /DecodeParms << /K -1 /Columns %d /Rows %d /BlackIs1 %s >>"
                % (W, H, W, H, b"true" if black_is_1 else b"false"), data))
            flt = None
        else:
            raise ValueError("codec must be jpeg, flate or ccitt")
        if flt is not None:
            img = w.add(w.stream(
                b"/Type /XObject /Subtype /Image /Width %d /Height %d /ColorSpace %s /BitsPerComponent 8 /Filter %s"
                % (W, H, cs, flt), data))
        wpt, hpt = W * 72.0 / dpi, H * 72.0 / dpi
        content = b"q %.2f 0 0 %.2f 0 0 cm /Im0 Do Q" % (wpt, hpt)
        cid = w.add(w.stream(b"", content))
        pid = w.add(
            b"<< /Type /Page /Parent %d 0 R /MediaBox [0 0 %.2f %.2f] /Resources << /XObject << /Im0 %d 0 R >> >> "
            b"/Contents %d 0 R%s >>" % (pages_id, wpt, hpt, img, cid, b" /Rotate %d" % rotate if rotate else b""))
        kids.append(pid)
    w.set(pages_id, b"<< /Type /Pages /Count %d /Kids [%s] >>" % (len(kids), b" ".join(b"%d 0 R" % k for k in kids)))
    w.set(catalog, b"<< /Type /Catalog /Pages %d 0 R >>" % pages_id)
    return w.finish(catalog, xref_stream=xref_stream)


def _pdf_escape(s: str) -> bytes:
    b = s.encode("latin-1", "replace")
    return b.replace(b"\\", b"\\\\").replace(b"(", b"\\(").replace(b")", b"\\)")


def make_text_pdf(pages: Sequence[Sequence[str]], font_size: float = 11.0, compress: bool = True,
                  xref_stream: bool = False, use_tj_arrays: bool = False) -> bytes:
    """Born-digital PDF: each page is a list of text lines set in Helvetica (WinAnsi)."""
    w = _PdfWriter()
    catalog = w.reserve()
    pages_id = w.reserve()
    font_id = w.add(b"<< /Type /Font /Subtype /Type1 /BaseFont /Helvetica /Encoding /WinAnsiEncoding >>")
    kids = []
    for lines in pages:
        ops = [b"BT", b"/F1 %.1f Tf" % font_size, b"%.1f TL" % (font_size * 1.4), b"72 770 Td"]
        for ln in lines:
            if use_tj_arrays and " " in ln:
                parts = ln.split(" ")
                arr = b" -278 ".join(b"(" + _pdf_escape(p) + b")" for p in parts)
                ops.append(b"[" + arr + b"] TJ T*")
            else:
                ops.append(b"(" + _pdf_escape(ln) + b") Tj T*")
        ops.append(b"ET")
        content = b"\n".join(ops)
        if compress:
            cid = w.add(w.stream(b"/Filter /FlateDecode", zlib.compress(content)))
        else:
            cid = w.add(w.stream(b"", content))
        pid = w.add(
            b"<< /Type /Page /Parent %d 0 R /MediaBox [0 0 595.28 841.89] "
            b"/Resources << /Font << /F1 %d 0 R >> >> /Contents %d 0 R >>" % (pages_id, font_id, cid))
        kids.append(pid)
    w.set(pages_id, b"<< /Type /Pages /Count %d /Kids [%s] >>" % (len(kids), b" ".join(b"%d 0 R" % k for k in kids)))
    w.set(catalog, b"<< /Type /Catalog /Pages %d 0 R >>" % pages_id)
    return w.finish(catalog, xref_stream=xref_stream)
