"""ctypes binding of the native C++ PDF reader (``csrc/pdf`` -> ``libdlamd_pdf.so``).

Role in the pipeline: what ``PymuDocDataset(pdf_bytes)`` +
``ds.classify()`` do in the reference (services/ocr_service/ocr_service.py:210-213):
open a PDF from bytes, extract born-digital text, find the page images of
scans, and decide TXT vs OCR.  Every native call releases the GIL, so page
analysis and Flate image decoding scale across the host thread pool.
"""
from __future__ import annotations

import ctypes
import threading
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np

LIB_PATH = Path(__file__).resolve().parents[1] / "_lib" / "libdlamd_pdf.so"

CODEC_RAW, CODEC_JPEG, CODEC_UNSUPPORTED = 0, 1, 2
JPEG_LUT_BITS = 10


class PdfError(ValueError):
    pass


class _PageInfo(ctypes.Structure):
    _fields_ = [("width_pt", ctypes.c_double), ("height_pt", ctypes.c_double), ("rotate", ctypes.c_int32),
                ("n_images", ctypes.c_int32), ("chars_visible", ctypes.c_int64), ("chars_invisible", ctypes.c_int64),
                ("unmapped", ctypes.c_int64), ("decoded", ctypes.c_int64), ("image_coverage", ctypes.c_double),
                ("has_error", ctypes.c_int32), ("pad", ctypes.c_int32)]


class _ImageInfo(ctypes.Structure):
    _fields_ = [("obj_num", ctypes.c_int32), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("bpc", ctypes.c_int32), ("ncomp", ctypes.c_int32), ("codec", ctypes.c_int32),
                ("out_channels", ctypes.c_int32), ("pad", ctypes.c_int32), ("stream_off", ctypes.c_int64),
                ("stream_len", ctypes.c_int64), ("x0", ctypes.c_double), ("y0", ctypes.c_double),
                ("x1", ctypes.c_double), ("y1", ctypes.c_double)]


class _Classify(ctypes.Structure):
    _fields_ = [("n_pages", ctypes.c_int32), ("is_text", ctypes.c_int32), ("pages_with_text", ctypes.c_int32),
                ("pages_image_dominant", ctypes.c_int32), ("avg_chars", ctypes.c_double),
                ("garbled_ratio", ctypes.c_double), ("avg_coverage", ctypes.c_double)]


class JpegInfo(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ncomp", ctypes.c_int32),
                ("progressive", ctypes.c_int32), ("precision", ctypes.c_int32), ("restart_interval", ctypes.c_int32),
                ("mcu_w", ctypes.c_int32), ("mcu_h", ctypes.c_int32), ("mcus_x", ctypes.c_int32),
                ("mcus_y", ctypes.c_int32), ("hs", ctypes.c_int32 * 4), ("vs", ctypes.c_int32 * 4),
                ("tq", ctypes.c_int32 * 4), ("td", ctypes.c_int32 * 4), ("ta", ctypes.c_int32 * 4),
                ("blocks_per_mcu", ctypes.c_int32), ("n_segments", ctypes.c_int32), ("scan_off", ctypes.c_int64),
                ("scan_len", ctypes.c_int64), ("supported", ctypes.c_int32), ("pad", ctypes.c_int32)]


class JpegSyncStats(ctypes.Structure):
    _fields_ = [("n_sub", ctypes.c_int32), ("max_rounds", ctypes.c_int32), ("max_seam_steps", ctypes.c_int32),
                ("blocks_decoded", ctypes.c_int32), ("seam_passes", ctypes.c_int32), ("pad", ctypes.c_int32),
                ("sub_decodes", ctypes.c_int64)]


# byte size of the device-side table blob (struct JpegTables in csrc/pdf/jpeg_scan.h)
JPEG_TABLES_BYTES = 3 * 64 * 2 + 4 * (1 << JPEG_LUT_BITS) * 2 + 4 * 18 * 4 + 4 * 17 * 4 + 4 * 256

_lib: Optional[ctypes.CDLL] = None
_lock = threading.Lock()


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not LIB_PATH.exists():
                raise PdfError(f"{LIB_PATH} is missing: run `python -m distributed_llms_amd.utils.build`")
            l = ctypes.CDLL(str(LIB_PATH))
            vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
            l.dlpdf_open.argtypes = [vp, sz, ctypes.c_char_p, i]
            l.dlpdf_open.restype = vp
            l.dlpdf_close.argtypes = [vp]
            l.dlpdf_close.restype = None
            l.dlpdf_page_count.argtypes = [vp]
            l.dlpdf_was_rebuilt.argtypes = [vp]
            l.dlpdf_page_info.argtypes = [vp, i, ctypes.POINTER(_PageInfo)]
            l.dlpdf_page_text.argtypes = [vp, i, i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz)]
            l.dlpdf_page_lines.argtypes = [vp, i, i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz)]
            l.dlpdf_page_error.argtypes = [vp, i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz)]
            l.dlpdf_image_info.argtypes = [vp, i, i, ctypes.POINTER(_ImageInfo)]
            l.dlpdf_image_jpeg_bytes.argtypes = [vp, i, i, ctypes.POINTER(vp), ctypes.POINTER(sz)]
            l.dlpdf_free.argtypes = [vp]
            l.dlpdf_free.restype = None
            l.dlpdf_image_decode.argtypes = [vp, i, i, vp, sz]
            l.dlpdf_image_decode.restype = ctypes.c_longlong
            l.dlpdf_image_decode_bits.argtypes = [vp, i, i, vp, sz]
            l.dlpdf_image_decode_bits.restype = ctypes.c_longlong
            l.dlpdf_classify.argtypes = [vp, ctypes.POINTER(_Classify)]
            l.dlpdf_fast_inflate.argtypes = [vp, sz, vp, sz]
            l.dlpdf_fast_inflate.restype = ctypes.c_longlong
            l.dljpeg_sizes.argtypes = [ctypes.POINTER(i), ctypes.POINTER(i)]
            l.dljpeg_scan.argtypes = [vp, sz, ctypes.POINTER(JpegInfo), vp, vp, i]
            l.dljpeg_decode_sync_host.argtypes = [vp, sz, vp, sz, ctypes.POINTER(JpegSyncStats)]
            a, b = ctypes.c_int(), ctypes.c_int()
            l.dljpeg_sizes(ctypes.byref(a), ctypes.byref(b))
            if a.value != ctypes.sizeof(JpegInfo) or b.value != JPEG_TABLES_BYTES:
                raise PdfError(f"libdlamd_pdf ABI mismatch: JpegInfo {a.value}/{ctypes.sizeof(JpegInfo)}, "
                               f"JpegTables {b.value}/{JPEG_TABLES_BYTES} - rebuild the library")
            _lib = l
    return _lib


@dataclass
class PageInfo:
    index: int
    width_pt: float
    height_pt: float
    rotate: int
    n_images: int
    chars_visible: int
    chars_invisible: int
    unmapped: int
    decoded: int
    image_coverage: float
    has_error: bool


@dataclass
class ImageInfo:
    page: int
    index: int
    obj_num: int
    width: int
    height: int
    bpc: int
    ncomp: int
    codec: int
    out_channels: int
    stream_off: int
    stream_len: int
    box_pt: Tuple[float, float, float, float]

    @property
    def area_pt(self) -> float:
        return max(0.0, self.box_pt[2] - self.box_pt[0]) * max(0.0, self.box_pt[3] - self.box_pt[1])


@dataclass
class Classification:
    n_pages: int
    is_text: bool
    pages_with_text: int
    pages_image_dominant: int
    avg_chars: float
    garbled_ratio: float
    avg_coverage: float


@dataclass
class JpegScan:
    info: JpegInfo
    tables: np.ndarray        # uint8 [JPEG_TABLES_BYTES]
    seg_offsets: np.ndarray   # uint32 [n_segments], relative to the start of the JPEG stream

    @property
    def gpu_decodable(self) -> bool:
        return self.info.supported in (1, 2)

    @property
    def self_synchronising(self) -> bool:
        """No restart markers: the GPU decodes the scan in 128-byte subsequences that find the code-word boundaries
        by themselves (ops/jpeg.py)."""
        return self.info.supported == 2


def scan_jpeg(data, max_segments: int = 1 << 20) -> JpegScan:
    """Parse a JPEG header + locate restart segments.  ``data``: bytes / memoryview / uint8 ndarray."""
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    info = JpegInfo()
    tables = np.zeros(JPEG_TABLES_BYTES, dtype=np.uint8)
    # first pass with room for the common cases (no restart markers: one segment; a marker per MCU row: a few hundred);
    # a restart-heavy file is scanned a second time with the exact count
    cap = 4096
    segs = np.empty(cap, dtype=np.uint32)
    rc = lib().dljpeg_scan(buf.ctypes.data, buf.size, ctypes.byref(info), tables.ctypes.data, segs.ctypes.data, cap)
    if rc != 0:
        raise PdfError(f"not a decodable JPEG stream (code {rc})")
    if info.n_segments > cap:
        if info.n_segments > max_segments:
            raise PdfError("JPEG has too many restart segments")
        segs = np.zeros(info.n_segments, dtype=np.uint32)
        lib().dljpeg_scan(buf.ctypes.data, buf.size, ctypes.byref(info), tables.ctypes.data, segs.ctypes.data,
                          info.n_segments)
    return JpegScan(info, tables, segs[: info.n_segments].copy())


def decode_jpeg_sync_host(data) -> Tuple[np.ndarray, JpegSyncStats]:
    """Serial host emulation of the GPU's self-synchronising decode of a restart-less baseline JPEG
    (csrc/pdf/jpeg_sync_host.h).  Returns the gray plane [height, width] and the synchronisation statistics."""
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    sc = scan_jpeg(buf)
    if not sc.self_synchronising:
        raise PdfError("not a restart-less baseline JPEG")
    i = sc.info
    H, W = i.mcus_y * i.vs[0] * 8, i.mcus_x * i.hs[0] * 8
    out = np.empty((H, W), dtype=np.uint8)
    stats = JpegSyncStats()
    rc = lib().dljpeg_decode_sync_host(buf.ctypes.data, buf.size, out.ctypes.data, out.size, ctypes.byref(stats))
    if rc != 0:
        raise PdfError(f"self-synchronising decode failed (code {rc})")
    return out[: i.height, : i.width], stats


def fast_inflate(data, capacity: int) -> Optional[np.ndarray]:
    """The library's own zlib-stream decoder (csrc/pdf/fast_inflate.h) on ``data``: the inflated bytes, or None when
    it refuses the stream (damaged, truncated, larger than ``capacity``) - inside the library zlib then takes over."""
    src = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    out = np.empty(max(1, int(capacity)), dtype=np.uint8)
    n = lib().dlpdf_fast_inflate(src.ctypes.data, src.size, out.ctypes.data, int(capacity))
    return None if n < 0 else out[:n]


class PdfDocument:
    """A parsed PDF.  Holds a reference to the input bytes (the native side reads them in place)."""

    def __init__(self, data):
        """``data``: bytes, bytearray, memoryview or a uint8 ndarray.  Nothing is copied - the native reader works on
        the caller's buffer in place (a shared-memory view included), which therefore must outlive the document."""
        if isinstance(data, np.ndarray):
            if data.dtype != np.uint8 or not data.flags["C_CONTIGUOUS"]:
                raise TypeError("PdfDocument expects a contiguous uint8 array")
            self._np = data.reshape(-1)
        elif isinstance(data, (bytes, bytearray, memoryview)):
            self._np = np.frombuffer(data, dtype=np.uint8)
        else:
            raise TypeError("PdfDocument expects bytes")
        self._data = data
        err = ctypes.create_string_buffer(256)
        self._h = lib().dlpdf_open(self._np.ctypes.data, self._np.size, err, 256)
        if not self._h:
            raise PdfError(err.value.decode("utf-8", "replace") or "cannot open PDF")
        self.page_count = lib().dlpdf_page_count(self._h)

    # -- lifecycle
    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().dlpdf_close(self._h)
            self._h = None
        # drop the views so that a shared-memory segment behind them can be unmapped
        self._np = None
        self._data = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def data(self) -> np.ndarray:
        return self._np

    @property
    def was_rebuilt(self) -> bool:
        return bool(lib().dlpdf_was_rebuilt(self._h))

    # -- pages
    def page_info(self, i: int) -> PageInfo:
        pi = _PageInfo()
        if lib().dlpdf_page_info(self._h, i, ctypes.byref(pi)) != 0:
            raise IndexError(i)
        return PageInfo(i, pi.width_pt, pi.height_pt, pi.rotate, pi.n_images, pi.chars_visible, pi.chars_invisible,
                        pi.unmapped, pi.decoded, pi.image_coverage, bool(pi.has_error))

    def page_text(self, i: int, include_invisible: bool = False) -> str:
        p, n = ctypes.c_char_p(), ctypes.c_size_t()
        if lib().dlpdf_page_text(self._h, i, 1 if include_invisible else 0, ctypes.byref(p), ctypes.byref(n)) != 0:
            raise IndexError(i)
        return ctypes.string_at(p, n.value).decode("utf-8", "replace") if n.value else ""

    def page_lines(self, i: int, include_invisible: bool = False) -> List[Tuple[Tuple[float, float, float, float], float, str]]:
        """Positioned text lines of a page: [((x0, y0, x1, y1), font size, text)] with y measured DOWN from the top
        of the page (points), in the native reader's top-to-bottom, left-to-right order."""
        p, n = ctypes.c_char_p(), ctypes.c_size_t()
        if lib().dlpdf_page_lines(self._h, i, 1 if include_invisible else 0, ctypes.byref(p), ctypes.byref(n)) != 0:
            raise IndexError(i)
        if not n.value:
            return []
        pi = _PageInfo()
        lib().dlpdf_page_info(self._h, i, ctypes.byref(pi))
        out = []
        for rec in ctypes.string_at(p, n.value).decode("utf-8", "replace").split("\n"):
            f = rec.split("\t", 5)
            if len(f) != 6:
                continue
            x0, y0, x1, y1, size = (float(v) for v in f[:5])
            out.append(((x0, pi.height_pt - y1, x1, pi.height_pt - y0), size, f[5]))
        return out

    def page_error(self, i: int) -> str:
        p, n = ctypes.c_char_p(), ctypes.c_size_t()
        lib().dlpdf_page_error(self._h, i, ctypes.byref(p), ctypes.byref(n))
        return ctypes.string_at(p, n.value).decode("utf-8", "replace") if n.value else ""

    def page_images(self, i: int) -> List[ImageInfo]:
        out = []
        for k in range(self.page_info(i).n_images):
            ii = _ImageInfo()
            if lib().dlpdf_image_info(self._h, i, k, ctypes.byref(ii)) != 0:
                continue
            out.append(ImageInfo(i, k, ii.obj_num, ii.width, ii.height, ii.bpc, ii.ncomp, ii.codec, ii.out_channels,
                                 ii.stream_off, ii.stream_len, (ii.x0, ii.y0, ii.x1, ii.y1)))
        return out

    def main_image(self, i: int) -> Optional[ImageInfo]:
        """The image covering most of the page (a scanned page has exactly one)."""
        imgs = [im for im in self.page_images(i) if im.width > 0 and im.height > 0]
        return max(imgs, key=lambda im: im.area_pt) if imgs else None

    def jpeg_bytes(self, im: ImageInfo) -> np.ndarray:
        """Encoded JPEG stream of a DCT image as a uint8 view (zero-copy when DCT is the only filter)."""
        if im.codec != CODEC_JPEG:
            raise PdfError("image is not DCT encoded")
        if im.stream_off >= 0:
            return self._np[im.stream_off : im.stream_off + im.stream_len]
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        rc = lib().dlpdf_image_jpeg_bytes(self._h, im.page, im.index, ctypes.byref(p), ctypes.byref(n))
        if rc != 0:
            raise PdfError(f"cannot unwrap DCT stream (code {rc})")
        try:
            return np.frombuffer(ctypes.string_at(p, n.value), dtype=np.uint8)
        finally:
            lib().dlpdf_free(p)

    def decode_image(self, im: ImageInfo, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Decode a non-JPEG image to uint8 [H, W] or [H, W, 3] (optionally into ``out``, e.g. pinned memory)."""
        if im.codec != CODEC_RAW:
            raise PdfError("image needs a codec-specific decoder")
        shape = (im.height, im.width) if im.out_channels == 1 else (im.height, im.width, 3)
        if out is None:
            out = np.empty(shape, dtype=np.uint8)
        need = int(np.prod(shape))
        assert out.dtype == np.uint8 and out.size >= need and out.flags["C_CONTIGUOUS"]
        rc = lib().dlpdf_image_decode(self._h, im.page, im.index, out.ctypes.data, out.size)
        if rc < 0:
            raise PdfError(f"image decode failed (code {rc})")
        return out.reshape(-1)[:need].reshape(shape)

    def decode_image_bits(self, im: ImageInfo, out: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
        """A plain bilevel image as packed bits: uint8 [H, (W + 7) // 8], most significant bit first, set bit = white.
        None when the image is not one (more than one bit per sample, a palette, a mask): use ``decode_image``."""
        if im.codec != CODEC_RAW or im.bpc != 1 or im.out_channels != 1:
            return None
        rowbytes = (im.width + 7) // 8
        if out is None:
            out = np.empty((im.height, rowbytes), dtype=np.uint8)
        assert out.dtype == np.uint8 and out.size >= im.height * rowbytes and out.flags["C_CONTIGUOUS"]
        rc = lib().dlpdf_image_decode_bits(self._h, im.page, im.index, out.ctypes.data, out.size)
        if rc == -6:
            return None
        if rc < 0:
            raise PdfError(f"image decode failed (code {rc})")
        return out.reshape(-1)[: im.height * rowbytes].reshape(im.height, rowbytes)

    def classify(self) -> Classification:
        c = _Classify()
        if lib().dlpdf_classify(self._h, ctypes.byref(c)) != 0:
            raise PdfError("classification failed")
        return Classification(c.n_pages, bool(c.is_text), c.pages_with_text, c.pages_image_dominant, c.avg_chars,
                              c.garbled_ratio, c.avg_coverage)
