"""Document-level extraction: PDF bytes -> combined text.

This is the MI355X-native ``process_pdf_sync`` (services/ocr_service/ocr_service.py:201-255):

    PymuDocDataset(bytes)         -> pdf.reader.PdfDocument (C++)
    ds.classify()                 -> PdfDocument.classify()            TXT | OCR
    apply(doc_analyze, ocr=True)  -> OcrEngine (HIP kernels) over page batches
    pipe_txt_mode                 -> native text-operator extraction
    get_content_list / markdown   -> ``content_list`` of {'type': 'text', 'text': ..., 'page_idx': n}
    join 'text' with "\\n\\n"       -> ``combine_text`` (same rule, same markdown fallback)

Page images reach the GPU in the cheapest form available: baseline JPEGs (with
or without restart markers) are shipped compressed and decoded by the GPU
kernels; progressive / 12-bit JPEGs go through the host decoder and Flate / CCITT
images through the native decoders on a host thread pool that runs ahead of the
GPU by two batches.
"""
from __future__ import annotations

import os
import re
import time
from concurrent.futures import Future, ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ..ops import jpeg as jpeg_ops
from ..pdf import reader
from . import layout
from .engine import Figure, OcrEngine, PageImage, PageResult, StageTimer, TextLine, _to_gray, balanced_chunks


class ExtractionError(RuntimeError):
    pass


@dataclass
class DocumentResult:
    document_id: str
    mode: str                                   # "txt" | "ocr"
    pages: List[PageResult] = field(default_factory=list)
    content_list: List[dict] = field(default_factory=list)
    markdown: str = ""
    text: str = ""
    timings: Dict[str, float] = field(default_factory=dict)
    error: Optional[str] = None                 # set instead of content when the document could not be read
    language: str = "und"                       # utils/langid.detect over the combined text ("und": too little text)
    language_confidence: float = 0.0
    warnings: List[str] = field(default_factory=list)   # pages that could not be read completely (damaged image streams)
    discarded: List[dict] = field(default_factory=list)  # running headers / footers left out of the content list


_MD_HEADER = re.compile(r"#+\s+")
_MD_MARKS = re.compile(r"\*\*|\*|__|\||---|___")


def combine_text(content_list: Sequence, markdown: str = "") -> str:
    """The reference's combined-text rule (ocr_service.py:229-248)."""
    out = ""
    for item in content_list:
        if isinstance(item, dict) and "text" in item:
            out += item["text"] + "\n\n"
        elif isinstance(item, str):
            out += item + "\n\n"
    if not out.strip():
        clean = _MD_HEADER.sub("", markdown or "")
        clean = _MD_MARKS.sub("", clean)
        out = clean
    return out.strip()


def lines_to_paragraphs(lines: Sequence[TextLine]) -> List[str]:
    """Group recognised lines (already in reading order) into paragraphs: a new paragraph starts at a text-block
    boundary of the layout analysis, after a vertical gap, or at a clear change of line height."""
    paras: List[List[str]] = []
    prev: Optional[TextLine] = None
    for ln in lines:
        if not ln.text.strip():
            continue
        new = prev is None
        if prev is not None:
            h = max(1.0, min(prev.box[3] - prev.box[1], ln.box[3] - ln.box[1]))
            gap = ln.box[1] - prev.box[3]
            same_row = abs((ln.box[1] + ln.box[3]) - (prev.box[1] + prev.box[3])) * 0.5 < 0.5 * h
            new = ln.block != prev.block or (
                not same_row and (gap > 0.8 * h or abs((ln.box[3] - ln.box[1]) - (prev.box[3] - prev.box[1])) > 0.35 * h))
        if new:
            paras.append([ln.text.strip()])
        else:
            paras[-1].append(ln.text.strip())
        prev = ln
    return [" ".join(p) for p in paras]


MARGIN_BAND = 0.075          # share of the page height at the top / bottom where running lines live (runtime/layout.py)


def table_item(grid: Sequence[Sequence[str]], page_idx: int, bbox: Sequence[float]) -> dict:
    """Content-list item of a table from its cell texts, row-major.  ``table_body`` is HTML as in magic-pdf's content
    list; ``text`` (rows on lines, cells separated by " | ") is what the combined text picks up
    (services/ocr_service/ocr_service.py:229-243 takes every item that has a 'text' key)."""
    import html

    body = "".join("<tr>" + "".join("<td>%s</td>" % html.escape(c) for c in row) + "</tr>" for row in grid)
    return {"type": "table", "table_body": "<table>%s</table>" % body, "text": "\n".join(" | ".join(r) for r in grid),
            "table_caption": [], "table_footnote": [], "page_idx": page_idx, "bbox": [round(float(v), 1) for v in bbox],
            "cells": [list(r) for r in grid]}


def table_grid(cells: Sequence[TextLine]) -> List[List[str]]:
    """Cells of one table (row by row, left to right, as the layout analysis orders them) -> rows x columns of text.
    A new row starts where x stops growing or the line changes.  The columns are bounded as the layout analysis bounds them
    (``runtime/layout.py::table_cuts``): by the x-intervals, at least 0.6 line heights wide, that at most ``tol`` cells
    reach into - from six rows on one over-long cell may run up to its neighbour.  A cell belongs to the column its
    left edge lies in; a row that lacks a cell gets an empty string there; the further lines of cells that wrap are joined
    to the cell above them."""
    import bisect

    rows: List[List[TextLine]] = []
    hs = sorted(c.box[3] - c.box[1] for c in cells)
    for c in cells:
        first = rows[-1][0] if rows else None
        if (first is not None and c.box[0] > rows[-1][-1].box[0] and
                abs(0.5 * (c.box[1] + c.box[3]) - 0.5 * (first.box[1] + first.box[3])) <= 0.5 * hs[len(hs) // 2]):
            rows[-1].append(c)
        else:
            rows.append([c])
    tol = 1 if sum(1 for r in rows if len(r) >= 2) >= 6 else 0
    min_gap = 0.6 * hs[len(hs) // 2]
    x0 = [c.box[0] for c in cells]
    x1 = [c.box[2] for c in cells]
    edges = sorted(set(x0) | set(x1))
    cuts: List[float] = []
    run = None                       # the current run of sparsely covered x-intervals
    for a, e in zip(edges[:-1], edges[1:]):
        cover = sum(1 for p, q in zip(x0, x1) if p < e and q > a)
        if cover <= tol:
            run = (a, e) if run is None else (run[0], e)
            continue
        if run is not None and run[1] - run[0] >= min_gap and any(q <= run[0] + 1e-9 for q in x1):
            cuts.append(0.5 * (run[0] + run[1]))
        run = None
    grid: List[List[str]] = []
    base_cols: set = set()
    base_size = 0
    h_med = hs[len(hs) // 2]
    cols_of = [[bisect.bisect_right(cuts, c.box[0] + 1e-6) for c in r] for r in rows]
    most = max(len(r) for r in rows)
    full = sorted(c.box[1] - a.box[3] for n in range(1, len(rows))
                  if len(rows[n]) == most and len(rows[n - 1]) == most and cols_of[n] == cols_of[n - 1]
                  for a, c in zip(rows[n - 1], rows[n]))
    ref_gap = (full[len(full) // 2] if len(full) % 2 else 0.5 * (full[len(full) // 2 - 1] + full[len(full) // 2])) if full else None
    limit = (1.0 if ref_gap is not None else 0.6) * h_med
    bottom = [0.0] * (len(cuts) + 1)             # lower edge of the last box seen in each column
    for n, r in enumerate(rows):
        cols = cols_of[n]
        # the next line of cells that wrap (``runtime/layout.py::continues_row``): fewer boxes than the row has cells,
        # all in columns the row has a cell in, each at line spacing under the box above it - joined to the cells above
        wraps = (grid and len(r) < base_size and set(cols) <= base_cols and
                 all(c.box[1] - bottom[k] <= limit and (ref_gap is None or c.box[1] - bottom[k] < 0.75 * ref_gap) and
                     (k >= len(cuts) or c.box[2] <= cuts[k] + 0.1 * h_med) for c, k in zip(r, cols)))
        if wraps:
            out = grid[-1]
        else:
            out = [""] * (len(cuts) + 1)
            grid.append(out)
            base_cols, base_size = set(cols), len(r)
        for c, k in zip(r, cols):
            out[k] = (out[k] + " " + c.text.strip()).strip()
            bottom[k] = c.box[3]
    return grid


def _line_items(lines: Sequence[TextLine], page_idx: int, page_h: float = 0.0) -> List[dict]:
    """Text paragraphs, with every display formula as an item of its own between them:
    {'type': 'equation', 'text': '$$ ... $$', 'text_format': 'latex'} (magic-pdf's shape for interline equations; the
    'text' key is what the combined text picks up, services/ocr_service/ocr_service.py:229-243), every table as one
    'table' item, and every line that lies in the top / bottom margin band as an item of its own marked
    ``'margin': 'top' | 'bottom'`` - ``drop_running_lines`` decides over the whole document which of those are running
    headers / footers."""
    items: List[dict] = []
    run: List[TextLine] = []
    cells: List[TextLine] = []

    def flush():
        for para in lines_to_paragraphs(run):
            items.append({"type": "text", "text": para, "page_idx": page_idx})
        run.clear()

    def flush_table():
        if cells:
            bb = (min(c.box[0] for c in cells), min(c.box[1] for c in cells), max(c.box[2] for c in cells),
                  max(c.box[3] for c in cells))
            items.append(table_item(table_grid(cells), page_idx, bb))
            cells.clear()

    top, bot = MARGIN_BAND * page_h, (1.0 - MARGIN_BAND) * page_h
    for ln in lines:
        kind = getattr(ln, "kind", "text")
        if kind == "cell":
            flush()
            if cells and cells[-1].block != ln.block:
                flush_table()
            cells.append(ln)
            continue
        flush_table()
        if kind == "equation":
            flush()
            items.append({"type": "equation", "text": ("$$ %s $$" % ln.text) if ln.text else "", "text_format": "latex",
                          "page_idx": page_idx, "bbox": [round(float(v), 1) for v in ln.box]})
        elif page_h > 0 and (ln.box[3] <= top or ln.box[1] >= bot):
            flush()
            if ln.text.strip():
                items.append({"type": "text", "text": ln.text.strip(), "page_idx": page_idx,
                              "margin": "top" if ln.box[3] <= top else "bottom"})
        else:
            run.append(ln)
    flush()
    flush_table()
    return items


_DIGITS = re.compile(r"\d+")
_SPACES = re.compile(r"\s+")


_NOT_WORD = re.compile(r"[^\w#]+", re.UNICODE)


def _running_key(text: str) -> str:
    """What must repeat for a margin line to count as running: its letters, with every number as one '#' and
    punctuation and spacing ignored ("- 12 -", "-13-" and "14" are the same line; so are "Page 3 of 40" and "page 4 of 40")."""
    return _NOT_WORD.sub(" ", _DIGITS.sub("#", text.lower())).strip()


def drop_running_lines(content: List[dict], n_pages: int) -> Tuple[List[dict], List[dict]]:
    """Running headers and footers: a line in the top / bottom margin band whose text - digits aside, so that page
    numbers count - shows up on three or more pages of the document, or on two when those are at least 40 % of it.
    magic-pdf's layout model labels such blocks and the reference's content list leaves them out
    (services/ocr_service/ocr_service.py:215-243); here the label comes from position plus repetition across pages.
    -> (content without them, the dropped items as {'type': 'header' | 'footer', 'text', 'page_idx'}).  A
    single-page document keeps everything: one page cannot show what repeats."""
    where: Dict[Tuple[str, str], set] = {}
    for it in content:
        m = it.get("margin")
        if m:
            key = (m, _running_key(it["text"]))
            where.setdefault(key, set()).add(it["page_idx"])
    running = {k for k, pages in where.items() if len(pages) >= 3 or (len(pages) >= 2 and len(pages) >= 0.4 * n_pages)}
    kept, dropped = [], []
    for it in content:
        m = it.pop("margin", None)
        if m and (m, _running_key(it["text"])) in running:
            dropped.append({"type": "header" if m == "top" else "footer", "text": it["text"], "page_idx": it["page_idx"]})
        else:
            kept.append(it)
    return kept, dropped


_FIG_CAPTION = re.compile(r"^(figure|fig\.?|abbildung|abb\.?|figura|рисунок|рис\.?|εικόνα|chart|image)\s*\d+", re.I)
_TAB_CAPTION = re.compile(r"^(table|tabelle|tab\.?|tabla|tableau|таблица|табл\.?|πίνακας)\s*\d+", re.I)


def attach_captions(items: List[dict]) -> None:
    """The paragraph next to a figure / table that starts like a caption ("Figure 3:", "Abb. 2", "Table 1.") is
    recorded on that item as ``img_caption`` / ``table_caption`` (magic-pdf's content-list fields).  Unlike there, the
    paragraph also stays in the list as text: the reference's combined text takes only items with a 'text' key
    (services/ocr_service/ocr_service.py:229-243), and a caption is worth keeping in it."""
    for k, it in enumerate(items):
        if it["type"] == "image":
            pat, key, near = _FIG_CAPTION, "img_caption", (k + 1, k - 1)       # usually below the figure
        elif it["type"] == "table":
            pat, key, near = _TAB_CAPTION, "table_caption", (k - 1, k + 1)     # usually above the table
        else:
            continue
        for j in near:
            if 0 <= j < len(items) and items[j]["type"] == "text" and pat.match(items[j]["text"]):
                it[key] = [items[j]["text"]]
                break


def page_content_items(pg: PageResult) -> List[dict]:
    """Content-list items of one page.  Image items carry ``'figure'``: the position of their crop in
    ``pg.figures`` (dropped by ``build_content_list``; the packed wire form keeps it, runtime/wire.py)."""
    items: List[dict] = []
    figs = sorted(enumerate(pg.figures), key=lambda kf: kf[1].order)
    # figures keep their place in the reading sequence: Figure.order counts lines + figures before it
    pos, li = 0, 0
    for k, fg in figs:
        take = max(0, fg.order - pos)
        items += _line_items(pg.lines[li : li + take], pg.page_index, float(pg.height))
        li += take
        pos += take + 1
        items.append({"type": "image", "img_path": fg.path, "img_caption": [], "img_footnote": [], "page_idx": pg.page_index,
                      "bbox": [round(float(v), 1) for v in fg.box], "figure": k})
    items += _line_items(pg.lines[li:], pg.page_index, float(pg.height))
    attach_captions(items)
    return items


def content_markdown(content: Sequence[dict]) -> str:
    """Markdown of a content list: paragraphs and formulas as they are, figures as image links, tables as HTML (the
    form magic-pdf writes them in)."""
    return "\n\n".join(("![](%s)" % it["img_path"]) if it["type"] == "image" else
                       (it["table_body"] if it["type"] == "table" else it["text"]) for it in content)


def build_content_list(pages: Sequence[PageResult]) -> List[dict]:
    """magic-pdf style content list: {'type': 'text', 'text', 'page_idx'} paragraphs, {'type': 'equation', ...}
    display formulas and {'type': 'image', 'img_path', 'page_idx'} figures in reading order.  Only items with a 'text'
    key reach the combined text (services/ocr_service/ocr_service.py:229-243)."""
    items = []
    for pg in pages:
        for it in page_content_items(pg):
            it.pop("figure", None)
            items.append(it)
    return items


def text_layer_page(doc: "reader.PdfDocument", i: int, document_id: str, include_invisible: bool = False) -> PageResult:
    """A born-digital page as a PageResult: the native reader's positioned text lines put into reading order by the
    same layout analysis the OCR branch uses (the reference's ``pipe_txt_mode``, ocr_service.py:219-220)."""
    pi = doc.page_info(i)
    pr = PageResult(document_id, i, int(round(pi.width_pt)), int(round(pi.height_pt)))
    recs = doc.page_lines(i, include_invisible=include_invisible)
    if not recs:
        return pr
    boxes = np.asarray([r[0] for r in recs], dtype=np.float32).reshape(-1, 4)
    order, block, kind = layout.order_page_native(boxes, float(pi.height_pt))
    for k, j in enumerate(order):
        box, _size, text = recs[int(j)]
        pr.lines.append(TextLine(tuple(float(v) for v in box), text, 1.0, int(block[k]), 0.0,
                                 "cell" if kind[k] == 2 else "text"))
    return pr


LOW_CONFIDENCE = 0.90        # median line confidence of a page below which its text is flagged
LOW_CONFIDENCE_MIN_LINES = 5


def low_confidence_warning(page_number: int, confidences) -> Optional[str]:
    """The recogniser reads Latin, Cyrillic and Greek.  A page in another script (or one too degraded to read) still
    comes back as text - letters that look a little like the glyphs - but the recogniser is unsure of every line of
    it: a page in a known script reads at a median line confidence of 0.98 - 0.99, Hebrew, Georgian or Armenian at
    0.69 - 0.81.  The reference stack picks its recogniser by detected language (services/ocr_service/
    download_models_hf.py:41); here such a page is flagged in the document's ``warnings`` instead of passing as read."""
    c = np.asarray(confidences, dtype=np.float32).reshape(-1)
    if c.size < LOW_CONFIDENCE_MIN_LINES:
        return None
    med = float(np.median(c))
    if med >= LOW_CONFIDENCE:
        return None
    return (f"page {page_number}: text read with low confidence (median {med:.2f} over {c.size} lines): the page may be in a "
            "script other than Latin, Cyrillic or Greek, or too degraded to read")


def assemble_document(document_id: str, mode: str, pages: Sequence[PageResult], t0: Optional[float] = None,
                      image_writer=None) -> "DocumentResult":
    """Pages -> content list, markdown and combined text (ocr_service.py:223-248).  Figure crops are handed to the
    image writer here (``LocalFileWriter.write(path, data)``, ocr_service.py:111-124, 216)."""
    res = DocumentResult(document_id, mode)
    res.pages = sorted(pages, key=lambda p: p.page_index)
    res.warnings = [f"page {p.page_index + 1}: {p.error}" for p in res.pages if getattr(p, "error", None)]
    for p in res.pages:
        w = low_confidence_warning(p.page_index + 1, [l.confidence for l in p.lines if l.kind in ("text", "cell")])
        if w:
            res.warnings.append(w)
    if image_writer is not None:
        write_figures(res.pages, document_id, image_writer)
    res.content_list, res.discarded = drop_running_lines(build_content_list(res.pages), len(res.pages))
    res.markdown = content_markdown(res.content_list)
    res.text = combine_text(res.content_list, res.markdown)
    from ..utils.langid import detect

    res.language, res.language_confidence = detect(res.text)
    if t0 is not None:
        res.timings = {"total_s": time.perf_counter() - t0}
    return res


def write_figures(pages: Sequence[PageResult], document_id: str, image_writer) -> int:
    """Store every figure crop through the image writer as ``<document>/p<page>_f<k>.png``; returns the count."""
    import io

    from PIL import Image

    n = 0
    safe = _safe_name(document_id)
    for pg in pages:
        for k, fg in enumerate(pg.figures):
            if fg.image is None:
                continue
            buf = io.BytesIO()
            Image.fromarray(fg.image).save(buf, format="PNG", compress_level=1)
            fg.path = f"{safe}/p{pg.page_index}_f{k}.png"
            image_writer.write(fg.path, buf.getvalue())
            fg.image = None
            n += 1
    return n


def _safe_name(document_id: str) -> str:
    return re.sub(r"[^A-Za-z0-9_.-]", "_", document_id or "doc")[:80]


def assemble_packed(document_id: str, mode: str, shards: Sequence, extra: Dict[int, PageResult], n_pages: int,
                    t0: Optional[float] = None, image_writer=None) -> "DocumentResult":
    """``assemble_document`` for results that came back from GPU workers in packed form (runtime/wire.py): the content
    list and the combined text are built from the item arrays the workers prepared, figure crops arrive as PNG bytes,
    and ``pages`` materialises ``PageResult`` objects only when somebody reads them.  ``extra``: pages answered in
    this process (text-layer fallback of pages without an image)."""
    from .wire import LazyPages

    res = DocumentResult(document_id, mode)
    where: Dict[int, tuple] = {}
    for sh in shards:
        for k, i in enumerate(sh.page_indices()):
            if sh.present(k):
                where[i] = (sh, k)
    safe = _safe_name(document_id)
    slots, content = [], []
    for i in sorted(set(where) | set(extra)):
        hit = where.get(i)
        if hit is None:
            pg = extra[i]
            slots.append((None, -1, pg))
            for it in page_content_items(pg):
                it.pop("figure", None)
                content.append(it)
            continue
        sh, k = hit
        err = sh.page_error(k)
        if err:
            res.warnings.append(f"page {i + 1}: {err}")
        w = low_confidence_warning(i + 1, sh.page_confidences(k))
        if w:
            res.warnings.append(w)
        paths: Dict[int, str] = {}
        f0 = int(sh.fig_off[k])
        for f in range(f0, int(sh.fig_off[k + 1])):
            png = sh.figure_png(f)
            if image_writer is not None and png:
                paths[f] = f"{safe}/p{i}_f{f - f0}.png"
                image_writer.write(paths[f], png)
        slots.append((sh, k, None))
        content += sh.items(k, i, paths)
    res.pages = LazyPages(slots, document_id)
    content, res.discarded = drop_running_lines(content, len(slots))
    res.content_list = content
    res.markdown = content_markdown(content)
    res.text = combine_text(content, res.markdown)
    from ..utils.langid import detect

    res.language, res.language_confidence = detect(res.text)
    if t0 is not None:
        res.timings = {"total_s": time.perf_counter() - t0}
    return res


class DocumentPipeline:
    def __init__(self, engine: OcrEngine, decode_threads: Optional[int] = None, prefetch_batches: int = 2,
                 ocr_dpi: int = 300, image_writer=None):
        self.engine = engine
        self.image_writer = image_writer        # ``write(path, data)``: receives figure crops (store.results.LocalFileWriter)
        # sized from the cores this process is allowed on: a GPU worker pinned to its device's share of the host
        # (utils/affinity.py) gets that share, an unpinned single engine gets the machine
        n = decode_threads or min(16, max(2, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)))
        self.pool = ThreadPoolExecutor(max_workers=n, thread_name_prefix="dlamd-decode")
        self.prefetch = prefetch_batches
        self._pin = bool(engine.on_gpu and getattr(engine, "device", None) is not None)   # a stub engine has no device
        self.timer = StageTimer()
        self.ocr_dpi = ocr_dpi

    def close(self) -> None:
        self.pool.shutdown(wait=False, cancel_futures=True)

    # ------------------------------------------------------------------ page sources
    def _load_page(self, doc: reader.PdfDocument, i: int, doc_id: str, prior) -> Optional[PageImage]:
        """Runs on the decode pool.  Returns None for pages without an image (text layer is used instead)."""
        im = doc.main_image(i)
        if im is None:
            return None
        page = self._load_raster(doc, im, i, doc_id, prior)
        rot = int(doc.page_info(i).rotate) % 360
        if rot in (90, 180, 270):
            page.rotation = rot
        return page

    def _load_raster(self, doc: reader.PdfDocument, im, i: int, doc_id: str, prior) -> PageImage:
        if im.codec == reader.CODEC_JPEG:
            stream = doc.jpeg_bytes(im)
            scan = reader.scan_jpeg(stream)
            if scan.gpu_decodable and self.engine.on_gpu:
                return PageImage(None, doc_id, i, prior, encoded=(stream, scan))
            arr = _to_gray(jpeg_ops.decode_jpeg_host(stream))
        elif im.codec == reader.CODEC_RAW:
            shape = (im.height, im.width) if im.out_channels == 1 else (im.height, im.width, 3)
            if self.engine.on_gpu:
                if im.bpc == 1 and im.out_channels == 1 and getattr(self.engine, "takes_bits", False):
                    # bilevel page: upload the packed bits (1.1 MB for A4 at 300 DPI), expand on the GPU
                    tb = torch.empty((im.height, (im.width + 7) // 8), dtype=torch.uint8, pin_memory=self._pin)
                    if doc.decode_image_bits(im, tb.numpy()) is not None:
                        return PageImage(None, doc_id, i, prior, bits=(tb, im.width))
                t = torch.empty(shape, dtype=torch.uint8, pin_memory=self._pin)
                doc.decode_image(im, t.numpy())
                return PageImage(t, doc_id, i, prior)
            arr = doc.decode_image(im)
        else:
            raise ExtractionError(f"page {i}: unsupported image encoding")
        t = torch.from_numpy(arr)
        if self._pin:
            t = t.pin_memory()
        return PageImage(t, doc_id, i, prior)

    def _page_batches(self, doc, page_ids, doc_id, priors, skipped: List[int],
                      track: Optional[List[Future]] = None) -> Iterator[List[PageImage]]:
        chunks = [page_ids[a:b] for a, b in balanced_chunks(len(page_ids), self.engine.cfg.max_batch_pages)]
        futs: List[List[Future]] = []

        def submit(c):
            fs = [self.pool.submit(self._load_page, doc, i, doc_id, None if priors is None else priors[i]) for i in c]
            if track is not None:
                track.extend(fs)
            return fs

        nxt = 0
        while nxt < len(chunks) and nxt < self.prefetch:
            futs.append(submit(chunks[nxt]))
            nxt += 1
        for k in range(len(chunks)):
            t0 = time.perf_counter()
            pages = []
            for i, f in zip(chunks[k], futs[k]):
                p = f.result()
                if p is None:
                    skipped.append(i)
                else:
                    pages.append(p)
            self.timer.add("wait_decode", time.perf_counter() - t0)
            if nxt < len(chunks):
                futs.append(submit(chunks[nxt]))
                nxt += 1
            yield pages

    # ------------------------------------------------------------------ entry points
    def process_pdf(self, pdf_bytes: bytes, document_id: str = "", layout_priors=None,
                    force_mode: Optional[str] = None) -> DocumentResult:
        """One document, nothing else in flight.  Raises ``ExtractionError`` when it cannot be read."""
        res = self.process_pdfs([(pdf_bytes, document_id, layout_priors)], force_mode=force_mode)[0]
        if res.error is not None:
            raise ExtractionError(res.error)
        return res

    def process_pdfs(self, docs: Sequence[Tuple[bytes, str, Optional[Sequence]]],
                     force_mode: Optional[str] = None) -> List[DocumentResult]:
        """Several documents back to back through ONE engine pipeline: ``docs`` = [(pdf_bytes, document_id,
        layout_priors or None)].  Page batches never mix documents, but the two-stream pipeline is not drained at a
        document boundary (detection of the next document's first batch overlaps recognition of the previous
        document's last one) - what a worker with a queue of uploads does.  Results come back in input order; a
        document that cannot be parsed or has no pages yields a result whose ``error`` is set and does not disturb
        the others."""
        states: List[dict] = []
        loads: List[Future] = []          # every page load handed to the decode pool (they run inside the native reader)
        try:
            def batches() -> Iterator[List[PageImage]]:
                for k, (pdf_bytes, did, pri) in enumerate(docs):
                    t0 = time.perf_counter()
                    st = {"doc": None, "id": did, "mode": "ocr", "n": 0, "skipped": [], "pages": [], "error": None, "t0": t0}
                    states.append(st)
                    try:
                        doc = reader.PdfDocument(pdf_bytes)
                    except (reader.PdfError, TypeError) as e:
                        st["error"] = f"cannot parse PDF: {e}"
                        continue
                    st["doc"] = doc
                    st["n"] = doc.page_count
                    if st["n"] == 0:
                        st["error"] = "PDF has no pages"
                        continue
                    st["mode"] = force_mode or ("txt" if doc.classify().is_text else "ocr")
                    self.timer.add("parse", time.perf_counter() - t0)
                    if st["mode"] == "ocr":
                        yield from self._page_batches(doc, list(range(st["n"])), f"{k}\x00{did}", pri, st["skipped"], loads)

            for batch_res in self.engine.process_batches(batches()):
                for pr in batch_res:
                    k, _, did = pr.doc_id.partition("\x00")
                    pr.doc_id = did
                    states[int(k)]["pages"].append(pr)
            out: List[DocumentResult] = []
            for st in states:
                if st["error"] is not None:
                    res = DocumentResult(st["id"], st["mode"])
                    res.error = st["error"]
                    out.append(res)
                    continue
                doc = st["doc"]
                if st["mode"] == "txt":
                    pages = [text_layer_page(doc, i, st["id"]) for i in range(st["n"])]
                else:
                    # pages without any image: fall back to whatever text layer they have
                    pages = st["pages"] + [text_layer_page(doc, i, st["id"], include_invisible=True) for i in st["skipped"]]
                res = assemble_document(st["id"], st["mode"], pages, st["t0"], self.image_writer)
                out.append(res)
            return out
        finally:
            # after a failure some page loads may still be running: no handle is freed under a thread that uses it
            for f in loads:
                f.cancel()
            from concurrent.futures import wait as _wait

            _wait([f for f in loads if not f.cancelled()], timeout=60.0)
            for st in states:
                if st["doc"] is not None:
                    st["doc"].close()

    def process_rasters(self, images: Sequence[torch.Tensor], document_id: str = "", layout_priors=None
                        ) -> DocumentResult:
        """Already-decoded page rasters (uint8 tensors) -> DocumentResult (OCR branch only)."""
        pages_in = [PageImage(img, document_id, i, None if layout_priors is None else layout_priors[i])
                    for i, img in enumerate(images)]
        return assemble_document(document_id, "ocr", self.engine.process_pages(pages_in), None, self.image_writer)
