"""Single-GPU OCR engine: page images -> text lines.

This is the MI355X replacement for what the reference delegates to magic-pdf's
CPU models (services/ocr_service/ocr_service.py:215-216: ``doc_analyze`` +
``pipe_ocr_mode``).  One engine instance owns one GPU:

    u8 pages --H2D--> resample --> DBNet --> DB post-process
        --boxes D2H--> crop table (host, reading order + width buckets)
        --> crop/resize --> CRNN --> CTC greedy decode --ids D2H--> strings

Design points
* Two batches are always in flight on two HIP streams (software pipeline run
  from one host thread): while the host turns batch i's boxes into a crop
  table, the GPU is already running the detector of batch i+1.
* Everything between the two D2H points is asynchronous; device workspaces are
  cached per shape so the steady state does not allocate.
* Recognition batches are grouped by crop width bucket; small buckets replay a
  captured hipGraph (``GraphCache``), large ones launch eagerly (13 launches
  against tens of ms of MFMA work).
* ``layout_prior``: with random-init weights the
  detector emits no meaningful boxes, so a caller may attach the page's known
  line boxes; they are used ONLY when the detector itself found nothing usable
  for that page.  Detector and post-process always run in full.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ..models.crnn import CRNN
from ..models.dbnet import DBNet
from ..ops import conv as conv_ops
from ..ops import dbpost
from ..ops import jpeg as jpeg_ops
from ..ops import lstm as lstm_ops
from ..ops import resample as rs
from ..utils.scripts import line_script, resolve_scripts, script_counts
from . import layout


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


@dataclass
class PageImage:
    """One page raster.

    One of ``image`` (uint8 [H, W] or [H, W, 3] tensor, host - ideally pinned - or device),
    ``encoded`` = (JPEG stream bytes as uint8 ndarray, its ``JpegScan``) for the GPU JPEG decoder, or
    ``bits`` = (packed bilevel rows as uint8 [H, (W + 7) // 8] tensor, most significant bit first, set bit = white;
    width in pixels), expanded on the GPU."""

    image: Optional[torch.Tensor] = None
    doc_id: str = ""
    page_index: int = 0
    layout_prior: Optional[np.ndarray] = None  # [n, 4] x0,y0,x1,y1 in page pixels (see module docstring)
    encoded: Optional[tuple] = None
    rotation: int = 0          # the page's /Rotate: degrees clockwise the stored raster must be turned to read upright
    bits: Optional[tuple] = None

    @property
    def raster_hw(self) -> Tuple[int, int]:
        if self.image is not None:
            return int(self.image.shape[0]), int(self.image.shape[1])
        if self.bits is not None:
            return int(self.bits[0].shape[0]), int(self.bits[1])
        info = self.encoded[1].info
        return int(info.height), int(info.width)

    @property
    def hw(self) -> Tuple[int, int]:
        """Upright page size (what detection boxes refer to)."""
        h, w = self.raster_hw
        return (w, h) if self.rotation in (90, 270) else (h, w)


@dataclass
class TextLine:
    box: Tuple[float, float, float, float]
    text: str
    confidence: float
    block: int = 0               # text block (leaf region of the layout analysis); paragraphs never span blocks
    angle: float = 0.0           # radians the line is turned about the centre of ``box`` (skewed scans), 0 = upright
    kind: str = "text"           # "equation": a display formula, ``text`` is its LaTeX source; "cell": a cell of a
                                 # table (the cells of one table share ``block`` and come row by row)
    formula_confidence: float = 1.0     # lowest mean token probability among the formulas inside the line ($...$)


@dataclass
class Figure:
    box: Tuple[float, float, float, float]
    order: int = 0               # position in the page's reading sequence (index among lines + figures)
    image: Optional[np.ndarray] = None     # uint8 crop of the page raster, present until the image writer took it
    path: str = ""               # where the image writer stored the crop


@dataclass
class PageResult:
    doc_id: str
    page_index: int
    width: int
    height: int
    lines: List[TextLine] = field(default_factory=list)
    figures: List[Figure] = field(default_factory=list)
    n_detected: int = 0          # boxes the detector itself produced
    used_layout_prior: bool = False
    error: Optional[str] = None  # the page image could not be decoded completely: no text is reported for it

    def text(self) -> str:
        return "\n".join(l.text for l in self.lines if l.text)


def _apply_page_script(lines: List[TextLine]) -> None:
    """Lines made only of letters that Latin shares with Cyrillic / Greek ("О НАС", "ТЕМА", a lone "а") cannot tell
    their alphabet themselves: on a page where at least two lines, and most of the lines that can tell, are Cyrillic
    (Greek), they are read as that alphabet too."""
    votes = {"cyrillic": 0, "greek": 0, "latin": 0}
    for ln in lines:
        if ln.kind == "equation" or not ln.text:
            continue
        cyr, grk, lat = script_counts(ln.text)
        if cyr > max(grk, lat):
            votes["cyrillic"] += 1
        elif grk > max(cyr, lat):
            votes["greek"] += 1
        elif lat:                                  # letters only Latin has; a line of shared shapes does not vote
            votes["latin"] += 1
    page = max(("cyrillic", "greek"), key=lambda k: votes[k])
    if votes[page] < 2 or votes[page] <= votes["latin"]:
        return
    for ln in lines:
        if ln.kind != "equation" and ln.text:
            ln.text = resolve_scripts(ln.text, hint=page)


@dataclass
class EngineConfig:
    det_limit_side: int = 1344
    rec_height: int = 48
    rec_width_quantum: int = 256
    rec_max_width: int = 2560
    rec_max_batch: int = 2048
    max_batch_pages: int = 36
    max_lines_per_page: int = 1024
    min_line_px: int = 6
    vertical_ratio: float = 1.5
    use_hip_graphs: bool = True
    graph_max_crops: int = 128
    pipeline_slots: int = 2           # HIP streams the batches rotate over (det of k+1 overlaps rec of k, ...)
    trocr_batch: int = 256            # crops per encoder / decoder pass of the transformer recognizer
    trocr_max_tokens: int = 64        # greedy decode length (the decoder runs a fixed number of steps)
    db: dbpost.DBPostConfig = field(default_factory=dbpost.DBPostConfig)
    detect_figures: bool = True       # run the detector's block head and keep figure regions out of the text flow
    # figure regions: components of the 1/4-scale block mask, no unclip, upright boxes
    fig_db: dbpost.DBPostConfig = field(default_factory=lambda: dbpost.DBPostConfig(
        bin_thresh=0.5, box_thresh=0.5, unclip_ratio=0.0, min_size=6, max_components=1024, max_boxes=32, max_angle_deg=0.0))
    fig_min_side_px: int = 90         # smaller regions are not figures (page pixels)
    max_figures_per_page: int = 8
    fig_max_page_share: float = 0.4   # a "figure" that covers more of the page than this is a tinted page, not a figure
    fig_max_lines: int = 8            # ... and one that holds this many text lines is a text region
    detect_formulas: bool = True      # second channel of the block head -> formula boxes -> formula recogniser
    # formula boxes inside a text line: the line is read in pieces, ``... $x_{i}$ ...``.  Off by default: it needs the
    # detector trained for it (``dbnet_inline.safetensors``, chosen by ``runtime.workers.build_engine`` when this is set),
    # which finds four formulas in five and takes one ordinary line in 170 for one (bench/inline_probe.py)
    inline_formulas: bool = False
    inline_min_score: float = 0.5     # mean probability of the formula map over the box, below which it cuts nothing
    inline_min_conf: float = 0.9      # mean token probability of a formula in a line, below which the line is read whole
    # "x = 3", "(a)", "12" are read as well by the text recogniser, and the formula map fires on such tokens of ordinary
    # prose: a formula inside a line counts when its LaTeX has a command, a subscript or a superscript
    inline_needs_markup: bool = True
    # line heights inside the detected end of a formula around which its true end is searched (``_refine_spans``; the
    # map overshoots by 0.3 - 0.6: of 266 formulas 140 / 148 / 145 / 130 / 108 are read exactly at 0.15 / 0.25 / 0.35 / 0.5 / 0.65)
    inline_inset: float = 0.25
    # a display formula is 60-100 page pixels tall = 6-10 pixels of the 1/4-scale block map: smaller size floor
    frm_db: dbpost.DBPostConfig = field(default_factory=lambda: dbpost.DBPostConfig(
        bin_thresh=0.5, box_thresh=0.5, unclip_ratio=0.0, min_size=2, max_components=1024, max_boxes=64, max_angle_deg=0.0))
    # (with ``inline_formulas`` the engine lowers min_size to 1: the band that marks a formula inside a text line is three
    # map pixels tall; without it such specks - the "9" of "Y<9" - are no formulas)
    formula_min_side_px: int = 16
    formula_max_tokens: int = 96
    max_formulas_per_page: int = 64


class StageTimer:
    """Per-stage wall-clock accumulators."""

    def __init__(self) -> None:
        self.t: Dict[str, float] = {}
        self.n: Dict[str, int] = {}

    def add(self, name: str, dt: float, count: int = 1) -> None:
        self.t[name] = self.t.get(name, 0.0) + dt
        self.n[name] = self.n.get(name, 0) + count

    def summary(self) -> Dict[str, float]:
        return {k: round(v, 6) for k, v in self.t.items()}


class _DetBatch:
    __slots__ = ("pages", "srcs", "boxes_host", "nbox_host", "fig_host", "nfig_host", "frm_host", "nfrm_host", "event", "stream",
                 "slot", "keep", "jpeg_counts", "jpeg_rows", "page_errors")


class _RecBatch:
    __slots__ = ("det", "groups", "event", "results", "keep", "fig_crops", "formula_out")


def deskewed_boxes(b4: np.ndarray, angles: np.ndarray, page_hw: Tuple[int, int]) -> Tuple[np.ndarray, float]:
    """Boxes of a skewed page in the frame of the upright page: every centre is turned back about the page centre by
    the page's skew (median angle of its slanted lines), sizes stay.  Layout analysis cuts along whitespace that is
    only straight in that frame.  Returns (boxes, skew in radians)."""
    sl = angles[angles != 0.0]
    if sl.size < 3 or sl.size < 0.3 * angles.size:
        return b4, 0.0
    a = float(np.median(sl))
    if abs(a) < 0.004:
        return b4, 0.0
    H, W = page_hw
    cx, cy = 0.5 * (b4[:, 0] + b4[:, 2]) - 0.5 * W, 0.5 * (b4[:, 1] + b4[:, 3]) - 0.5 * H
    c, s_ = np.cos(-a), np.sin(-a)
    rx, ry = cx * c - cy * s_ + 0.5 * W, cx * s_ + cy * c + 0.5 * H
    hw, hh = 0.5 * (b4[:, 2] - b4[:, 0]), 0.5 * (b4[:, 3] - b4[:, 1])
    return np.stack([rx - hw, ry - hh, rx + hw, ry + hh], 1).astype(np.float32), a


KIND_TEXT, KIND_FORMULA, KIND_CELL, KIND_TEXT_CONT, KIND_MATH_CONT, KIND_MATH_HEAD, KIND_LINE_ALT, KIND_MATH_TEXT = range(8)
_MATH_KINDS = (KIND_FORMULA, KIND_MATH_CONT, KIND_MATH_HEAD)


def assign_inline_formulas(lines: np.ndarray, q: np.ndarray, min_score: float = 0.0) -> Tuple[dict, np.ndarray]:
    """Which formula boxes sit INSIDE a text line.  ``lines`` [n, >= 7]: x0, y0, x1, y1, score, block, angle of the line
    rectangles (turned by ``angle`` about their centre; rows of NaN are skipped), ``q`` [m, >= 4] upright formula boxes.
    A formula box whose centre lies in a line's rectangle, with at least one line height of that line left over beside
    it, is part of the line (the text detector fires on the pieces of a display formula too, but those pieces end where
    the formula ends).  So is a box on a line's baseline just beyond its end, up to 1.2 line heights away: a line that
    begins or ends with a formula, which the text detector tends to leave out.  A box inside a line whose mean
    probability is below ``min_score`` is set aside too (it is no display formula) but cuts nothing out of its line.  -> ({line index: [(a0, a1, score)] along the line's axis
    relative to its centre, sorted}, mask over ``q``)."""
    taken = np.zeros((q.shape[0],), bool)
    spans: dict = {}
    if lines.shape[0] == 0 or q.shape[0] == 0:
        return spans, taken
    cx, cy = 0.5 * (lines[:, 0] + lines[:, 2]), 0.5 * (lines[:, 1] + lines[:, 3])
    w, h = lines[:, 2] - lines[:, 0], lines[:, 3] - lines[:, 1]
    th = lines[:, 6] if lines.shape[1] > 6 else np.zeros_like(cx)
    c, sn = np.cos(th), np.sin(th)
    with np.errstate(invalid="ignore"):
        for j in range(q.shape[0]):
            qx, qy = 0.5 * (q[j, 0] + q[j, 2]), 0.5 * (q[j, 1] + q[j, 3])
            qw, qh = q[j, 2] - q[j, 0], q[j, 3] - q[j, 1]
            dx, dy = qx - cx, qy - cy
            along, across = dx * c + dy * sn, -dx * sn + dy * c
            ok = ((np.abs(along) <= 0.5 * w) & (np.abs(across) <= 0.5 * h) & (w >= qw + 1.0 * h) & (qh <= 1.3 * h) &
                  (w >= 2.0 * h))
            if ok.any():
                i = int(np.argmin(np.where(ok, np.abs(across), np.inf)))
                a0, a1 = max(-0.5 * w[i], along[i] - 0.5 * qw), min(0.5 * w[i], along[i] + 0.5 * qw)
            else:
                # a line that begins or ends with a formula: the text detector tends to stop where the words stop, and
                # the formula lies just beyond the line's end, on its baseline
                gap = np.abs(along) - 0.5 * w - 0.5 * qw
                # (beyond it: what the text detector finds INSIDE a display formula, its pieces, has a large negative gap)
                near = ((gap <= 1.2 * h) & (gap >= -0.5 * h) & (np.abs(across) <= 0.35 * h) & (qh <= 1.5 * h) &
                        (w >= 1.5 * h) & (qw <= 12.0 * h))
                if not near.any():
                    continue
                i = int(np.argmin(np.where(near, gap, np.inf)))
                a0, a1 = along[i] - 0.5 * qw, along[i] + 0.5 * qw
            if a1 - a0 < 4.0:
                continue
            taken[j] = True
            if q.shape[1] > 4 and q[j, 4] < min_score:
                continue
            spans.setdefault(i, []).append((float(a0), float(a1), float(q[j, 4]) if q.shape[1] > 4 else 1.0))
    for i in spans:
        spans[i].sort()
    return spans, taken


def split_line_at_formulas(row: np.ndarray, spans, min_text: float = 0.45) -> np.ndarray:
    """One line row (x0, y0, x1, y1, score, block, angle, kind) -> its pieces left to right as rows of the same form:
    stretches of text (kind 0 for the first piece, 3 after it) and formulas (5 / 4).  A stretch of text shorter than
    ``min_text`` line heights is a sliver between a formula and the line's end, not text.  Every formula piece is
    followed by the same box as kind 7 (read by the text recogniser: what the piece falls back to when the formula
    reading does not count), and the last row (kind 6) is the whole line once more, read as plain text: what the line
    falls back to when none of its formulas counts (``OcrEngine._join_pieces``)."""
    cx, cy = 0.5 * (row[0] + row[2]), 0.5 * (row[1] + row[3])
    w, h = row[2] - row[0], row[3] - row[1]
    c, sn = np.cos(row[6]), np.sin(row[6])
    parts = []
    # (a formula beyond the line's end - the line begins or ends with it - makes the line that much longer)
    lo = min([-0.5 * w] + [a0 for a0, _, _ in spans])
    hi = max([0.5 * w] + [a1 for _, a1, _ in spans])
    cur = lo
    for a0, a1, sc in spans:
        a0 = max(a0, cur)
        if a1 - a0 < 4.0:
            continue
        if a0 - cur >= min_text * h:
            parts.append((cur, a0, False, row[4]))
        parts.append((a0, a1, True, sc))
        cur = a1
    if hi - cur >= min_text * h:
        parts.append((cur, hi, False, row[4]))
    out = []
    for k, (u, v, is_math, sc) in enumerate(parts):
        m, half = 0.5 * (u + v), 0.5 * (v - u)
        px, py = cx + m * c, cy + m * sn
        piece = row.copy()
        piece[:5] = (px - half, py - 0.5 * h, px + half, py + 0.5 * h, sc)
        if is_math:
            piece[7] = KIND_MATH_HEAD if k == 0 else KIND_MATH_CONT
            out.append(piece)
            piece = piece.copy()                     # the same crop for the text recogniser
            piece[7] = KIND_MATH_TEXT
        else:
            piece[7] = KIND_TEXT if k == 0 else KIND_TEXT_CONT
        out.append(piece)
    whole = row.copy()
    m, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    whole[:4] = (cx + m * c - half, cy + m * sn - 0.5 * h, cx + m * c + half, cy + m * sn + 0.5 * h)
    whole[7] = KIND_LINE_ALT
    out.append(whole)
    return np.asarray(out, np.float32).reshape(-1, row.shape[0])


def widest_gap(profile: np.ndarray, centre: int, min_run: int = 3) -> Optional[int]:
    """Column of the middle of the widest run of paper in ``profile`` (darkest sample per pixel column, 0 = ink),
    ``centre`` breaking ties towards the estimate.  None when no run of ``min_run`` columns is blank."""
    if profile.size == 0:
        return None
    lo, hi = float(profile.min()), float(profile.max())
    if hi - lo < 24.0:                      # nothing but paper (or nothing but ink): keep the estimate
        return None
    blank = profile >= hi - 0.3 * (hi - lo)
    best, best_key = None, None
    k, n = 0, blank.size
    while k < n:
        if not blank[k]:
            k += 1
            continue
        e = k
        while e < n and blank[e]:
            e += 1
        if e - k >= min_run:
            mid = (k + e - 1) // 2
            key = (e - k) - 0.15 * abs(mid - centre)
            if best is None or key > best_key:
                best, best_key = mid, key
        k = e
    return best


def balanced_chunks(n: int, limit: int) -> List[Tuple[int, int]]:
    """[start, end) ranges that cut ``n`` items into the fewest batches of at most ``limit``, all of (nearly) the same
    size: 100 pages at a limit of 36 are 34 + 33 + 33, not 36 + 36 + 28.  A short tail batch runs every layer of both
    models at a worse tile fill for the same launch count (measured on 100-page scans: 25 x 4 890 pages/s, 40 + 40 +
    20 845, 34 + 33 + 33 920)."""
    if n <= 0:
        return []
    nb = -(-n // max(1, limit))
    base, extra = divmod(n, nb)
    out, a = [], 0
    for k in range(nb):
        b = a + base + (1 if k < extra else 0)
        out.append((a, b))
        a = b
    return out


class OcrEngine:
    def __init__(self, device: torch.device | str = "cuda:0", det: Optional[DBNet] = None, rec: Optional[CRNN] = None,
                 cfg: Optional[EngineConfig] = None, seed: int = 0, formula_rec=None):   # rec: CRNN or TrOCR
        self.device = torch.device(device)
        self.cfg = cfg or EngineConfig()
        if self.cfg.inline_formulas and self.cfg.frm_db.min_size > 1:
            from dataclasses import replace

            self.cfg.frm_db = replace(self.cfg.frm_db, min_size=1)
        self.on_gpu = self.device.type == "cuda"
        if det is None or rec is None:
            g = torch.random.get_rng_state()
            torch.manual_seed(seed)
            det = det or DBNet()
            rec = rec or CRNN()
            torch.random.set_rng_state(g)
        self.det = det.to(self.device).prepare()
        self.rec = rec.to(self.device).prepare()
        # OCR_RECOGNIZER=trocr: ViT encoder + transformer decoder instead of the CRNN
        self.rec_kind = "crnn" if isinstance(rec, CRNN) else "trocr"
        self._trocr_event = None      # the decoder's static graph state is shared by both pipeline slots
        self._formula_event = None
        # formula recogniser (models/trocr.py at formula_config() size): without one, formula regions are reported
        # as equations with empty LaTeX
        self.formula_rec = formula_rec.to(self.device).prepare() if formula_rec is not None else None
        self.timer = StageTimer()
        self._ws: Dict[Tuple[int, int, int], dbpost.DBPostWorkspace] = {}
        ns = max(2, int(self.cfg.pipeline_slots))
        self._streams = [torch.cuda.Stream(self.device) for _ in range(ns)] if self.on_gpu else [None] * ns
        self._graphs: Dict[Tuple[int, int, int], "_RecGraph"] = {}
        cs = getattr(self.rec, "charset", "")
        self._lut = None
        self._bit_lut = None
        self.takes_bits = self.on_gpu          # bilevel pages may arrive as packed bits (PageImage.bits)
        if cs and all(ord(c) < 0x10000 and not 0xD800 <= ord(c) < 0xE000 for c in cs):
            # class id -> code point (id 0 = CTC blank): strings are built by one gather + one UTF-32 decode per line
            self._lut = np.frombuffer(("\0" + cs).encode("utf-32-le"), dtype=np.uint32)
        self.pages_done = 0
        self.lines_done = 0

    # ------------------------------------------------------------------ geometry
    def det_geometry(self, H: int, W: int) -> Tuple[int, int, int, int]:
        """-> (out_h, out_w, Hd, Wd): resized content size and its /32 bucket."""
        lim = self.cfg.det_limit_side
        s = min(1.0, lim / float(max(H, W)))
        oh, ow = max(32, int(round(H * s))), max(32, int(round(W * s)))
        return oh, ow, _round_up(oh, 32), _round_up(ow, 32)

    # ------------------------------------------------------------------ detection stage
    def _submit_det(self, pages: Sequence[PageImage], slot: int) -> _DetBatch:
        cfg = self.cfg
        st = self._streams[slot]
        db = _DetBatch()
        db.pages = list(pages)
        db.stream = st
        db.slot = slot
        db.jpeg_counts, db.jpeg_rows, db.page_errors = None, [], {}
        B = len(pages)
        geo = [self.det_geometry(*p.hw) for p in pages]
        Hd = max(g[2] for g in geo)
        Wd = max(g[3] for g in geo)
        ctx = torch.cuda.stream(st) if self.on_gpu else _null_ctx()
        with ctx:
            srcs: List[Optional[torch.Tensor]] = [None] * B
            for i, p in enumerate(pages):
                if p.bits is not None:
                    # bilevel page: 8 pixels per uploaded byte, expanded by one table look-up on the device
                    packed = p.bits[0].to(self.device, non_blocking=True)
                    if self._bit_lut is None:
                        v = torch.arange(256, dtype=torch.int32).view(256, 1) >> torch.arange(7, -1, -1, dtype=torch.int32).view(1, 8)
                        self._bit_lut = ((v & 1) * 255).to(torch.uint8).to(self.device)
                    img = self._bit_lut[packed.view(-1).long()].view(packed.shape[0], packed.shape[1] * 8)
                    srcs[i] = img if img.shape[1] == p.bits[1] else img[:, : p.bits[1]].contiguous()
            enc = [i for i, p in enumerate(pages) if p.image is None and p.bits is None]
            if enc:
                if self.on_gpu:
                    dec = jpeg_ops.decode_jpeg_batch_gpu([pages[i].encoded[0] for i in enc],
                                                         [pages[i].encoded[1] for i in enc], self.device)
                else:
                    dec = [torch.from_numpy(_to_gray(jpeg_ops.decode_jpeg_host(pages[i].encoded[0]))) for i in enc]
                for i, t in zip(enc, dec):
                    srcs[i] = t
                if self.on_gpu:
                    # how many blocks each restart-less stream really decoded: read with the boxes, compared in _det_boxes
                    counted, db.jpeg_rows = jpeg_ops.blocks_short(srcs)
                    if counted is not None:
                        db.jpeg_counts = torch.empty(counted.shape, dtype=torch.int32, pin_memory=True)
                        db.jpeg_counts.copy_(counted, non_blocking=True)
            regions = []
            for i, (p, (oh, ow, _, _)) in enumerate(zip(pages, geo)):
                img = srcs[i]
                if img is None:
                    img = p.image
                    if img.device != self.device:
                        img = img.to(self.device, non_blocking=True)
                    srcs[i] = img
                if p.rotation:
                    # /Rotate: turn the decoded raster upright on the device once; crops then come from the same copy
                    img = torch.rot90(img, k=-(p.rotation // 90), dims=(0, 1)).contiguous()
                    srcs[i] = img
                regions.append(rs.Region(img, (0.0, 0.0, float(img.shape[1]), float(img.shape[0])), oh, ow))
            figs = cfg.detect_figures and hasattr(self.det, "fig_head")
            fprob = fmask = qprob = qmask = None
            frm = figs and cfg.detect_formulas
            det_infer = self.det.infer
            if self.on_gpu and hasattr(self.det, "stem1_packed") and rs.stem_serves(self.det.stem1_packed()[0], (2, 2), (1, 1)):
                # page resample fused with the detector's first convolution: the [B, Hd, Wd, 8] tensor is never written
                w1, b1, act1 = self.det.stem1_packed()
                x = rs.resample_stem_items_gpu(self._upload_items(regions), len(regions), Hd, Wd, rs.IMAGENET_MEAN,
                                               rs.IMAGENET_STD, w1, b1, act1)
                det_infer = self.det.infer_after_stem1
            else:
                x = self._resample(regions, Hd, Wd, rs.IMAGENET_MEAN, rs.IMAGENET_STD)
            if figs:
                prob, mask, fig = det_infer(x, with_figures=True)
                planes = conv_ops.block_head_split(fig.contiguous(), 2 if frm else 1)
                fprob, fmask = planes[0]
                if frm:
                    qprob, qmask = planes[1]
            else:
                prob, mask = det_infer(x)
            scale = torch.tensor([[p.hw[1] / g[1], p.hw[0] / g[0]] for p, g in zip(pages, geo)],
                                 dtype=torch.float32)
            valid = torch.tensor([[g[1], g[0]] for g in geo], dtype=torch.int32)
            if self.on_gpu:
                key = (B, Hd, Wd)
                ws = self._ws.get((slot,) + key)
                if ws is None:
                    ws = dbpost.DBPostWorkspace(B, Hd, Wd, cfg.db, self.device)
                    self._ws[(slot,) + key] = ws
                scale_d = scale.to(self.device, non_blocking=True)
                valid_d = valid.to(self.device, non_blocking=True)
                boxes, nbox = dbpost.db_postprocess_gpu(prob, mask, ws, scale_d, valid_d)
                db.boxes_host = torch.empty(boxes[:B].shape, dtype=boxes.dtype, pin_memory=True)
                db.nbox_host = torch.empty((B,), dtype=torch.int32, pin_memory=True)
                db.boxes_host.copy_(boxes[:B], non_blocking=True)
                db.nbox_host.copy_(nbox[:B], non_blocking=True)
                db.fig_host = db.nfig_host = db.frm_host = db.nfrm_host = None
                keep_f = None
                if figs:
                    fkey = ("fig", slot, B, Hd // 4, Wd // 4)
                    wf = self._ws.get(fkey)
                    if wf is None:
                        wf = dbpost.DBPostWorkspace(B, Hd // 4, Wd // 4, cfg.fig_db, self.device)
                        self._ws[fkey] = wf
                    fscale_d = (scale * 4.0).to(self.device, non_blocking=True)
                    fvalid_d = ((valid + 3) // 4).to(self.device, non_blocking=True)
                    fb, nf = dbpost.db_postprocess_gpu(fprob, fmask, wf, fscale_d, fvalid_d)
                    db.fig_host = torch.empty(fb[:B].shape, dtype=fb.dtype, pin_memory=True)
                    db.nfig_host = torch.empty((B,), dtype=torch.int32, pin_memory=True)
                    db.fig_host.copy_(fb[:B], non_blocking=True)
                    db.nfig_host.copy_(nf[:B], non_blocking=True)
                    keep_f = (fprob, fmask, fscale_d, fvalid_d)
                    if frm:
                        qkey = ("frm", slot, B, Hd // 4, Wd // 4)
                        wq = self._ws.get(qkey)
                        if wq is None:
                            wq = dbpost.DBPostWorkspace(B, Hd // 4, Wd // 4, cfg.frm_db, self.device)
                            self._ws[qkey] = wq
                        qb, nq = dbpost.db_postprocess_gpu(qprob, qmask, wq, fscale_d, fvalid_d)
                        db.frm_host = torch.empty(qb[:B].shape, dtype=qb.dtype, pin_memory=True)
                        db.nfrm_host = torch.empty((B,), dtype=torch.int32, pin_memory=True)
                        db.frm_host.copy_(qb[:B], non_blocking=True)
                        db.nfrm_host.copy_(nq[:B], non_blocking=True)
                        keep_f = keep_f + (qprob, qmask)
                db.event = torch.cuda.Event()
                db.event.record(st)
                db.keep = (x, prob, mask, scale_d, valid_d, keep_f)
            else:
                res = dbpost.db_postprocess_reference(prob, mask, cfg.db, scale, valid)
                db.boxes_host, db.nbox_host, db.event = res, None, None
                db.fig_host = db.nfig_host = db.frm_host = db.nfrm_host = None
                if figs:
                    db.fig_host = dbpost.db_postprocess_reference(fprob, fmask, cfg.fig_db, scale * 4.0, (valid + 3) // 4)
                if frm:
                    db.frm_host = dbpost.db_postprocess_reference(qprob, qmask, cfg.frm_db, scale * 4.0, (valid + 3) // 4)
                db.keep = None
            db.srcs = srcs
        return db

    def _upload_items(self, regions) -> torch.Tensor:
        items = rs.pack_items(regions)
        host = torch.from_numpy(items.view(np.uint8).reshape(-1)).pin_memory()
        return host.to(self.device, non_blocking=True)

    def _resample(self, regions, Hd, Wd, mean, std) -> torch.Tensor:
        if not self.on_gpu:
            return rs.resample_regions(regions, Hd, Wd, mean, std)
        return rs.resample_items_gpu(self._upload_items(regions), len(regions), Hd, Wd, mean, std)

    def _det_boxes(self, db: _DetBatch) -> List[Tuple[np.ndarray, np.ndarray]]:
        """Wait for the detector of this batch.  Per page: (lines [n, 8] = x0, y0, x1, y1, score, block, angle, kind
        (0 text line, 1 display formula, 2 table cell) in reading order, figures [m, 5] = x0, y0, x1, y1, position in the reading
        sequence)."""
        cfg = self.cfg
        nfig = None
        frms = None
        if db.event is not None:
            db.event.synchronize()
            if db.jpeg_counts is not None:
                got = db.jpeg_counts.numpy()
                for pos, row, want in db.jpeg_rows:
                    if int(got[row]) != want:
                        db.page_errors[pos] = (f"page image is damaged: {int(got[row])} of {want} JPEG blocks could be decoded; "
                                               "no text is reported for this page")
            nb = db.nbox_host.numpy()
            bx = db.boxes_host.numpy()
            res = [bx[b, : min(int(nb[b]), cfg.db.max_boxes)].copy() for b in range(len(db.pages))]
            figs = None
            if db.fig_host is not None:
                nfig, fx = db.nfig_host.numpy(), db.fig_host.numpy()
                figs = [fx[b, : min(int(nfig[b]), cfg.fig_db.max_boxes), :4].copy() for b in range(len(db.pages))]
            if db.frm_host is not None:
                nq, qx = db.nfrm_host.numpy(), db.frm_host.numpy()
                frms = [qx[b, : min(int(nq[b]), cfg.frm_db.max_boxes), :5].copy() for b in range(len(db.pages))]
        else:
            res = db.boxes_host
            figs = None if db.fig_host is None else [np.asarray(f, np.float32).reshape(-1, 8)[:, :4] for f in db.fig_host]
            frms = None if db.frm_host is None else [np.asarray(f, np.float32).reshape(-1, 8)[:, :5] for f in db.frm_host]
        out = []
        for pi, (r, p) in enumerate(zip(res, db.pages)):
            if pi in db.page_errors:
                # what the detector saw is noise from the point where the stream broke: report nothing rather than garbage
                out.append((np.zeros((0, 8), np.float32), np.zeros((0, 5), np.float32)))
                continue
            r = np.asarray(r, dtype=np.float32)
            r = r.reshape(-1, r.shape[1] if r.ndim == 2 and r.shape[1] else 8)
            if r.shape[1] < 8:
                r = np.concatenate([r, np.zeros((r.shape[0], 8 - r.shape[1]), np.float32)], 1)
            r[:, 7] = 0.0
            inline_q = None
            if frms is not None and frms[pi].shape[0]:
                q = frms[pi]
                q = q[np.minimum(q[:, 2] - q[:, 0], q[:, 3] - q[:, 1]) >= cfg.formula_min_side_px][: cfg.max_formulas_per_page]
                if q.shape[0] and cfg.inline_formulas:
                    # formulas inside a text line are set aside here (whatever their score: they are no display formulas)
                    # and cut out of their line after the layout analysis
                    _, in_line = assign_inline_formulas(r, q)
                    inline_q = q[in_line]
                    q = q[~in_line]
                if q.shape[0]:
                    # the text detector also fires on the pieces of a formula: what lies inside a formula box is dropped
                    cx, cy = 0.5 * (r[:, 0] + r[:, 2]), 0.5 * (r[:, 1] + r[:, 3])
                    inside = ((cx[:, None] >= q[None, :, 0]) & (cx[:, None] <= q[None, :, 2]) &
                              (cy[:, None] >= q[None, :, 1]) & (cy[:, None] <= q[None, :, 3])).any(1)
                    rows = np.zeros((q.shape[0], 8), np.float32)
                    rows[:, :5] = q[:, :5]
                    rows[:, 7] = 1.0
                    # on a skewed scan the formula map gives the upright bounding box of a slanted formula: take the
                    # page's skew from its text lines, solve the box for the turned rectangle (as the text boxes are)
                    # and let the crop kernel turn it upright - the formula recogniser is trained on level formulas
                    th = float(np.median(r[:, 6])) if r.shape[0] >= 3 else 0.0
                    if 0.004 <= abs(th) <= 0.35:
                        c, sn = np.cos(th), abs(np.sin(th))
                        c2 = c * c - sn * sn
                        Wb, Hb = rows[:, 2] - rows[:, 0], rows[:, 3] - rows[:, 1]
                        L, T = (Wb * c - Hb * sn) / c2, (Hb * c - Wb * sn) / c2
                        okr = (L >= 0.5 * Wb) & (T >= 0.35 * Hb) & (T >= cfg.formula_min_side_px * 0.5)
                        cxr, cyr = 0.5 * (rows[:, 0] + rows[:, 2]), 0.5 * (rows[:, 1] + rows[:, 3])
                        L, T = np.where(okr, L, Wb), np.where(okr, T, Hb)
                        rows[:, 0], rows[:, 2] = cxr - 0.5 * L, cxr + 0.5 * L
                        rows[:, 1], rows[:, 3] = cyr - 0.5 * T, cyr + 0.5 * T
                        rows[:, 6] = np.where(okr, th, 0.0)
                    r = np.concatenate([r[~inside], rows], 0)
            fg = np.zeros((0, 4), np.float32)
            if figs is not None and figs[pi].shape[0]:
                f = figs[pi]
                big = (np.minimum(f[:, 2] - f[:, 0], f[:, 3] - f[:, 1]) >= cfg.fig_min_side_px)
                f = f[big]
                if f.shape[0] > cfg.max_figures_per_page:
                    area = (f[:, 2] - f[:, 0]) * (f[:, 3] - f[:, 1])
                    f = f[np.argsort(-area)[: cfg.max_figures_per_page]]
                fg = f
            if fg.shape[0]:
                area = (fg[:, 2] - fg[:, 0]) * (fg[:, 3] - fg[:, 1])
                fg = fg[area <= cfg.fig_max_page_share * float(p.hw[0]) * float(p.hw[1])]
            if r.shape[0] and fg.shape[0]:
                # what the detector reads inside a figure (axis labels, legends) is part of the figure, not of the
                # text - unless it is a lot of text, in which case the region is no figure
                cx, cy = 0.5 * (r[:, 0] + r[:, 2]), 0.5 * (r[:, 1] + r[:, 3])
                inside = ((cx[:, None] >= fg[None, :, 0]) & (cx[:, None] <= fg[None, :, 2]) &
                          (cy[:, None] >= fg[None, :, 1]) & (cy[:, None] <= fg[None, :, 3]))
                real = inside.sum(0) < cfg.fig_max_lines
                for k in np.nonzero(real & (inside.sum(0) >= 2))[0]:
                    # a ruled table looks like a framed figure to the block head (often only part of it): when the
                    # text lines on the rows that cross the "figure" form a grid of cells, it is a table - its lines
                    # stay text and the layout analysis makes them a table block
                    rows = (cy >= fg[k, 1]) & (cy <= fg[k, 3])
                    if int(rows.sum()) >= 6 and layout.boxes_form_table(r[rows, :4]):
                        real[k] = False
                fg, inside = fg[real], inside[:, real]
                r = r[~inside.any(1)]
            if r.shape[0] == 0 and fg.shape[0] == 0:
                out.append((np.zeros((0, 8), np.float32), np.zeros((0, 5), np.float32)))
                continue
            # reading order by layout analysis (columns, headings, header / footer, figures as obstacles), in the
            # frame of the upright page when the scan is skewed
            ang = r[:, 6] if r.shape[1] > 6 else np.zeros((r.shape[0],), np.float32)
            b4, _ = deskewed_boxes(r[:, :4], ang, p.hw)
            order, block, kind = layout.order_page_native(b4, float(p.hw[0]), fg if fg.shape[0] else None)
            n = r.shape[0]
            is_line = order < n
            lo = order[is_line]
            score = r[lo, 4:5] if r.shape[1] > 4 else np.ones((lo.size, 1), np.float32)
            # kind: 2 for a cell of a table the layout analysis found (a short cell that the formula channel took for a
            # formula is a cell all the same: it sits in the grid), else the detector's display-formula flag
            knd = np.where(kind[is_line] == 2, 2.0, r[lo, 7]).astype(np.float32)
            lines = np.concatenate([r[lo, :4], score, block[is_line][:, None].astype(np.float32), ang[lo][:, None],
                                    knd[:, None]], 1)
            pos = np.nonzero(~is_line)[0]
            fo = np.concatenate([fg[order[pos] - n], pos[:, None].astype(np.float32)], 1) if pos.size else np.zeros((0, 5), np.float32)
            if inline_q is not None and inline_q.shape[0]:
                # (the pieces of a line become one line again in ``_finish``: figure positions stay as they are)
                spans, _ = assign_inline_formulas(np.where(lines[:, 7:8] == KIND_TEXT, lines, np.nan), inline_q,
                                                  cfg.inline_min_score)
                if spans and self.on_gpu and pi < len(db.srcs):
                    spans = self._refine_spans(db.srcs[pi], lines, spans)
                if spans:
                    lines = np.concatenate([split_line_at_formulas(lines[k], spans[k]) if k in spans else lines[k : k + 1]
                                            for k in range(lines.shape[0])], 0)
            out.append((lines.astype(np.float32), fo.astype(np.float32)))
        return out

    def _refine_spans(self, src: torch.Tensor, lines: np.ndarray, spans: dict) -> dict:
        """The formula map has a quarter of the detector's resolution and its thresholded regions overshoot a formula's
        ends by about half a line height: cutting there would clip the neighbouring words.  Move every end to the
        middle of the widest gap between glyphs near it (the word space), read from the page pixels: the darkest sample
        of each pixel column over the middle of the line, searched around a point ``inline_inset`` line heights in."""
        cfg = self.cfg
        H, W = int(src.shape[0]), int(src.shape[1])
        L = 128
        jobs, profs = [], []
        for k, sp in spans.items():
            cx, cy = 0.5 * (lines[k, 0] + lines[k, 2]), 0.5 * (lines[k, 1] + lines[k, 3])
            w, h = lines[k, 2] - lines[k, 0], lines[k, 3] - lines[k, 1]
            c, sn = float(np.cos(lines[k, 6])), float(np.sin(lines[k, 6]))
            # the thresholded map reaches beyond the formula at either end (measured on the trained detector: 25 - 35
            # page pixels on average, half a line height): start the search ``inline_inset`` line heights further in
            reach = int(min(L // 2 - 1, max(10.0, 0.8 * h)))
            for j, (a0, a1, _) in enumerate(sp):
                ins = min(cfg.inline_inset * h, 0.3 * (a1 - a0))
                for end, a in ((0, a0 + ins), (1, a1 - ins)):
                    if abs(abs(a0 if end == 0 else a1) - 0.5 * w) < 2.0:      # the formula ends where the line ends
                        continue
                    bx, by = int(round(cx + a * c)), int(round(cy + a * sn))
                    x0, x1 = max(0, bx - reach), min(W, bx + reach + 1)
                    y0, y1 = max(0, int(by - 0.28 * h)), min(H, int(by + 0.28 * h) + 1)
                    if x1 - x0 < 8 or y1 - y0 < 2:
                        continue
                    strip = src[y0:y1, x0:x1]
                    if strip.dim() == 3:
                        strip = strip[..., 0]
                    pr = strip.amin(0)
                    profs.append(torch.nn.functional.pad(pr, (0, L - pr.shape[0])))
                    jobs.append((k, j, end, x0 - bx, int(pr.shape[0]), ins))
        if not jobs:
            return spans
        host = torch.stack(profs).cpu().numpy()
        out = {k: [list(t) for t in sp] for k, sp in spans.items()}
        for row, (k, j, end, off, n, ins) in zip(host, jobs):
            g = widest_gap(row[:n], -off)
            if g is not None:                                 # along the (nearly level) line: one column = one pixel
                out[k][j][end] += float(g + off) + (ins if end == 0 else -ins)
        return {k: [tuple(t) for t in sp if t[1] - t[0] >= 4.0] for k, sp in out.items()}

    # ------------------------------------------------------------------ recognition stage
    def _plan_crops(self, db: _DetBatch, det_boxes):
        """Host side of the line crops: choose boxes per page, compute crop geometry, group by width bucket.  ``boxes`` carries
        the line angle in column 4."""
        cfg = self.cfg
        per_page = []
        all_boxes, all_page = [], []
        for pi, (p, (bx, fg)) in enumerate(zip(db.pages, det_boxes)):
            n_det = int(bx.shape[0])
            used_prior = False
            b5 = bx[:, [0, 1, 2, 3, 6]] if n_det else np.zeros((0, 5), np.float32)
            blk = bx[:, 5].astype(np.int32) if n_det else np.zeros((0,), np.int32)
            knd = bx[:, 7].astype(np.int32) if n_det and bx.shape[1] > 7 else np.zeros((b5.shape[0],), np.int32)
            if p.layout_prior is not None and pi not in db.page_errors and not self._detections_usable(bx, p):
                b4 = np.asarray(p.layout_prior, dtype=np.float32).reshape(-1, 4)
                order, blk, lk = layout.order_page_native(b4, float(p.hw[0]))
                b5 = np.concatenate([b4[order], np.zeros((b4.shape[0], 1), np.float32)], 1)
                knd = np.where(lk == 2, 2, 0).astype(np.int32)
                used_prior = True
                fg = fg[:0]
            b5 = b5[: cfg.max_lines_per_page]
            blk = blk[: cfg.max_lines_per_page]
            knd = knd[: cfg.max_lines_per_page]
            per_page.append((n_det, used_prior, b5, blk, fg, knd))
            all_boxes.append(b5)
            all_page.append(np.full((b5.shape[0],), pi, dtype=np.int32))
        boxes = np.concatenate(all_boxes, 0) if all_boxes else np.zeros((0, 5), np.float32)
        page_of = np.concatenate(all_page, 0) if all_page else np.zeros((0,), np.int32)
        kind = np.concatenate([pp[5] for pp in per_page], 0) if per_page else np.zeros((0,), np.int32)
        w = boxes[:, 2] - boxes[:, 0]
        h = boxes[:, 3] - boxes[:, 1]
        ok = (w >= cfg.min_line_px) & (h >= cfg.min_line_px) & ~np.isin(kind, _MATH_KINDS)
        rot = (h > cfg.vertical_ratio * w).astype(np.int32)
        long_side = np.where(rot == 1, h, w)
        short_side = np.maximum(np.where(rot == 1, w, h), 1.0)
        out_w = np.clip(np.rint(long_side * (cfg.rec_height / short_side)), 8, cfg.rec_max_width).astype(np.int32)
        bucket = np.minimum(_round_up_np(out_w, cfg.rec_width_quantum), cfg.rec_max_width)
        return per_page, boxes, page_of, ok, rot, out_w, bucket

    def _detections_usable(self, bx: np.ndarray, p: PageImage) -> bool:
        """A trained detector yields a plausible number of line-shaped boxes; random weights do not."""
        if bx.shape[0] == 0:
            return False
        n_prior = 0 if p.layout_prior is None else len(p.layout_prior)
        w = bx[:, 2] - bx[:, 0]
        h = bx[:, 3] - bx[:, 1]
        line_like = int(((w > 2.0 * h) | (h > 2.0 * w)).sum())
        return line_like >= max(1, n_prior // 2) and bx.shape[0] <= 4 * max(n_prior, 8)

    def _submit_rec(self, db: _DetBatch, det_boxes: List[np.ndarray]) -> _RecBatch:
        cfg = self.cfg
        rb = _RecBatch()
        rb.det = db
        per_page, boxes, page_of, ok, rot, out_w, bucket = self._plan_crops(db, det_boxes)
        rb.results = (per_page, boxes, page_of, ok)
        rb.groups = []
        rb.keep = []
        rb.fig_crops = self._fetch_figures(db, per_page)
        rb.formula_out = self._submit_formulas(db, per_page, boxes, page_of)
        idx_all = np.nonzero(ok)[0]
        if idx_all.size == 0:
            rb.event = None
            if (rb.fig_crops or rb.formula_out is not None) and self.on_gpu:
                rb.event = torch.cuda.Event()
                rb.event.record(db.stream)
            return rb
        # source pointer / dims per page
        ptrs = np.array([s.data_ptr() for s in db.srcs], dtype=np.uint64)
        Hs = np.array([s.shape[0] for s in db.srcs], dtype=np.int32)
        Ws = np.array([s.shape[1] for s in db.srcs], dtype=np.int32)
        Cs = np.array([1 if s.dim() == 2 else s.shape[2] for s in db.srcs], dtype=np.int32)
        Ps = np.array([s.stride(0) for s in db.srcs], dtype=np.int32)
        ctx = torch.cuda.stream(db.stream) if self.on_gpu else _null_ctx()
        used_graphs: set = set()
        lens_all = np.maximum((out_w + 3) // 4, 1).astype(np.int32)
        if self.rec_kind == "trocr":
            with ctx:
                self._submit_rec_trocr(rb, db, idx_all, boxes, page_of, rot, (ptrs, Hs, Ws, Cs, Ps))
            return rb
        with ctx:
            buckets = [(int(wb), idx_all[bucket[idx_all] == wb]) for wb in np.unique(bucket[idx_all])]
            # The recurrence costs T sequential steps per launch whatever the batch, so width buckets must not
            # run one after another: a single page spread over 8 buckets spent 49 ms in 16 BiLSTM launches
            # (profiles/09) against 2 launches in the ragged pass.  Replayed per-bucket graphs are kept for
            # the one case they win: every crop in a single bucket.
            small = cfg.use_hip_graphs and len(buckets) == 1 and idx_all.size <= cfg.graph_max_crops
            if self.on_gpu and not small:
                # throughput regime: every bucket of the batch in ONE ragged recognition pass
                rb.groups.append(self._rec_gpu_ragged(buckets, boxes, page_of, rot, out_w, lens_all,
                                                      (ptrs, Hs, Ws, Cs, Ps)))
            else:
                for wb, sel in buckets:
                    for c0 in range(0, sel.size, cfg.rec_max_batch):
                        ids = sel[c0 : c0 + cfg.rec_max_batch]
                        if self.on_gpu:
                            items = self._crop_items(ids, boxes, page_of, rot, out_w, (ptrs, Hs, Ws, Cs, Ps))
                            out = self._rec_gpu(items, lens_all[ids], wb, db.slot, used_graphs)
                        else:
                            regions = [
                                rs.Region(db.srcs[page_of[i]], tuple(float(v) for v in boxes[i, :4]), cfg.rec_height,
                                          int(out_w[i]), int(rot[i]), float(boxes[i, 4])) for i in ids
                            ]
                            x = rs.resample_regions(regions, cfg.rec_height, wb, rs.HALF_MEAN, rs.HALF_STD)
                            out = self.rec.infer(x, torch.from_numpy(lens_all[ids]))
                        rb.groups.append((ids, out))
            if self.on_gpu:
                rb.event = torch.cuda.Event()
                rb.event.record(db.stream)
            else:
                rb.event = None
        return rb

    def _submit_rec_trocr(self, rb: _RecBatch, db: _DetBatch, idx_all, boxes, page_of, rot, src) -> None:
        """Transformer recognizer: every line is resized to the encoder's S x S input (as TrOCR does), encoded in
        batches of ``trocr_batch`` and decoded greedily for ``trocr_max_tokens`` hipGraph-replayed steps."""
        cfg = self.cfg
        S = self.rec.cfg.image_size
        if self.on_gpu and self._trocr_event is not None:
            db.stream.wait_event(self._trocr_event)          # one decode state: recognition passes run in order
        for c0 in range(0, idx_all.size, cfg.trocr_batch):
            ids = idx_all[c0 : c0 + cfg.trocr_batch]
            if self.on_gpu:
                items = self._crop_items(ids, boxes, page_of, rot, np.full(boxes.shape[0], S, np.int32), src, out_h=S)
                host = torch.from_numpy(items.view(np.uint8).reshape(-1)).pin_memory()
                dev = host.to(self.device, non_blocking=True)
                x = rs.resample_items_gpu(dev, ids.size, S, S, rs.HALF_MEAN, rs.HALF_STD)
                rb.keep.append((host, dev))
            else:
                regions = [rs.Region(db.srcs[page_of[i]], tuple(float(v) for v in boxes[i, :4]), S, S, int(rot[i]),
                                     float(boxes[i, 4])) for i in ids]
                x = rs.resample_regions(regions, S, S, rs.HALF_MEAN, rs.HALF_STD)
            toks, conf = self.rec.generate(x, max_len=cfg.trocr_max_tokens, use_graph=cfg.use_hip_graphs)
            if self.on_gpu:
                h_tok = torch.empty(toks.shape, dtype=torch.int32, pin_memory=True)
                h_cf = torch.empty(conf.shape, dtype=torch.float32, pin_memory=True)
                h_tok.copy_(toks, non_blocking=True)
                h_cf.copy_(conf, non_blocking=True)
                rb.groups.append((ids, ("tokens", h_tok, h_cf)))
            else:
                rb.groups.append((ids, ("tokens", toks.to(torch.int32), conf.float())))
        if self.on_gpu:
            rb.event = torch.cuda.Event()
            rb.event.record(db.stream)
            self._trocr_event = rb.event
        else:
            rb.event = None

    def _submit_formulas(self, db: _DetBatch, per_page, boxes: np.ndarray, page_of: np.ndarray):
        """Formula boxes of the batch -> letter-boxed 96 x 384 crops -> encoder-decoder -> LaTeX token ids (started
        here, read in ``_finish``).  Returns (global line indices, tokens, confidences) or None."""
        kind = np.concatenate([pp[5] for pp in per_page], 0) if per_page else np.zeros((0,), np.int32)
        idx = np.nonzero(np.isin(kind, _MATH_KINDS))[0]
        if idx.size == 0 or self.formula_rec is None:
            return None
        from ..models.trocr import FORMULA_HW

        cfg = self.cfg
        H, W = FORMULA_HW
        w = np.maximum(boxes[idx, 2] - boxes[idx, 0], 1.0)
        h = np.maximum(boxes[idx, 3] - boxes[idx, 1], 1.0)
        sc = np.minimum(W / w, H / h)
        ow = np.clip(np.rint(w * sc), 1, W).astype(np.int32)
        oh = np.clip(np.rint(h * sc), 1, H).astype(np.int32)
        ctx = torch.cuda.stream(db.stream) if self.on_gpu else _null_ctx()
        with ctx:
            if self.on_gpu:
                if self._formula_event is not None:
                    db.stream.wait_event(self._formula_event)     # one decode state: formula passes run in order
                ptrs = np.array([s_.data_ptr() for s_ in db.srcs], dtype=np.uint64)
                Hs = np.array([s_.shape[0] for s_ in db.srcs], dtype=np.int32)
                Ws = np.array([s_.shape[1] for s_ in db.srcs], dtype=np.int32)
                Cs = np.array([1 if s_.dim() == 2 else s_.shape[2] for s_ in db.srcs], dtype=np.int32)
                Ps = np.array([s_.stride(0) for s_ in db.srcs], dtype=np.int32)
                out_w_all = np.zeros((boxes.shape[0],), np.int32)
                out_w_all[idx] = ow
                items = self._crop_items(idx, boxes, page_of, np.zeros((boxes.shape[0],), np.int32), out_w_all,
                                         (ptrs, Hs, Ws, Cs, Ps))
                items["out_h"] = oh
                host = torch.from_numpy(items.view(np.uint8).reshape(-1)).pin_memory()
                dev = host.to(self.device, non_blocking=True)
                x = rs.resample_items_gpu(dev, idx.size, H, W, rs.HALF_MEAN, rs.HALF_STD)
                toks, conf = self.formula_rec.generate(x, max_len=cfg.formula_max_tokens, use_graph=cfg.use_hip_graphs,
                                                       eos_check_every=0)
                h_tok = torch.empty(toks.shape, dtype=torch.int32, pin_memory=True)
                h_cf = torch.empty(conf.shape, dtype=torch.float32, pin_memory=True)
                h_tok.copy_(toks, non_blocking=True)
                h_cf.copy_(conf, non_blocking=True)
                self._formula_event = torch.cuda.Event()
                self._formula_event.record(db.stream)
                return idx, h_tok, h_cf, (host, dev, x)
            regions = [rs.Region(db.srcs[page_of[i]], tuple(float(v) for v in boxes[i, :4]), int(oh[k]), int(ow[k]), 0,
                                 float(boxes[i, 4]) if boxes.shape[1] > 4 else 0.0) for k, i in enumerate(idx)]
            x = rs.resample_regions(regions, H, W, rs.HALF_MEAN, rs.HALF_STD)
            toks, conf = self.formula_rec.generate(x, max_len=cfg.formula_max_tokens, use_graph=False, eos_check_every=8)
            return idx, toks.to(torch.int32), conf.float(), None

    def _fetch_figures(self, db: _DetBatch, per_page) -> Dict[Tuple[int, int], torch.Tensor]:
        """Start the device-to-host copies of the figure crops of this batch (page raster pixels, full resolution)."""
        crops: Dict[Tuple[int, int], torch.Tensor] = {}
        ctx = torch.cuda.stream(db.stream) if self.on_gpu else _null_ctx()
        with ctx:
            for pi, (_, _, _, _, fg, _) in enumerate(per_page):
                src = db.srcs[pi]
                Hs, Ws = int(src.shape[0]), int(src.shape[1])
                for k in range(fg.shape[0]):
                    x0, y0 = max(0, int(np.floor(fg[k, 0]))), max(0, int(np.floor(fg[k, 1])))
                    x1, y1 = min(Ws, int(np.ceil(fg[k, 2]))), min(Hs, int(np.ceil(fg[k, 3])))
                    if x1 - x0 < 2 or y1 - y0 < 2:
                        continue
                    view = src[y0:y1, x0:x1]
                    if self.on_gpu:
                        host = torch.empty(view.shape, dtype=torch.uint8, pin_memory=True)
                        host.copy_(view, non_blocking=True)
                    else:
                        host = view.clone()
                    crops[(pi, k)] = host
        return crops

    def _crop_items(self, ids, boxes, page_of, rot, out_w, src, out_h: Optional[int] = None) -> np.ndarray:
        ptrs, Hs, Ws, Cs, Ps = src
        items = np.zeros(ids.size, dtype=rs.ITEM_DTYPE)
        pg = page_of[ids]
        items["src"], items["Hs"], items["Ws"], items["C"] = ptrs[pg], Hs[pg], Ws[pg], Cs[pg]
        items["rot"] = rot[ids]
        items["pitch"] = Ps[pg]
        items["x0"], items["y0"] = boxes[ids, 0], boxes[ids, 1]
        items["x1"], items["y1"] = boxes[ids, 2], boxes[ids, 3]
        items["angle"] = boxes[ids, 4]
        items["out_h"], items["out_w"] = (out_h or self.cfg.rec_height), out_w[ids]
        return items

    def _rec_gpu_ragged(self, buckets, boxes, page_of, rot, out_w, lens_all, src):
        """One upload (crop table + ragged layout), per-bucket crop kernels, one ragged CRNN pass, one D2H."""
        cfg = self.cfg
        order = np.concatenate([sel for _, sel in buckets])
        n = order.size
        items = self._crop_items(order, boxes, page_of, rot, out_w, src)
        tcap = np.concatenate([np.full(sel.size, wb // 4, dtype=np.int32) for wb, sel in buckets])
        lens = np.minimum(lens_all[order], tcap).astype(np.int32)
        row_off = (np.cumsum(tcap, dtype=np.int64) - tcap).astype(np.int64)
        perm = np.argsort(-lens, kind="stable").astype(np.int32)
        rows, tmax = int(tcap.sum()), int(tcap.max())
        blob = np.concatenate([row_off.view(np.uint8), items.view(np.uint8).reshape(-1), lens.view(np.uint8),
                               tcap.view(np.uint8), perm.view(np.uint8)])
        host = torch.from_numpy(blob).pin_memory()
        dev = host.to(self.device, non_blocking=True)
        o = 0
        row_off_d = dev[o : o + 8 * n].view(torch.int64); o += 8 * n
        items_d = dev[o : o + 56 * n]; o += 56 * n
        lens_d = dev[o : o + 4 * n].view(torch.int32); o += 4 * n
        tcap_d = dev[o : o + 4 * n].view(torch.int32); o += 4 * n
        perm_d = dev[o : o + 4 * n].view(torch.int32)
        xs, c0 = [], 0
        # crop resample fused with the recogniser's first convolution where the model offers it
        first = self.rec.first_conv_packed() if hasattr(self.rec, "first_conv_packed") else None
        fused = first is not None and rs.stem_serves(first[0], first[3], first[4])
        for wb, sel in buckets:
            it = items_d[56 * c0 : 56 * (c0 + sel.size)]
            if fused:
                xs.append(rs.resample_stem_items_gpu(it, sel.size, cfg.rec_height, wb, rs.HALF_MEAN, rs.HALF_STD,
                                                     first[0], first[1], first[2]))
            else:
                xs.append(rs.resample_items_gpu(it, sel.size, cfg.rec_height, wb, rs.HALF_MEAN, rs.HALF_STD))
            c0 += sel.size
        lay = lstm_ops.RaggedLayout(row_off_d, tcap_d, lens_d, perm_d, rows, tmax)
        ids, ln, conf = self.rec.infer_ragged(xs, lay, first_conv_done=fused)
        h_ids = torch.empty((n, tmax), dtype=torch.int32, pin_memory=True)
        h_ln = torch.empty((n,), dtype=torch.int32, pin_memory=True)
        h_cf = torch.empty((n,), dtype=torch.float32, pin_memory=True)
        h_ids.copy_(ids, non_blocking=True)
        h_ln.copy_(ln, non_blocking=True)
        h_cf.copy_(conf, non_blocking=True)
        return order, (h_ids, h_ln, h_cf)

    def _rec_gpu(self, items: np.ndarray, lens_np: np.ndarray, wb: int, slot: int, used_graphs: set):
        """Upload one crop table, run crop+CRNN+decode, start the D2H of the result.  Returns pinned host tensors."""
        cfg = self.cfg
        n = items.shape[0]
        T = wb // 4
        use_graph = cfg.use_hip_graphs and n <= cfg.graph_max_crops
        nb = max(16, 1 << (n - 1).bit_length())
        if use_graph and (slot, nb, wb) not in used_graphs:
            g = self._graphs.get((slot, nb, wb))
            if g is None:
                g = _RecGraph(self, nb, wb)
                self._graphs[(slot, nb, wb)] = g
            used_graphs.add((slot, nb, wb))
            return g.run(items, lens_np)
        blob = np.concatenate([items.view(np.uint8).reshape(-1), lens_np.view(np.uint8).reshape(-1)])
        host = torch.from_numpy(blob).pin_memory()
        dev = host.to(self.device, non_blocking=True)
        items_dev = dev[: n * 56]
        lens_dev = dev[n * 56 :].view(torch.int32)
        x = rs.resample_items_gpu(items_dev, n, cfg.rec_height, wb, rs.HALF_MEAN, rs.HALF_STD)
        ids, ln, conf = self.rec.infer(x, lens_dev)
        h_ids = torch.empty((n, T), dtype=torch.int32, pin_memory=True)
        h_ln = torch.empty((n,), dtype=torch.int32, pin_memory=True)
        h_cf = torch.empty((n,), dtype=torch.float32, pin_memory=True)
        h_ids.copy_(ids, non_blocking=True)
        h_ln.copy_(ln, non_blocking=True)
        h_cf.copy_(conf, non_blocking=True)
        return h_ids, h_ln, h_cf

    def _finish(self, rb: _RecBatch) -> List[PageResult]:
        if rb.event is not None:
            rb.event.synchronize()
        per_page, boxes, page_of, ok = rb.results
        n_total = boxes.shape[0]
        texts: List[str] = [""] * n_total
        confs = np.zeros((n_total,), np.float32)
        for ids, out in rb.groups:
            if isinstance(out[0], str):            # transformer recognizer: token ids -> native BPE decode
                _, h_tok, h_cf = out
                strs = self.rec.tokens_to_text(h_tok[: len(ids)])
                for j, gi in enumerate(ids):
                    texts[gi] = strs[j]
                confs[ids] = h_cf[: len(ids)].numpy()
                continue
            h_ids, h_ln, h_cf = out
            strs = self._strings(h_ids[: len(ids)], h_ln[: len(ids)])
            cf = h_cf[: len(ids)].numpy() if isinstance(h_cf, torch.Tensor) else h_cf
            for j, gi in enumerate(ids):
                texts[gi] = strs[j]
            confs[ids] = cf
        if rb.formula_out is not None:
            fidx, h_tok, h_cf = rb.formula_out[:3]
            strs = self.formula_rec.tokens_to_text(h_tok[: len(fidx)])
            for j, gi in enumerate(fidx):
                texts[gi] = strs[j].strip()
            confs[fidx] = h_cf[: len(fidx)].numpy()
        results = []
        pos = 0
        for pi, (p, (n_det, used_prior, b4, blk, fg, knd)) in enumerate(zip(rb.det.pages, per_page)):
            r = PageResult(p.doc_id, p.page_index, p.hw[1], p.hw[0],
                           n_detected=n_det, used_layout_prior=used_prior, error=rb.det.page_errors.get(pi))
            # rows of one line follow each other: its first piece, then the other pieces of a line cut at formulas
            # (text, formula + the same crop read as text, ...) and the whole line once more
            n_rows, k = b4.shape[0], 0
            follow = (KIND_TEXT_CONT, KIND_MATH_CONT, KIND_MATH_TEXT, KIND_LINE_ALT)
            while k < n_rows:
                e = k + 1
                while e < n_rows and knd[e] in follow:
                    e += 1
                if e == k + 1 and knd[k] in (KIND_TEXT, KIND_CELL, KIND_FORMULA):          # the ordinary case
                    gi = pos + k
                    if knd[k] == KIND_FORMULA:
                        r.lines.append(TextLine(tuple(float(v) for v in b4[k, :4]), texts[gi], float(confs[gi]), int(blk[k]),
                                                0.0, "equation"))
                    elif ok[gi]:
                        r.lines.append(TextLine(tuple(float(v) for v in b4[k, :4]), texts[gi], float(confs[gi]), int(blk[k]),
                                                float(b4[k, 4]), "cell" if knd[k] == KIND_CELL else "text"))
                else:
                    line = self._join_pieces(range(k, e), pos, b4, blk, knd, ok, texts, confs)
                    if line is not None:
                        r.lines.append(line)
                k = e
            for k in range(fg.shape[0]):
                crop = rb.fig_crops.get((pi, k))
                if crop is not None:
                    r.figures.append(Figure(tuple(float(v) for v in fg[k, :4]), int(fg[k, 4]), crop.numpy()))
            pos += b4.shape[0]
            if getattr(self.rec, "multi_script", False):
                _apply_page_script(r.lines)
            self.lines_done += len(r.lines)
            results.append(r)
        self.pages_done += len(results)
        return results

    def _join_pieces(self, rows, pos, b4, blk, knd, ok, texts, confs) -> Optional[TextLine]:
        """One text line from the pieces it was cut into at the formulas inside it.  A formula counts when it was read
        with a mean token probability of at least ``inline_min_conf``, has a command, subscript or superscript
        (``inline_needs_markup``: "x = 3", "(a)", "12" are read as well by the text recogniser, and such tokens of
        ordinary prose are what the formula map fires on by mistake) and, in a line of Greek prose, more than Greek
        letters and operators; one that does not is replaced by the same crop read as text.  When no formula of the
        line counts, the line is taken as the text recogniser read it whole (no cuts, no seams)."""
        cfg = self.cfg
        alt = None
        for k in rows:
            if knd[k] == KIND_LINE_ALT and ok[pos + k]:
                alt = TextLine(tuple(float(v) for v in b4[k, :4]), texts[pos + k], float(confs[pos + k]), int(blk[k]),
                               float(b4[k, 4]), "text")
        greek = alt is not None and line_script(alt.text) == "greek"
        pieces, box, conf, fconf, n_math = [], None, 1.0, 1.0, 0
        rows = list(rows)
        for n, k in enumerate(rows):
            gi = pos + k
            if knd[k] in (KIND_LINE_ALT, KIND_MATH_TEXT):
                continue
            if knd[k] in (KIND_MATH_HEAD, KIND_MATH_CONT):
                latex = texts[gi].strip()
                counts = (bool(latex) and float(confs[gi]) >= cfg.inline_min_conf and
                          (not cfg.inline_needs_markup or any(c in latex for c in "\\_^")) and
                          not (greek and not any(c in latex for c in "_^(")))
                if counts:
                    piece, used = "$" + latex + "$", float(confs[gi])
                    n_math += 1
                    fconf = min(fconf, used)
                else:
                    nxt = rows[n + 1] if n + 1 < len(rows) else -1
                    if nxt < 0 or knd[nxt] != KIND_MATH_TEXT or not ok[pos + nxt]:
                        continue
                    piece, used = texts[pos + nxt], float(confs[pos + nxt])
            elif ok[gi]:
                piece, used = texts[gi], float(confs[gi])
            else:
                continue
            if piece:
                pieces.append(piece)
            conf = min(conf, used)
            bb = b4[k, :4]
            box = tuple(float(v) for v in bb) if box is None else (min(box[0], float(bb[0])), min(box[1], float(bb[1])),
                                                                  max(box[2], float(bb[2])), max(box[3], float(bb[3])))
        if (n_math == 0 and alt is not None) or box is None:
            return alt
        k0 = rows[0]
        line = TextLine(box, " ".join(pieces), conf, int(blk[k0]), float(b4[k0, 4]), "text")
        line.formula_confidence = fconf
        return line

    def _strings(self, ids: torch.Tensor, lens: torch.Tensor) -> List[str]:
        if self._lut is None:
            return self.rec.decode_strings(ids, lens)
        a = ids.numpy()
        L = lens.numpy()
        ch = self._lut[np.clip(a, 0, len(self._lut) - 1)]
        out = [ch[i, : L[i]].tobytes().decode("utf-32-le") for i in range(a.shape[0])]
        if getattr(self.rec, "multi_script", False):
            # shape classes -> code points (utils/scripts.py); ASCII lines are returned as they are
            out = [t if t.isascii() else resolve_scripts(t) for t in out]
        return out

    # ------------------------------------------------------------------ public API
    def process_batches(self, batches: Iterable[Sequence[PageImage]]) -> Iterator[List[PageResult]]:
        """Software-pipelined execution over an iterable of page batches (results keep batch order)."""
        det_q: List[_DetBatch] = []
        rec_q: List[_RecBatch] = []
        slot = 0
        for batch in batches:
            if len(batch) == 0:
                continue
            t0 = time.perf_counter()
            det_q.append(self._submit_det(batch, slot))
            slot = (slot + 1) % len(self._streams)
            self.timer.add("submit_det", time.perf_counter() - t0)
            if len(det_q) > 1:
                rec_q.append(self._advance(det_q.pop(0)))
            if len(rec_q) > 1:
                yield self._timed_finish(rec_q.pop(0))
        while det_q:
            rec_q.append(self._advance(det_q.pop(0)))
        while rec_q:
            yield self._timed_finish(rec_q.pop(0))

    STREAM_END = object()

    def run_stream(self, next_batch) -> Iterator[List[PageResult]]:
        """Continuous form of ``process_batches`` for a service loop.

        ``next_batch(block)`` returns a list of pages, ``None`` when nothing is ready (only when ``block`` is False)
        or ``OcrEngine.STREAM_END``.  While batches are in flight the engine never blocks on the source: with no new
        batch it advances the oldest stage instead (wait for the detector, submit recognition, finish), so results of
        a lone small document are not held back waiting for a successor, and a successor that arrives meanwhile is
        picked up after one stage."""
        det_q: List[_DetBatch] = []
        rec_q: List[_RecBatch] = []
        slot = 0
        while True:
            batch = next_batch(not det_q and not rec_q)
            if batch is OcrEngine.STREAM_END:
                break
            if batch:
                t0 = time.perf_counter()
                det_q.append(self._submit_det(batch, slot))
                slot = (slot + 1) % len(self._streams)
                self.timer.add("submit_det", time.perf_counter() - t0)
                if len(det_q) > 1:
                    rec_q.append(self._advance(det_q.pop(0)))
                if len(rec_q) > 1:
                    yield self._timed_finish(rec_q.pop(0))
            elif det_q:
                rec_q.append(self._advance(det_q.pop(0)))
                if len(rec_q) > 1:
                    yield self._timed_finish(rec_q.pop(0))
            elif rec_q:
                yield self._timed_finish(rec_q.pop(0))
        while det_q:
            rec_q.append(self._advance(det_q.pop(0)))
        while rec_q:
            yield self._timed_finish(rec_q.pop(0))

    def _advance(self, db: _DetBatch) -> _RecBatch:
        t0 = time.perf_counter()
        boxes = self._det_boxes(db)
        t1 = time.perf_counter()
        rb = self._submit_rec(db, boxes)
        self.timer.add("wait_det", t1 - t0)
        self.timer.add("submit_rec", time.perf_counter() - t1)
        return rb

    def _timed_finish(self, rb: _RecBatch) -> List[PageResult]:
        t0 = time.perf_counter()
        r = self._finish(rb)
        self.timer.add("finish", time.perf_counter() - t0)
        return r

    def process_pages(self, pages: Sequence[PageImage]) -> List[PageResult]:
        out: List[PageResult] = []
        for res in self.process_batches(pages[a:b] for a, b in balanced_chunks(len(pages), self.cfg.max_batch_pages)):
            out.extend(res)
        return out

    def warmup(self, page_hw: Tuple[int, int] = (3508, 2480), batch: Optional[int] = None, lines: int = 40) -> None:
        """Run one synthetic batch so that workspaces, kernels and graphs exist before serving."""
        H, W = page_hw
        B = batch or self.cfg.max_batch_pages
        img = torch.full((H, W), 240, dtype=torch.uint8)
        ys = np.linspace(0.1 * H, 0.9 * H, lines).astype(np.float32)
        prior = np.stack([np.full_like(ys, 0.1 * W), ys, np.full_like(ys, 0.9 * W), ys + 0.014 * H], 1)
        pages = [PageImage(img, "warmup", i, prior) for i in range(B)]
        self.process_pages(pages)
        self.process_pages(pages)


class _RecGraph:
    """hipGraph-captured recognition step for one (crop-count bucket, width bucket).

    Static inputs: the crop table and valid lengths (one device blob, refreshed with a
    single H2D copy per replay).  Static outputs: ids / lengths / confidences."""

    def __init__(self, eng: OcrEngine, nb: int, wb: int):
        self.eng, self.nb, self.wb = eng, nb, wb
        cfg = eng.cfg
        dev = eng.device
        self.blob_host = torch.zeros((nb * 56 + nb * 4,), dtype=torch.uint8).pin_memory()
        self.blob_dev = torch.zeros_like(self.blob_host, device=dev)
        self.items_dev = self.blob_dev[: nb * 56]
        self.lens_dev = self.blob_dev[nb * 56 :].view(torch.int32)
        self.lens_dev.fill_(1)
        T = wb // 4
        self.h_ids = torch.empty((nb, T), dtype=torch.int32, pin_memory=True)
        self.h_ln = torch.empty((nb,), dtype=torch.int32, pin_memory=True)
        self.h_cf = torch.empty((nb,), dtype=torch.float32, pin_memory=True)

        def step():
            x = rs.resample_items_gpu(self.items_dev, nb, cfg.rec_height, wb, rs.HALF_MEAN, rs.HALF_STD)
            return eng.rec.infer(x, self.lens_dev)

        # warm up on a side stream (allocations, lazily-set kernel attributes), then capture
        cur = torch.cuda.current_stream(dev)
        s = torch.cuda.Stream(dev)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            step()
        cur.wait_stream(s)
        torch.cuda.current_stream(dev).synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=s):
            self.ids, self.ln, self.conf = step()
        self.replays = 0

    def run(self, items: np.ndarray, lens_np: np.ndarray):
        n = items.shape[0]
        nb = self.nb
        hb = self.blob_host.numpy()
        hb[: n * 56] = items.view(np.uint8).reshape(-1)
        hb[n * 56 : nb * 56] = 0                      # null src -> zero crops for the padded slots
        lv = hb[nb * 56 :].view(np.int32)
        lv[:n] = lens_np
        lv[n:] = 1
        self.blob_dev.copy_(self.blob_host, non_blocking=True)
        self.graph.replay()
        self.h_ids.copy_(self.ids, non_blocking=True)
        self.h_ln.copy_(self.ln, non_blocking=True)
        self.h_cf.copy_(self.conf, non_blocking=True)
        self.replays += 1
        # One graph exists per pipeline slot, and a slot's previous batch is always finished before
        # its next one is submitted, so handing out views of the pinned outputs is race-free.
        return self.h_ids[:n], self.h_ln[:n], self.h_cf[:n]


class _null_ctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _to_gray(a: np.ndarray) -> np.ndarray:
    if a.ndim == 2:
        return np.ascontiguousarray(a)
    return np.ascontiguousarray((a.astype(np.float32) @ np.array([0.299, 0.587, 0.114], np.float32)).astype(np.uint8))


def _round_up_np(v: np.ndarray, m: int) -> np.ndarray:
    return (v + m - 1) // m * m
