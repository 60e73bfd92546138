"""Page layout analysis: reading order and text blocks from line boxes.

The reference gets reading order from a learned model (``hantian/layoutreader``,
services/ocr_service/download_models_hf.py:51) that magic-pdf runs inside ``pipe_ocr_mode`` / ``pipe_txt_mode``
(ocr_service.py:215-220), with a recursive XY-cut as its own fallback.  No checkpoint is reachable offline, so this
module is the XY-cut: recursive whitespace cuts over the boxes of a page, the same code for detector boxes of a scan
and for the text-line boxes the native PDF reader reports for a born-digital page.  Parity with the learned model
on real documents is unpinned; the behaviours covered by tests are multi-column pages, full-width headings above
columns, page headers / footers, figures inside a column, label / value rows (which must NOT be cut), and tables:
grids of aligned cells (alone on a page, inside a column, under a paragraph without a block gap, with one over-long
cell, with cells that wrap onto further lines) read row by row as one block.

    order, block_of = reading_order(boxes)          # boxes [n, 4] x0, y0, x1, y1, y down

``order`` lists box indices in reading order; ``block_of[i]`` is the leaf region (text block) of box i - lines of
different blocks never join one paragraph.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def _gaps(lo: np.ndarray, hi: np.ndarray) -> List[Tuple[float, float, float]]:
    """Whitespace intervals in the union of [lo, hi] intervals: [(gap width, gap start, gap end)]."""
    o = np.argsort(lo, kind="stable")
    out = []
    reach = hi[o[0]]
    for k in o[1:]:
        if lo[k] > reach:
            out.append((float(lo[k] - reach), float(reach), float(lo[k])))
        reach = max(reach, hi[k])
    return out


def _split_at(idx: np.ndarray, lo: np.ndarray, cuts: Sequence[float]) -> List[np.ndarray]:
    """Partition ``idx`` by cut coordinates (each box belongs to the segment its low edge falls in)."""
    cuts = sorted(cuts)
    seg = np.searchsorted(np.asarray(cuts), lo[idx] + 1e-6)
    return [idx[seg == s] for s in range(len(cuts) + 1) if np.any(seg == s)]


def _rows_then_x(idx: np.ndarray, b: np.ndarray, h_med: float) -> List[int]:
    """Leaf order: top to bottom; boxes whose vertical centres are within half a line height form one row, read
    left to right."""
    if idx.size <= 1:
        return [int(i) for i in idx]
    cy = 0.5 * (b[idx, 1] + b[idx, 3])
    o = idx[np.argsort(cy, kind="stable")]
    cy = np.sort(cy, kind="stable")
    out: List[int] = []
    start = 0
    for i in range(1, o.size + 1):
        if i == o.size or cy[i] - cy[start] > 0.5 * h_med:
            row = o[start:i]
            out.extend(int(k) for k in row[np.argsort(b[row, 0], kind="stable")])
            start = i
    return out


class _Cutter:
    def __init__(self, boxes: np.ndarray, col_gap: float, block_gap: float, min_col_lines: int, text_aspect: float):
        self.b = boxes
        h = boxes[:, 3] - boxes[:, 1]
        self.h_med = float(np.median(h)) if h.size else 1.0
        self.tx = col_gap * self.h_med
        self.ty = block_gap * self.h_med
        self.min_col_lines = min_col_lines
        self.text_aspect = text_aspect
        self.order: List[int] = []
        self.block_of = np.zeros(boxes.shape[0], dtype=np.int32)
        self.table_block: List[bool] = []       # per leaf: its boxes are the cells of a table, row by row
        self.n_blocks = 0

    def leaf(self, idx: np.ndarray, table: bool = False) -> None:
        self.table_block.append(table)
        for i in _rows_then_x(idx, self.b, self.h_med):
            self.order.append(i)
            self.block_of[i] = self.n_blocks
        self.n_blocks += 1

    def rows_of(self, idx: np.ndarray) -> List[np.ndarray]:
        """Rows of a region: boxes whose vertical centres lie within half a line height of the row's first box, each
        row left to right (the order ``leaf`` emits)."""
        b = self.b
        cy = 0.5 * (b[idx, 1] + b[idx, 3])
        o = idx[np.argsort(cy, kind="stable")]
        cy = np.sort(cy, kind="stable")
        rows, start = [], 0
        for i in range(1, o.size + 1):
            if i == o.size or cy[i] - cy[start] > 0.5 * self.h_med:
                row = o[start:i]
                rows.append(row[np.argsort(b[row, 0], kind="stable")])
                start = i
        return rows

    def table_cuts(self, idx: np.ndarray, tol: int) -> List[float]:
        """Column boundaries of a would-be table: the middles of the x-intervals, at least 0.6 line heights wide, that
        at most ``tol`` cells reach into (one over-long cell that runs up to its neighbour must not glue two columns
        of an eight-row table together), with cells on both sides."""
        b = self.b
        x0, x1 = b[idx, 0], b[idx, 2]
        edges = np.unique(np.concatenate([x0, x1]))
        cuts: List[float] = []
        start = None                     # left end of the current run of sparsely covered intervals
        for a, e in zip(edges[:-1], edges[1:]):
            cover = int(np.count_nonzero((x0 < e) & (x1 > a)))
            if cover <= tol:
                if start is None:
                    start = float(a)
                end = float(e)
            else:
                if start is not None and end - start >= 0.6 * self.h_med and np.any(x1 <= start + 1e-9):
                    cuts.append(0.5 * (start + end))
                start = None
        return cuts

    def is_table(self, idx: np.ndarray, min_rows: int) -> bool:
        """A region is a table when its boxes form a grid: at least ``min_rows`` rows of two or more cells each, the
        cells of all rows falling into common columns (the whitespace between the columns runs through every row - from
        six rows on, through all but one - at least 0.6 line heights wide), no row with two cells in one column, every column filled in at least half of the
        rows, and at most one column of long text (two columns of running text whose lines happen to share baselines
        are not a table).  Cells may wrap: a row of fewer boxes at line spacing under a table row (``continues_row``)
        belongs to that row."""
        b = self.b
        if idx.size < 2 * min_rows:
            return False
        if np.any(b[idx, 3] - b[idx, 1] > 3.0 * self.h_med):      # a figure or a rotated line: no cell
            return False
        rows = self.rows_of(idx)
        n_full = sum(1 for r in rows if r.size >= 2)
        if n_full < min_rows or rows[0].size < 2:
            return False
        cuts = self.table_cuts(idx, 1 if n_full >= 6 else 0)
        if not cuts:
            return False
        C = len(cuts) + 1
        filled = [0] * C
        aspect: List[List[float]] = [[] for _ in range(C)]
        ca = np.asarray(cuts)
        n_rows = 0
        base_cols: set = set()
        base_size = 0
        # the white between the cells of two complete rows that follow each other, column by column: the lines of one
        # cell are closer than that (in two columns of short lines that merely share some baselines they are not)
        most = max(r.size for r in rows)
        cols_of = [np.searchsorted(ca, b[r, 0] + 1e-6) for r in rows]
        gaps = []
        for k in range(1, len(rows)):
            if rows[k].size == most and rows[k - 1].size == most and np.array_equal(cols_of[k], cols_of[k - 1]):
                gaps.extend((b[rows[k], 1] - b[rows[k - 1], 3]).tolist())
        ref_gap = float(np.median(gaps)) if gaps else None
        bottom = [0.0] * C                     # lower edge of the last box seen in each column
        for k, r in enumerate(rows):
            col = cols_of[k]
            if np.unique(col).size != col.size:
                return False
            if k > 0 and self.continues_row(r, col, bottom, base_cols, base_size, ca, ref_gap):
                # the second (third ...) line of cells that wrap: part of the row above, no row of its own
                for i, c in zip(r, col):
                    aspect[c].append(float((b[i, 2] - b[i, 0]) / max(b[i, 3] - b[i, 1], 1.0)))
                    bottom[c] = float(b[i, 3])
                continue
            if r.size < 2:
                return False
            n_rows += 1
            base_cols, base_size = set(int(c) for c in col), int(r.size)
            for i, c in zip(r, col):
                filled[c] += 1
                aspect[c].append(float((b[i, 2] - b[i, 0]) / max(b[i, 3] - b[i, 1], 1.0)))
                bottom[c] = float(b[i, 3])
        if n_rows < min_rows or any(2 * f < n_rows for f in filled):
            return False
        return sum(1 for a in aspect if float(np.median(a)) >= 10.0) <= 1

    def continues_row(self, r: np.ndarray, col: np.ndarray, bottom: List[float], base_cols: set, base_size: int,
                      cuts: np.ndarray, ref_gap: Optional[float]) -> bool:
        """Is row ``r`` the next line of cells of the table row above it that wrap?  It has fewer boxes than that row
        has cells, each in a column the row has a cell in and none reaching over a column boundary, and each follows the
        box above it in its column (``bottom``) at line spacing: less than three quarters of the white between the cells
        of two complete rows of this table (``ref_gap``) and at most a line height; at most 0.6 line heights where the
        table has no two complete rows in a row to compare with."""
        b = self.b
        if r.size >= base_size or not base_cols:
            return False
        for i, c in zip(r, col):
            if int(c) not in base_cols:
                return False
            if c < cuts.size and b[i, 2] > cuts[c] + 0.1 * self.h_med:
                return False
            gap = float(b[i, 1]) - bottom[int(c)]
            if gap > (1.0 if ref_gap is not None else 0.6) * self.h_med or (ref_gap is not None and gap >= 0.75 * ref_gap):
                return False
        return True

    def _is_text_column(self, idx: np.ndarray) -> bool:
        """A side of a vertical cut counts as a column of running text when it holds several lines that are much
        wider than tall.  Label / value pairs, right-aligned amounts and table cells fail this and stay in rows."""
        if idx.size < self.min_col_lines:
            return False
        w = self.b[idx, 2] - self.b[idx, 0]
        h = np.maximum(self.b[idx, 3] - self.b[idx, 1], 1.0)
        if float(np.median(w / h)) >= self.text_aspect:
            return True
        # a column that also holds a table: the many short cells pull the median down, the long lines are still there
        longs = int(np.count_nonzero(w / h >= 8.0))
        return longs >= self.min_col_lines

    def column_cuts(self, idx: np.ndarray) -> List[float]:
        g = [x for x in _gaps(self.b[idx, 0], self.b[idx, 2]) if x[0] >= self.tx]
        if not g:
            return []
        cuts = [0.5 * (s + e) for _, s, e in g]
        parts = _split_at(idx, self.b[:, 0], cuts)
        if len(parts) < 2 or not all(self._is_text_column(p) for p in parts):
            return []
        return cuts

    def cut(self, idx: np.ndarray, depth: int = 0) -> None:
        if idx.size == 0:
            return
        if idx.size == 1 or depth > 12:
            self.leaf(idx)
            return
        b = self.b
        # 0. a grid of cells is read row by row, whatever its columns look like
        if self.is_table(idx, 3):
            self.leaf(idx, True)
            return
        # 1. columns first: a gutter that runs through the whole region means the columns are read one after another
        cuts = self.column_cuts(idx)
        if cuts:
            for part in _split_at(idx, b[:, 0], cuts):
                self.cut(part, depth + 1)
            return
        # 2. horizontal whitespace at least a block gap tall: stacked blocks, top to bottom
        g = [x for x in _gaps(b[idx, 1], b[idx, 3]) if x[0] >= self.ty]
        if g:
            parts = _split_at(idx, b[:, 1], [0.5 * (s + e) for _, s, e in g])
            if len(parts) > 1:
                # Bands that together still split into columns stay together: the headings of two columns often sit
                # on one baseline with the same gap below them, and cutting there would read the headings across.
                i = 0
                while i < len(parts):
                    # the rows of a table whose row spacing is as large as a block gap belong together
                    j = i
                    while j + 1 < len(parts) and self.is_table(np.concatenate(parts[i : j + 2]), 2):
                        j += 1
                    if j > i and self.is_table(np.concatenate(parts[i : j + 1]), 3):
                        self.leaf(np.concatenate(parts[i : j + 1]), True)
                        i = j + 1
                        continue
                    j = i
                    while j + 1 < len(parts) and self.column_cuts(np.concatenate(parts[i : j + 2])):
                        j += 1
                    self.cut(np.concatenate(parts[i : j + 1]), depth + 1)
                    i = j + 1
                return
        # 3. a box spanning the region (full-width heading, wide caption) blocks the gutter: take spanning boxes
        #    as separators when the rest of the region does split into columns
        x0, x1 = float(b[idx, 0].min()), float(b[idx, 2].max())
        wide = idx[(b[idx, 2] - b[idx, 0]) >= 0.6 * (x1 - x0)]
        rest = idx[(b[idx, 2] - b[idx, 0]) < 0.6 * (x1 - x0)]
        if wide.size and rest.size >= 2 * self.min_col_lines and wide.size < idx.size and self.column_cuts(rest):
            wide = wide[np.argsort(b[wide, 1], kind="stable")]
            ys = [0.5 * (b[k, 1] + b[k, 3]) for k in wide]
            bands = np.searchsorted(np.asarray(ys), 0.5 * (b[rest, 1] + b[rest, 3]))
            for s in range(len(ys) + 1):
                self.cut(rest[bands == s], depth + 1)
                if s < len(ys):
                    self.leaf(wide[s : s + 1])
            return
        self.leaf_with_tables(idx)

    def tables_across_leaves(self) -> None:
        """Last pass over the finished reading sequence: rows of a table that the cuts left in different leaves (the
        first row glued to the paragraph above it, the others a block gap apart) are consecutive in the sequence all
        the same.  Consecutive rows of two or more cells, at most three line heights from one to the next (with the
        further lines of wrapped cells between them), that form a grid become one table block of their own."""
        b, order = self.b, self.order
        n = len(order)
        cy = lambda i: 0.5 * (b[i, 1] + b[i, 3])   # noqa: E731
        rows, s = [], 0
        for k in range(1, n + 1):
            if k == n or abs(cy(order[k]) - cy(order[s])) > 0.5 * self.h_med or b[order[k], 0] <= b[order[k - 1], 0]:
                rows.append((s, k))
                s = k
        i = 0
        while i < len(rows):
            if rows[i][1] - rows[i][0] < 2:
                i += 1
                continue
            follows = lambda j, lim: (b[order[rows[j][0]], 1] - b[order[rows[j - 1][0]], 3] <= lim * self.h_med    # noqa: E731
                                      and cy(order[rows[j][0]]) > cy(order[rows[j - 1][0]]))
            size = lambda j: rows[j][1] - rows[j][0]                                                               # noqa: E731
            j = i + 1
            while j < len(rows) and size(j) >= 2 and follows(j, 3.0):
                j += 1
            # the same run with the further lines of wrapped cells (single boxes at line spacing) counted in; tried
            # first, the run of complete rows alone after it
            e, full = i + 1, 1
            while e < len(rows) and follows(e, 3.0 if size(e) >= 2 else 1.0):
                full += size(e) >= 2
                e += 1
            for end, n_full in ((e, full), (j, j - i)) if e > j else ((j, j - i),):
                cells = np.asarray(order[rows[i][0] : rows[end - 1][1]], dtype=np.int64)
                if n_full >= 3 and not all(self.table_block[self.block_of[c]] for c in cells) and self.is_table(cells, 3):
                    self.table_block.append(True)
                    self.block_of[cells] = self.n_blocks
                    self.n_blocks += 1
                    j = end
                    break
            i = j

    def leaf_with_tables(self, idx: np.ndarray) -> None:
        """A region that no cut divides any further may still hold a table between its paragraphs (a table set close
        under a paragraph leaves no block gap): consecutive rows of two or more cells that form a grid become a table
        leaf, what lies above and below stays ordinary leaves, top to bottom."""
        rows = self.rows_of(idx)
        R = len(rows)
        tables = []
        i = 0
        while i < R:
            if rows[i].size < 2:
                i += 1
                continue
            j = i
            while j < R and rows[j].size >= 2:
                j += 1
            # with the lines of wrapped cells between its rows a table is a longer run: rows of fewer boxes that follow
            # the row above at line spacing count in (what comes of it is checked as a whole, then without them)
            e = i
            while e < R and (rows[e].size >= 2 or (e > i and self.b[rows[e], 1].min() - self.b[rows[e - 1], 3].max()
                                                   <= 1.0 * self.h_med)):
                e += 1
            if e > j and sum(1 for r in rows[i:e] if r.size >= 2) >= 3 and self.is_table(np.concatenate(rows[i:e]), 3):
                tables.append((i, e))
                i = e
                continue
            if j - i >= 3 and self.is_table(np.concatenate(rows[i:j]), 3):
                tables.append((i, j))
            i = j
        if not tables:
            self.leaf(idx)
            return
        at = 0
        for a, b in tables:
            if at < a:
                self.leaf(np.concatenate(rows[at:a]))
            self.leaf(np.concatenate(rows[a:b]), True)
            at = b
        if at < R:
            self.leaf(np.concatenate(rows[at:]))


def reading_order(boxes: np.ndarray, col_gap: float = 1.2, block_gap: float = 0.9, min_col_lines: int = 3,
                  text_aspect: float = 4.0, with_tables: bool = False):
    """Recursive XY-cut.  ``col_gap`` / ``block_gap`` are the minimum whitespace for a vertical / horizontal cut in
    units of the median box height.  Returns (order, block id per box) and, ``with_tables``, a bool per box: it is a
    cell of a table (the cells of one table share a block and come row by row)."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    n = boxes.shape[0]
    if n == 0:
        e = (np.zeros((0,), np.int64), np.zeros((0,), np.int32))
        return e + (np.zeros((0,), bool),) if with_tables else e
    c = _Cutter(boxes, col_gap, block_gap, min_col_lines, text_aspect)
    c.cut(np.arange(n))
    c.tables_across_leaves()
    if with_tables:
        return np.asarray(c.order, dtype=np.int64), c.block_of, np.asarray(c.table_block, bool)[c.block_of]
    return np.asarray(c.order, dtype=np.int64), c.block_of


def boxes_form_table(boxes: np.ndarray, min_rows: int = 3) -> bool:
    """Do these boxes, taken together, form one table (``_Cutter.is_table``)?"""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    if boxes.shape[0] < 2 * min_rows:
        return False
    return _Cutter(boxes, 1.2, 0.9, 3, 4.0).is_table(np.arange(boxes.shape[0]), min_rows)


def split_margins(boxes: np.ndarray, page_h: float, band: float = 0.075) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Indices of (header, body, footer) boxes.  A box is a running header / footer when it lies entirely inside
    the top / bottom ``band`` of the page and the body starts at least a line height away from it."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    n = boxes.shape[0]
    idx = np.arange(n)
    if n < 3 or page_h <= 0:
        return idx[:0], idx, idx[:0]
    h_med = float(np.median(boxes[:, 3] - boxes[:, 1]))
    top = boxes[:, 3] <= band * page_h
    bot = boxes[:, 1] >= (1.0 - band) * page_h
    body = ~(top | bot)
    if not body.any():
        return idx[:0], idx, idx[:0]
    if top.any() and boxes[body, 1].min() - boxes[top, 3].max() < h_med:
        body |= top
        top = np.zeros(n, bool)
    if bot.any() and boxes[bot, 1].min() - boxes[body, 3].max() < h_med:
        body |= bot
        bot = np.zeros(n, bool)
    return idx[top], idx[body], idx[bot]


def order_page_native(boxes: np.ndarray, page_h: float, figures: Optional[np.ndarray] = None
                      ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """``order_page`` through the native implementation (csrc/runtime/layout.cpp): the engine's hot path."""
    import ctypes

    from .batcher import lib

    l = lib()
    if not getattr(l, "_layout_declared", False):
        fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
        l.dlrt_layout_order.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, ctypes.c_float, ip, ip, ip]
        l.dlrt_layout_order.restype = ctypes.c_int
        l._layout_declared = True
    b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
    f = np.zeros((0, 4), np.float32) if figures is None else np.ascontiguousarray(figures, dtype=np.float32).reshape(-1, 4)
    n = b.shape[0] + f.shape[0]
    out = np.empty((3, max(n, 1)), dtype=np.int32)
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    k = l.dlrt_layout_order(b.ctypes.data_as(fp), b.shape[0], f.ctypes.data_as(fp), f.shape[0], float(page_h),
                            out[0].ctypes.data_as(ip), out[1].ctypes.data_as(ip), out[2].ctypes.data_as(ip))
    if k != n:
        raise RuntimeError(f"native layout returned {k} of {n} boxes")
    return out[0, :n].astype(np.int64), out[1, :n].copy(), out[2, :n].copy()


def order_page(boxes: np.ndarray, page_h: float, figures: Optional[np.ndarray] = None
               ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Full-page ordering: header lines, body in XY-cut order (figure boxes take part in the cut as obstacles and
    keep their place in the sequence), footer lines.

    Returns (order, block id, kind) where ``order`` indexes the concatenation [boxes ; figures] and ``kind`` is 0 for
    a text line, 1 for a figure and 2 for a cell of a table, aligned with ``order``."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    figs = np.zeros((0, 4)) if figures is None else np.asarray(figures, dtype=np.float64).reshape(-1, 4)
    nb = boxes.shape[0]
    head, body, foot = split_margins(boxes, page_h)
    allb = np.concatenate([boxes[body], figs], axis=0)
    order_b, block_b, cell_b = reading_order(allb, with_tables=True)
    order: List[int] = []
    block: List[int] = []
    nblk = 0
    for grp in (head,):
        o, _ = reading_order(boxes[grp])
        order += [int(grp[i]) for i in o]
        block += [nblk] * len(o)
        nblk += 1 if len(o) else 0
    for i in order_b:
        gi = int(body[i]) if i < body.size else nb + int(i - body.size)
        order.append(gi)
        block.append(nblk + int(block_b[i]))
    nblk += int(block_b.max()) + 1 if block_b.size else 0
    o, _ = reading_order(boxes[foot])
    order += [int(foot[i]) for i in o]
    block += [nblk] * len(o)
    order_a = np.asarray(order, dtype=np.int64)
    kind = (order_a >= nb).astype(np.int32)
    n_head = len(head)
    for k, i in enumerate(order_b):
        if cell_b[i] and i < body.size:
            kind[n_head + k] = 2
    return order_a, np.asarray(block, dtype=np.int32), kind
