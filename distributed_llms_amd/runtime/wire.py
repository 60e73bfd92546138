"""Compact wire form of a shard's results (GPU worker -> front end).

The reference pickles a result dict back from its pool worker (services/ocr_service/ocr_service.py:274-278).  A scanned
page here carries ~40 recognised lines; as pickled ``PageResult`` / ``TextLine`` dataclasses that is ~45 Python objects
per page to build in the worker and rebuild in the front end - 0.4 M objects per second on an 8-GPU node, all under the
front end's interpreter lock.  On the wire a shard is ONE bytes object instead:

* flat arrays for the lines of all its pages (boxes, confidence, block, angle, kind, code-point offsets into one
  UTF-8 text blob);
* the content-list items of every page, already assembled in the worker (paragraph grouping needs the per-line
  objects, which only exist there): kind, bounding box, code-point offsets into a second text blob;
* figure crops as PNG bytes.

The front end builds the combined text and the content list from the item arrays (a few slices per page) and
materialises ``PageResult`` objects only for a caller that asks for them (``LazyPages``).
"""
from __future__ import annotations

import io
import json
import struct
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

MAGIC = b"DLS1"
ITEM_TEXT, ITEM_EQUATION, ITEM_IMAGE, ITEM_TABLE, ITEM_MARGIN_TOP, ITEM_MARGIN_BOTTOM = 0, 1, 2, 3, 4, 5
_LINE_KINDS = ("text", "equation", "cell")


def _png(arr: np.ndarray) -> bytes:
    from PIL import Image

    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="PNG", compress_level=1)
    return buf.getvalue()


def pack_shard(pages: Sequence[Tuple[int, Optional[object]]], items_of=None) -> bytes:
    """``pages``: [(page index, PageResult or None)] in page order.  ``items_of(page_result)`` -> content-list items of
    one page (``runtime.document.page_content_items``); figure items refer to ``page.figures`` by position."""
    if items_of is None:
        from .document import page_content_items as items_of
    P = len(pages)
    idx = np.zeros((P,), np.int32)
    meta = np.zeros((P, 5), np.int32)            # width, height, n_detected, used_prior, present
    line_off = np.zeros((P + 1,), np.int32)
    item_off = np.zeros((P + 1,), np.int32)
    fig_off = np.zeros((P + 1,), np.int32)
    boxes: List[Tuple[float, float, float, float]] = []
    conf: List[float] = []
    block: List[int] = []
    angle: List[float] = []
    kind: List[int] = []
    texts: List[str] = []
    i_kind: List[int] = []
    i_box: List[Tuple[float, float, float, float]] = []
    i_fig: List[int] = []
    i_texts: List[str] = []
    f_box: List[Tuple[float, float, float, float]] = []
    f_order: List[int] = []
    f_png: List[bytes] = []
    errs: List[str] = []
    for k, (i, pr) in enumerate(pages):
        idx[k] = i
        errs.append((getattr(pr, "error", None) or "") if pr is not None else "")
        if pr is not None:
            meta[k] = (pr.width, pr.height, pr.n_detected, int(pr.used_layout_prior), 1)
            for ln in pr.lines:
                boxes.append(ln.box)
                conf.append(ln.confidence)
                block.append(ln.block)
                angle.append(ln.angle)
                kind.append(_LINE_KINDS.index(ln.kind) if ln.kind in _LINE_KINDS else 0)
                texts.append(ln.text)
            base = len(f_png)
            for fg in pr.figures:
                f_box.append(fg.box)
                f_order.append(fg.order)
                f_png.append(_png(fg.image) if fg.image is not None else b"")
            for it in items_of(pr):
                t = it["type"]
                if t == "table":
                    # cells never hold a tab or a line break: rows on lines, cells between tabs
                    i_kind.append(ITEM_TABLE)
                    i_texts.append("".join(it.get("table_caption") or []) + "\f" +
                                   "\n".join("\t".join(c.replace("\t", " ").replace("\n", " ") for c in row)
                                             for row in it["cells"]))
                elif t == "text" and it.get("margin"):
                    i_kind.append(ITEM_MARGIN_TOP if it["margin"] == "top" else ITEM_MARGIN_BOTTOM)
                    i_texts.append(it["text"])
                else:
                    i_kind.append(ITEM_IMAGE if t == "image" else (ITEM_EQUATION if t == "equation" else ITEM_TEXT))
                    i_texts.append("".join(it.get("img_caption") or []) if t == "image" else it.get("text", ""))
                i_box.append(tuple(it.get("bbox") or (0.0, 0.0, 0.0, 0.0)))
                i_fig.append(base + int(it.get("figure", 0)) if t == "image" else -1)
        line_off[k + 1] = len(texts)
        item_off[k + 1] = len(i_kind)
        fig_off[k + 1] = len(f_png)

    def offsets(strs: List[str]) -> np.ndarray:
        o = np.zeros((len(strs) + 1,), np.int32)
        if strs:
            np.cumsum([len(s) for s in strs], out=o[1:])
        return o

    png_off = np.zeros((len(f_png) + 1,), np.int64)
    if f_png:
        np.cumsum([len(b) for b in f_png], out=png_off[1:])
    sections: Dict[str, object] = {
        "idx": idx, "meta": meta, "line_off": line_off, "item_off": item_off, "fig_off": fig_off,
        "boxes": np.asarray(boxes, np.float32).reshape(-1, 4), "conf": np.asarray(conf, np.float32),
        "block": np.asarray(block, np.int32), "angle": np.asarray(angle, np.float32), "kind": np.asarray(kind, np.uint8),
        "text_off": offsets(texts), "text": "".join(texts).encode("utf-8", "surrogatepass"),
        "i_kind": np.asarray(i_kind, np.uint8), "i_box": np.asarray(i_box, np.float32).reshape(-1, 4),
        "i_fig": np.asarray(i_fig, np.int32), "i_text_off": offsets(i_texts),
        "i_text": "".join(i_texts).encode("utf-8", "surrogatepass"),
        "f_box": np.asarray(f_box, np.float32).reshape(-1, 4), "f_order": np.asarray(f_order, np.int32),
        "png_off": png_off, "png": b"".join(f_png),
        "err_off": offsets(errs), "err": "".join(errs).encode("utf-8", "surrogatepass"),
    }
    head: Dict[str, list] = {}
    blobs: List[bytes] = []
    pos = 0
    for name, v in sections.items():
        if isinstance(v, np.ndarray):
            raw = np.ascontiguousarray(v).tobytes()
            head[name] = [v.dtype.str, list(v.shape), pos, len(raw)]
        else:
            raw = v
            head[name] = ["bytes", [len(raw)], pos, len(raw)]
        pad = (-len(raw)) % 8
        blobs.append(raw)
        if pad:
            blobs.append(b"\0" * pad)
        pos += len(raw) + pad
    hj = json.dumps(head, separators=(",", ":")).encode()
    hj += b" " * ((-(len(hj) + 8)) % 8)
    return b"".join([MAGIC, struct.pack("<I", len(hj)), hj] + blobs)


class PackedShard:
    """Read side of ``pack_shard``: numpy views over the received bytes, text blobs decoded on first use."""

    def __init__(self, data: bytes):
        if data[:4] != MAGIC:
            raise ValueError("not a packed shard")
        (hl,) = struct.unpack_from("<I", data, 4)
        head = json.loads(bytes(data[8 : 8 + hl]))
        base = 8 + hl
        self._raw = data
        self._sec: Dict[str, object] = {}
        for name, (dt, shape, off, nb) in head.items():
            if dt == "bytes":
                self._sec[name] = memoryview(data)[base + off : base + off + nb]
            else:
                n = int(np.prod(shape)) if shape else 1
                self._sec[name] = np.frombuffer(data, dtype=np.dtype(dt), count=n, offset=base + off).reshape(shape)
        self._text: Optional[str] = None
        self._itext: Optional[str] = None

    def __getattr__(self, name):
        try:
            return self.__dict__["_sec"][name]
        except KeyError:
            raise AttributeError(name) from None

    @property
    def n_pages(self) -> int:
        return int(self.idx.shape[0])

    @property
    def n_lines(self) -> int:
        return int(self.line_off[-1])

    def page_indices(self) -> List[int]:
        return [int(v) for v in self.idx]

    def present(self, k: int) -> bool:
        return bool(self.meta[k, 4])

    def _t(self) -> str:
        if self._text is None:
            self._text = bytes(self._sec["text"]).decode("utf-8", "surrogatepass")
        return self._text

    def _it(self) -> str:
        if self._itext is None:
            self._itext = bytes(self._sec["i_text"]).decode("utf-8", "surrogatepass")
        return self._itext

    def page_error(self, k: int) -> Optional[str]:
        """Why page slot ``k`` carries no text although it had an image (a damaged image stream), or None."""
        a, b = int(self.err_off[k]), int(self.err_off[k + 1])
        if a == b:
            return None
        return bytes(self._sec["err"]).decode("utf-8", "surrogatepass")[a:b]

    def page_confidences(self, k: int) -> np.ndarray:
        """Recogniser confidence of the text lines and table cells of page slot ``k`` (formulas left out)."""
        a, e = int(self.line_off[k]), int(self.line_off[k + 1])
        kd = np.asarray(self.kind[a:e])
        return np.asarray(self.conf[a:e])[(kd == _LINE_KINDS.index("text")) | (kd == _LINE_KINDS.index("cell"))]

    def figure_png(self, f: int) -> bytes:
        o = self.png_off
        return bytes(self._sec["png"][int(o[f]) : int(o[f + 1])])

    def items(self, k: int, page_idx: Optional[int] = None, fig_paths: Optional[Dict[int, str]] = None) -> List[dict]:
        """Content-list items of page slot ``k`` (``runtime.document.build_content_list`` shape)."""
        a, b = int(self.item_off[k]), int(self.item_off[k + 1])
        if a == b:
            return []
        t, off = self._it(), self.i_text_off
        pi = int(self.idx[k]) if page_idx is None else page_idx
        out = []
        for j in range(a, b):
            kd = int(self.i_kind[j])
            if kd == ITEM_IMAGE:
                f = int(self.i_fig[j])
                cap = t[off[j] : off[j + 1]]
                out.append({"type": "image", "img_path": (fig_paths or {}).get(f, ""), "img_caption": [cap] if cap else [],
                            "img_footnote": [], "page_idx": pi, "bbox": [round(float(v), 1) for v in self.i_box[j]]})
            elif kd == ITEM_EQUATION:
                out.append({"type": "equation", "text": t[off[j] : off[j + 1]], "text_format": "latex", "page_idx": pi,
                            "bbox": [round(float(v), 1) for v in self.i_box[j]]})
            elif kd == ITEM_TABLE:
                from .document import table_item

                cap, _, body = t[off[j] : off[j + 1]].partition("\f")
                tab = table_item([r.split("\t") for r in body.split("\n")], pi, self.i_box[j])
                if cap:
                    tab["table_caption"] = [cap]
                out.append(tab)
            elif kd in (ITEM_MARGIN_TOP, ITEM_MARGIN_BOTTOM):
                out.append({"type": "text", "text": t[off[j] : off[j + 1]], "page_idx": pi,
                            "margin": "top" if kd == ITEM_MARGIN_TOP else "bottom"})
            else:
                out.append({"type": "text", "text": t[off[j] : off[j + 1]], "page_idx": pi})
        return out

    def page_result(self, k: int, doc_id: str = ""):
        """Materialise page slot ``k`` as a ``PageResult`` (None for a page that had no image)."""
        from .engine import Figure, PageResult, TextLine

        if not self.present(k):
            return None
        w, h, nd, up, _ = (int(v) for v in self.meta[k])
        pr = PageResult(doc_id, int(self.idx[k]), w, h, n_detected=nd, used_layout_prior=bool(up), error=self.page_error(k))
        t, off = self._t(), self.text_off
        bx, cf, bl, an, kd = self.boxes, self.conf, self.block, self.angle, self.kind
        for j in range(int(self.line_off[k]), int(self.line_off[k + 1])):
            pr.lines.append(TextLine(tuple(float(v) for v in bx[j]), t[off[j] : off[j + 1]], float(cf[j]), int(bl[j]),
                                     float(an[j]), _LINE_KINDS[int(kd[j])] if int(kd[j]) < len(_LINE_KINDS) else "text"))
        for f in range(int(self.fig_off[k]), int(self.fig_off[k + 1])):
            pr.figures.append(Figure(tuple(float(v) for v in self.f_box[f]), int(self.f_order[f]), None, ""))
        return pr


class LazyPages:
    """``DocumentResult.pages`` of a document that came back packed: a sequence of ``PageResult`` built on access."""

    def __init__(self, slots: List[Tuple[Optional[PackedShard], int, Optional[object]]], doc_id: str):
        # slots: per page in page order, (shard, slot in shard, None) or (None, -1, ready PageResult)
        self._slots = slots
        self._doc_id = doc_id
        self._cache: Dict[int, object] = {}

    def __len__(self) -> int:
        return len(self._slots)

    @property
    def n_lines(self) -> int:
        """Recognised lines of the whole document, counted without materialising a page."""
        return sum(len(ready.lines) if sh is None else int(sh.line_off[k + 1]) - int(sh.line_off[k])
                   for sh, k, ready in self._slots)

    def _get(self, i: int):
        pr = self._cache.get(i)
        if pr is None:
            sh, k, ready = self._slots[i]
            pr = ready if sh is None else sh.page_result(k, self._doc_id)
            self._cache[i] = pr
        return pr

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._get(j) for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._get(i)

    def __iter__(self):
        return (self._get(i) for i in range(len(self)))
