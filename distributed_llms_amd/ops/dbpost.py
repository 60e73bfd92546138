"""DB post-processing: probability map -> text boxes.

GPU path: ``dlamd_db_postprocess`` (csrc/kernels/db_postprocess.hip) -
union-find connected components, per-component statistics and closed-form
rectangle unclip, all on device; only the small box table crosses to the host.

CPU path / oracle: ``db_postprocess_reference`` (scipy.ndimage.label with
8-connectivity, same scoring / unclip arithmetic).

The reference stack does this with OpenCV findContours + pyclipper + shapely
(services/ocr_service/requirements.txt:160,200,271) inside magic-pdf.
Thresholds default to the published DBNet values (binarise 0.3, box score
0.6, unclip ratio 1.5); the values magic-pdf actually ships are not visible in the reference's files.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _loader


@dataclass
class DBPostConfig:
    bin_thresh: float = 0.3
    box_thresh: float = 0.6
    unclip_ratio: float = 1.5
    min_size: int = 3
    max_components: int = 4096
    max_boxes: int = 1024
    max_angle_deg: float = 20.0     # wide components tilted less than this get an oriented box (0: upright boxes only)


ANG_BINS = 401                       # 0.1 degree bins over +-20 degrees (db_postprocess.hip)
ANG_BIN_RAD = float(np.float32(0.0017453292))


def _axis(cnt, SX, SY, SXX, SYY, SXY):
    n = float(cnt)
    mx, my = SX / n, SY / n
    u20, u02, u11 = SXX / n - mx * mx, SYY / n - my * my, SXY / n - mx * my
    r = math.sqrt(0.25 * (u20 - u02) ** 2 + u11 * u11)
    return 0.5 * math.atan2(2.0 * u11, u20 - u02), 0.5 * (u20 + u02) + r, max(0.5 * (u20 + u02) - r, 0.0), mx, my


def _page_angle(stats, cfg: "DBPostConfig") -> float:
    """Mode of the width-weighted histogram of principal-axis angles of the wide components (angle_vote_kernel +
    angle_mode_kernel)."""
    max_angle = math.radians(cfg.max_angle_deg)
    hist = np.zeros(ANG_BINS, dtype=np.int64)
    for (xmin, ymin, xmax, ymax, cnt, psum, *mom) in stats:
        if cnt <= 0:
            continue
        w, h = float(xmax + 1 - xmin), float(ymax + 1 - ymin)
        if min(w, h) < cfg.min_size or w < 3 * h or np.float32(psum) / np.float32(cnt) < np.float32(cfg.box_thresh):
            continue
        th, l1, l2, _, _ = _axis(cnt, *mom)
        if abs(th) > max_angle or l1 < 64.0 * (l2 + 1.0 / 12.0):
            continue
        bn = int(np.rint(th / ANG_BIN_RAD)) + ANG_BINS // 2
        if 0 <= bn < ANG_BINS:
            hist[bn] += int(w)
    sm = hist.copy()
    sm[1:] += hist[:-1]
    sm[:-1] += hist[1:]
    if sm.max() <= 0:
        return 0.0
    k = int(np.argmax(sm))                      # first maximum = lowest bin on ties
    c0 = float(hist[k - 1]) if k > 0 else 0.0
    c1 = float(hist[k])
    c2 = float(hist[k + 1]) if k + 1 < ANG_BINS else 0.0
    return float((np.float32(k - ANG_BINS // 2) + np.float32(c2 - c0) / np.float32(c0 + c1 + c2)) * np.float32(ANG_BIN_RAD))


def _finalize(stats, scale, valid, cfg: DBPostConfig, H: int, W: int) -> np.ndarray:
    """Shared arithmetic of box_finalize_kernel for one page (numpy).  ``stats`` rows: xmin, ymin, xmax, ymax, count,
    sum(prob), sum x, sum y, sum x^2, sum y^2, sum xy."""
    out = []
    sx, sy = (np.float32(scale[0]), np.float32(scale[1])) if scale is not None else (np.float32(1), np.float32(1))
    vw, vh = (int(valid[0]), int(valid[1])) if valid is not None else (W, H)
    f32 = np.float32
    max_angle = math.radians(cfg.max_angle_deg)
    pa = _page_angle(stats, cfg) if max_angle > 0 else 0.0
    for (xmin, ymin, xmax, ymax, cnt, psum, SX, SY, SXX, SYY, SXY) in stats:
        if cnt <= 0:
            continue
        score = f32(psum) / f32(cnt)
        x0, y0, x1, y1 = f32(xmin), f32(ymin), f32(xmax + 1), f32(ymax + 1)
        w, h = x1 - x0, y1 - y0
        if min(w, h) < cfg.min_size or score < f32(cfg.box_thresh):
            continue
        angle = 0.0
        if max_angle > 0 and w >= 3 * h:
            th, l_major, l_minor, mx, my = _axis(cnt, SX, SY, SXX, SYY, SXY)
            own_ok = abs(th) <= max_angle and l_major >= 16.0 * (l_minor + 1.0 / 12.0)
            a = pa
            if own_ok and l_major >= 64.0 * (l_minor + 1.0 / 12.0) and abs(th - pa) > 0.0524:
                a = float(f32(th))
            if 0.004 <= abs(a) <= max_angle:
                angle = float(f32(a))
        if angle == 0.0:
            d = (w * h) * f32(cfg.unclip_ratio) / (f32(2.0) * (w + h))
            x0, y0 = max(x0 - d, f32(0)), max(y0 - d, f32(0))
            x1, y1 = min(x1 + d, f32(vw)), min(y1 + d, f32(vh))
            if x1 - x0 < cfg.min_size + 2 or y1 - y0 < cfg.min_size + 2:
                continue
            out.append([x0 * sx, y0 * sy, x1 * sx, y1 * sy, score, f32(cnt), f32(0), f32(0)])
            continue
        c, sg = f32(math.cos(angle)), f32(math.sin(angle))
        sa = abs(sg)
        c2 = c * c - sa * sa
        L, T = (w * c - h * sa) / c2, (h * c - w * sa) / c2
        L = min(max(L, f32(0.8) * f32(math.sqrt(12.0 * l_major))), f32(math.hypot(w, h)))
        T = min(max(T, max(f32(1), f32(0.8) * f32(math.sqrt(12.0 * l_minor + 1.0)))), h)
        if T < cfg.min_size:
            continue
        d = (L * T) * f32(cfg.unclip_ratio) / (f32(2.0) * (L + T))
        L, T = L + 2 * d, T + 2 * d
        if L < cfg.min_size + 2 or T < cfg.min_size + 2:
            continue
        ux, uy, vx, vy = c * sx, sg * sy, -sg * sx, c * sy
        Lp, Tp = L * f32(math.hypot(ux, uy)), T * f32(math.hypot(vx, vy))
        ap = f32(math.atan2(uy, ux))
        cxp, cyp = (f32(mx) + f32(0.5)) * sx, (f32(my) + f32(0.5)) * sy
        out.append([cxp - f32(0.5) * Lp, cyp - f32(0.5) * Tp, cxp + f32(0.5) * Lp, cyp + f32(0.5) * Tp, score, f32(cnt), ap, f32(0)])
    return np.asarray(out, dtype=np.float32).reshape(-1, 8)


def db_postprocess_reference(
    prob: torch.Tensor, mask: torch.Tensor, cfg: DBPostConfig,
    page_scale: Optional[torch.Tensor] = None, valid: Optional[torch.Tensor] = None,
) -> List[np.ndarray]:
    """prob [B,H,W] (any float), mask [B,H,W] u8 -> list of [n_i, 8] float32 box tables (x0, y0, x1, y1, score, pixel
    count, angle in radians, 0)."""
    from scipy import ndimage

    B, H, W = mask.shape
    pr = prob.float().cpu().numpy()
    mk = mask.cpu().numpy() > 0
    res = []
    structure = np.ones((3, 3), dtype=bool)
    for b in range(B):
        lab, n = ndimage.label(mk[b], structure=structure)
        stats = []
        if n > 0:
            objs = ndimage.find_objects(lab)
            idx = np.arange(1, n + 1)
            sums = ndimage.sum(pr[b], lab, index=idx)
            cnts = ndimage.sum(mk[b], lab, index=idx)
            yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
            mom = [ndimage.sum(v, lab, index=idx) for v in (xx, yy, xx * xx, yy * yy, xx * yy)]
            for i, sl in enumerate(objs):
                ys, xs = sl
                stats.append((xs.start, ys.start, xs.stop - 1, ys.stop - 1, int(cnts[i]), float(sums[i]),
                              *(float(m[i]) for m in mom)))
        sc = None if page_scale is None else page_scale[b].cpu().numpy()
        vd = None if valid is None else valid[b].cpu().numpy()
        res.append(_finalize(stats, sc, vd, cfg, H, W))
    return res


class DBPostWorkspace:
    """Device workspaces for a fixed (B, H, W); reused across batches (no allocs in the hot loop)."""

    def __init__(self, B: int, H: int, W: int, cfg: DBPostConfig, device: torch.device):
        self.B, self.H, self.W, self.cfg = B, H, W, cfg
        self.parent = torch.empty((B, H, W), dtype=torch.int32, device=device)
        self.comp_id = torch.empty((B, H, W), dtype=torch.int32, device=device)
        self.stats = torch.empty((B, cfg.max_components, 16), dtype=torch.int32, device=device)
        self.ncomp = torch.empty((B,), dtype=torch.int32, device=device)
        self.nbox = torch.empty((B,), dtype=torch.int32, device=device)
        self.boxes = torch.empty((B, cfg.max_boxes, 8), dtype=torch.float32, device=device)
        self.hist = torch.empty((B, ANG_BINS), dtype=torch.int32, device=device)        # page-skew votes
        self.page_angle = torch.zeros((B,), dtype=torch.float32, device=device)


def db_postprocess_gpu(
    prob: torch.Tensor, mask: torch.Tensor, ws: DBPostWorkspace,
    page_scale: Optional[torch.Tensor] = None, valid: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Launch the device post-process.  Returns (boxes [B,max_boxes,8], nbox [B]) device tensors."""
    B, H, W = mask.shape
    assert B <= ws.B and (H, W) == (ws.H, ws.W)
    assert prob.dtype == torch.bfloat16 and mask.dtype == torch.uint8
    assert prob.is_contiguous() and mask.is_contiguous()
    cfg = ws.cfg
    if page_scale is not None:
        assert page_scale.dtype == torch.float32 and page_scale.is_contiguous()
    if valid is not None:
        assert valid.dtype == torch.int32 and valid.is_contiguous()
    rc = _loader.lib().dlamd_db_postprocess(
        _loader.ptr(mask), _loader.ptr(prob), _loader.ptr(ws.parent), _loader.ptr(ws.comp_id),
        _loader.ptr(ws.stats), _loader.ptr(ws.ncomp), _loader.ptr(ws.nbox),
        _loader.ptr(page_scale), _loader.ptr(valid), _loader.ptr(ws.boxes),
        B, H, W, cfg.max_components, cfg.max_boxes, cfg.box_thresh, cfg.unclip_ratio, cfg.min_size,
        math.radians(cfg.max_angle_deg), _loader.ptr(ws.hist), _loader.ptr(ws.page_angle), _loader.stream_ptr(),
    )
    _loader.check(rc, "dlamd_db_postprocess")
    return ws.boxes, ws.nbox


def sort_boxes_reading_order(boxes: np.ndarray, line_tol: float = 0.5) -> np.ndarray:
    """Top-to-bottom, left-to-right.  Boxes whose vertical centres differ by less than
    ``line_tol`` x median height are treated as the same text row."""
    if boxes.shape[0] == 0:
        return boxes
    cy = 0.5 * (boxes[:, 1] + boxes[:, 3])
    h = np.median(boxes[:, 3] - boxes[:, 1])
    order = np.argsort(cy, kind="stable")
    boxes = boxes[order]
    cy = cy[order]
    rows, start = [], 0
    for i in range(1, len(boxes) + 1):
        if i == len(boxes) or cy[i] - cy[start] > line_tol * h:
            seg = boxes[start:i]
            rows.append(seg[np.argsort(seg[:, 0], kind="stable")])
            start = i
    return np.concatenate(rows, axis=0)


def db_postprocess(
    prob: torch.Tensor, mask: torch.Tensor, cfg: DBPostConfig,
    page_scale: Optional[torch.Tensor] = None, valid: Optional[torch.Tensor] = None,
    ws: Optional[DBPostWorkspace] = None,
) -> List[np.ndarray]:
    """Boxes per page as host arrays in reading order."""
    if not mask.is_cuda:
        res = db_postprocess_reference(prob, mask, cfg, page_scale, valid)
    else:
        B, H, W = mask.shape
        if ws is None:
            ws = DBPostWorkspace(B, H, W, cfg, mask.device)
        boxes, nbox = db_postprocess_gpu(prob, mask, ws, page_scale, valid)
        nb = nbox[:B].cpu().numpy()
        bx = boxes[:B].cpu().numpy()
        res = [bx[b, : min(int(nb[b]), cfg.max_boxes)].copy() for b in range(B)]
    return [sort_boxes_reading_order(r) for r in res]
