"""JPEG page-image decode.

GPU path (csrc/kernels/jpeg_decode.hip): a whole batch of baseline JPEGs per call.  Streams with restart markers
take one lane per restart segment (``dlamd_jpeg_decode``, two launches); streams without - what most scanners
write - are cut into 128-byte subsequences whose lanes find the code-word boundaries by self-synchronisation
(``dlamd_jpeg_sync`` / ``_sync_decode`` / ``_sync_finish`` with two prefix sums in between).  The host only parses
headers (native ``dljpeg_scan``, ~0.1 ms per page) and ships the *compressed* bytes: ~1 MB instead of 8.7 MB per
300-DPI A4 page over PCIe.

Host path / oracle: PIL (libjpeg-turbo), used for files the GPU kernels do not take (progressive, exotic sampling,
a large image in a handful of long restart segments) and as the numerics reference in tests.

The reference rasterises pages with PyMuPDF inside magic-pdf
(services/ocr_service/ocr_service.py:215).
"""
from __future__ import annotations

import io
from typing import List, Optional, Sequence

import numpy as np
import torch

from ..pdf.reader import JPEG_TABLES_BYTES, JpegScan
from . import _loader
from .scan import row_cumsum_i32

DESC_DTYPE = np.dtype(
    [
        ("data", "<u8"), ("seg_off", "<u8"), ("tables", "<u8"), ("coef", "<u8"), ("out", "<u8"),
        ("n_segments", "<i4"), ("restart_interval", "<i4"), ("mcus_x", "<i4"), ("total_mcus", "<i4"),
        ("ncomp", "<i4"), ("hs0", "<i4"), ("vs0", "<i4"), ("scan_end", "<i4"),
        ("Wb", "<i4"), ("Hb", "<i4"), ("pad0", "<i4"), ("pad1", "<i4"),
        ("dcdiff", "<u8"), ("sub_state", "<u8"), ("sub_count", "<u8"), ("sub_off", "<u8"),
        ("n_sub", "<i4"), ("scan_off", "<i4"), ("total_blocks", "<i4"), ("pad2", "<i4"),
    ],
    align=False,
)
assert DESC_DTYPE.itemsize == 136
SUB_BYTES = 128   # JPEG_SUB_BYTES in csrc/kernels/jpeg_core.h


def _a16(n: int) -> int:
    return (n + 15) & ~15


def decode_jpeg_host(stream) -> np.ndarray:
    """libjpeg-turbo via PIL -> uint8 [H, W] (gray) or [H, W, 3]."""
    from PIL import Image

    im = Image.open(io.BytesIO(bytes(stream)))
    if im.mode not in ("L", "RGB"):
        im = im.convert("L" if im.mode in ("1", "LA", "I;16") else "RGB")
    return np.asarray(im, dtype=np.uint8)


def decode_jpeg_batch_gpu(streams: Sequence[np.ndarray], scans: Sequence[JpegScan], device: torch.device
                          ) -> List[torch.Tensor]:
    """Decode baseline JPEGs on ``device``.  Returns gray uint8 [H, W] tensors (row-pitched views of MCU-padded
    planes).  Every scan must be ``gpu_decodable``; streams with and without restart markers may be mixed."""
    n = len(streams)
    if n == 0:
        return []
    sync = [k for k in range(n) if scans[k].self_synchronising]
    if not sync:
        return _decode_restart_segments(streams, scans, device)
    if len(sync) == n:
        return _decode_self_synchronising(streams, scans, device)
    rst = [k for k in range(n) if not scans[k].self_synchronising]
    res: List[Optional[torch.Tensor]] = [None] * n
    for idx, fn in ((rst, _decode_restart_segments), (sync, _decode_self_synchronising)):
        for k, t in zip(idx, fn([streams[k] for k in idx], [scans[k] for k in idx], device)):
            res[k] = t
    return res  # type: ignore[return-value]


def _decode_self_synchronising(streams: Sequence[np.ndarray], scans: Sequence[JpegScan], device: torch.device
                               ) -> List[torch.Tensor]:
    """Restart-less streams: synchronise (states at subsequence ends) -> prefix sum of block counts -> coefficient
    decode -> prefix sum of DC differences -> DC apply + IDCT.  Everything stays on the current stream; the host
    never waits."""
    n = len(streams)
    data_off, tab_offs = [], []
    off = 0
    for st, sc in zip(streams, scans):
        i = sc.info
        assert sc.self_synchronising, "stream has restart markers or is not decodable on the GPU"
        assert 0 < i.scan_len and int(i.scan_off + i.scan_len) <= st.size
        data_off.append(off)
        off += _a16(st.size + 16)
    for sc in scans:
        tab_offs.append(off)
        off += _a16(JPEG_TABLES_BYTES)
    desc_off = off
    off += _a16(n * DESC_DTYPE.itemsize)
    ptr_off = off
    off += 8 * n
    host = torch.empty((off,), dtype=torch.uint8, pin_memory=True)
    hb = host.numpy()
    geo, n_subs, n_lumas = [], [], []
    for k, (st, sc) in enumerate(zip(streams, scans)):
        i = sc.info
        hb[data_off[k] : data_off[k] + st.size] = st
        hb[data_off[k] + st.size : data_off[k] + _a16(st.size + 16)] = 0
        hb[tab_offs[k] : tab_offs[k] + JPEG_TABLES_BYTES] = sc.tables
        geo.append((i.mcus_x * i.hs[0], i.mcus_y * i.vs[0]))
        n_subs.append((int(i.scan_len) + SUB_BYTES - 1) // SUB_BYTES)
        n_lumas.append(i.mcus_x * i.mcus_y * i.hs[0] * i.vs[0])
    max_sub, max_luma = max(n_subs), max(n_lumas)
    sizes = [w * h * 64 for w, h in geo]
    coef = torch.zeros((sum(sizes),), dtype=torch.int16, device=device)
    out = torch.empty((sum(sizes),), dtype=torch.uint8, device=device)
    # [0] end states, [1] block counts, [2] exclusive prefix of the counts; rows padded with zeros
    sub = torch.zeros((3, n, max_sub), dtype=torch.int32, device=device)
    dc = torch.zeros((2, n, max_luma), dtype=torch.int32, device=device)      # [0] differences, [1] running sums
    dev = torch.empty((off,), dtype=torch.uint8, device=device)
    base = dev.data_ptr()
    descs = np.zeros(n, dtype=DESC_DTYPE)
    co = 0
    for k, sc in enumerate(scans):
        i = sc.info
        Wb, Hb = geo[k]
        total_mcus = i.mcus_x * i.mcus_y
        nblk = 1 if i.ncomp == 1 else i.hs[0] * i.vs[0] + 2
        descs[k] = (
            base + data_off[k], 0, base + tab_offs[k], coef.data_ptr() + 2 * co, out.data_ptr() + co,
            1, 0, i.mcus_x, total_mcus, i.ncomp, i.hs[0], i.vs[0], int(i.scan_off + i.scan_len), Wb, Hb, 0, 0,
            dc[0, k].data_ptr(), sub[0, k].data_ptr(), sub[1, k].data_ptr(), sub[2, k].data_ptr(),
            n_subs[k], int(i.scan_off), total_mcus * nblk, 0,
        )
        co += sizes[k]
    hb[desc_off : desc_off + n * DESC_DTYPE.itemsize] = descs.view(np.uint8)
    hb[ptr_off:] = np.asarray([dc[1, k].data_ptr() for k in range(n)], dtype=np.uint64).view(np.uint8)
    dev.copy_(host, non_blocking=True)
    l, vp = _loader.lib(), _loader.ctypes.c_void_p
    _loader.check(l.dlamd_jpeg_sync(vp(base + desc_off), n, max_sub, _loader.stream_ptr()), "dlamd_jpeg_sync")
    row_cumsum_i32(sub[1], sub[2], exclusive=True)
    _loader.check(l.dlamd_jpeg_sync_decode(vp(base + desc_off), n, max_sub, _loader.stream_ptr()), "dlamd_jpeg_sync_decode")
    row_cumsum_i32(dc[0], dc[1])
    _loader.check(l.dlamd_jpeg_sync_finish(vp(base + desc_off), vp(base + ptr_off), n, max_luma, max(w * h for w, h in geo),
                                           _loader.stream_ptr()), "dlamd_jpeg_sync_finish")
    res = []
    oo = 0
    # blocks the synchronised decode accounted for, per stream (one reduction for the whole call): equals
    # mcus * blocks_per_mcu for an intact stream; a truncated / damaged scan, or a seam pass that ran into its limit,
    # comes out short and the page holds garbage from there on (``blocks_short`` lets the caller find out)
    counted = sub[1].sum(dim=1, dtype=torch.int32)
    for k, sc in enumerate(scans):
        Wb, Hb = geo[k]
        plane = out[oo : oo + sizes[k]].view(Hb * 8, Wb * 8)
        t = plane[: sc.info.height, : sc.info.width]
        t._dlamd_keep = (dev, coef, sub, dc)  # type: ignore[attr-defined]
        t._dlamd_blocks = (counted, k, scans[k].info.mcus_x * scans[k].info.mcus_y * scans[k].info.blocks_per_mcu)  # type: ignore[attr-defined]
        res.append(t)
        oo += sizes[k]
    return res


def blocks_short(pages: Sequence[Optional[torch.Tensor]]):
    """For decoded pages of ONE ``decode_jpeg_batch_gpu`` call: (device int32 tensor of decoded block counts or None,
    [(position in ``pages``, row in that tensor, expected count)]).  The caller copies the tensor to the host with its
    other results and compares: a page whose count differs from the expectation was not decoded completely."""
    counted, rows = None, []
    for pos, t in enumerate(pages):
        info = getattr(t, "_dlamd_blocks", None) if t is not None else None
        if info is None:
            continue
        if counted is None:
            counted = info[0]
        if info[0] is counted:
            rows.append((pos, int(info[1]), int(info[2])))
    return counted, rows


def _decode_restart_segments(streams: Sequence[np.ndarray], scans: Sequence[JpegScan], device: torch.device
                             ) -> List[torch.Tensor]:
    n = len(streams)
    geo = []
    off = 0
    data_off, seg_offs, tab_offs = [], [], []
    for st, sc in zip(streams, scans):
        i = sc.info
        assert sc.gpu_decodable and not sc.self_synchronising, "stream is not decodable one lane per restart segment"
        assert int(i.scan_off + i.scan_len) <= st.size
        data_off.append(off)
        off += _a16(st.size + 16)
    for sc in scans:
        seg_offs.append(off)
        off += _a16(4 * sc.info.n_segments)
    for sc in scans:
        tab_offs.append(off)
        off += _a16(JPEG_TABLES_BYTES)
    desc_off = off
    off += n * DESC_DTYPE.itemsize
    host = torch.empty((off,), dtype=torch.uint8, pin_memory=True)
    hb = host.numpy()
    coef_sizes, out_sizes = [], []
    for k, (st, sc) in enumerate(zip(streams, scans)):
        i = sc.info
        hb[data_off[k] : data_off[k] + st.size] = st
        hb[data_off[k] + st.size : data_off[k] + _a16(st.size + 16)] = 0
        hb[seg_offs[k] : seg_offs[k] + 4 * i.n_segments] = sc.seg_offsets.view(np.uint8)
        hb[tab_offs[k] : tab_offs[k] + JPEG_TABLES_BYTES] = sc.tables
        Wb, Hb = i.mcus_x * i.hs[0], i.mcus_y * i.vs[0]
        geo.append((Wb, Hb))
        coef_sizes.append(Wb * Hb * 64)
        out_sizes.append(Wb * Hb * 64)
    coef = torch.zeros((sum(coef_sizes),), dtype=torch.int16, device=device)
    out = torch.empty((sum(out_sizes),), dtype=torch.uint8, device=device)
    dev = torch.empty((off,), dtype=torch.uint8, device=device)
    base = dev.data_ptr()
    descs = np.zeros(n, dtype=DESC_DTYPE)
    co = oo = 0
    max_seg, max_blk = 1, 1
    for k, sc in enumerate(scans):
        i = sc.info
        Wb, Hb = geo[k]
        descs[k] = (
            base + data_off[k], base + seg_offs[k], base + tab_offs[k], coef.data_ptr() + 2 * co, out.data_ptr() + oo,
            i.n_segments, i.restart_interval, i.mcus_x, i.mcus_x * i.mcus_y,
            i.ncomp, i.hs[0], i.vs[0], int(i.scan_off + i.scan_len), Wb, Hb, 0, 0,
            0, 0, 0, 0, 0, 0, 0, 0,
        )
        co += coef_sizes[k]
        oo += out_sizes[k]
        max_seg = max(max_seg, i.n_segments)
        max_blk = max(max_blk, Wb * Hb)
    hb[desc_off:] = descs.view(np.uint8)
    dev.copy_(host, non_blocking=True)
    rc = _loader.lib().dlamd_jpeg_decode(
        _loader.ctypes.c_void_p(base + desc_off), n, max_seg, max_blk, _loader.stream_ptr())
    _loader.check(rc, "dlamd_jpeg_decode")
    res = []
    oo = 0
    for k, sc in enumerate(scans):
        Wb, Hb = geo[k]
        plane = out[oo : oo + out_sizes[k]].view(Hb * 8, Wb * 8)
        res.append(plane[: sc.info.height, : sc.info.width])
        oo += out_sizes[k]
    # keep the staging blob and coefficient planes alive until the stream has consumed them
    for t in res:
        t._dlamd_keep = (dev, coef)  # type: ignore[attr-defined]
    return res
