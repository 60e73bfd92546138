"""JPEG page-image decode (hard part 7.4-7).

GPU path: ``dlamd_jpeg_decode`` (csrc/kernels/jpeg_decode.hip) decodes a whole
batch of baseline JPEGs in two launches, one lane per restart segment.  The
host only parses headers (native ``dljpeg_scan``, ~0.1 ms per page) and ships
the *compressed* bytes: ~1 MB instead of 8.7 MB per 300-DPI A4 page over PCIe.

Host path / oracle: PIL (libjpeg-turbo), used for files the GPU kernel does not
take (no restart markers, progressive, exotic sampling) and as the numerics
reference in tests.

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

DESC_DTYPE = np.dtype(
    [
        ("data", "<u8"), ("seg_off", "<u8"), ("tables", "<u8"), ("coef", "<u8"), ("out", "<u8"),
        ("n_segments", "<i4"), ("restart_interval", "<i4"), ("mcus_x", "<i4"), ("total_mcus", "<i4"),
        ("ncomp", "<i4"), ("hs0", "<i4"), ("vs0", "<i4"), ("scan_end", "<i4"),
        ("Wb", "<i4"), ("Hb", "<i4"), ("pad0", "<i4"), ("pad1", "<i4"),
    ],
    align=False,
)
assert DESC_DTYPE.itemsize == 88


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
    """Decode baseline JPEGs with restart markers on ``device``.  Returns gray uint8 [H, W] tensors
    (row-pitched views of MCU-padded planes).  Every scan must be ``gpu_decodable``."""
    n = len(streams)
    if n == 0:
        return []
    geo = []
    off = 0
    data_off, seg_offs, tab_offs = [], [], []
    for st, sc in zip(streams, scans):
        i = sc.info
        assert sc.gpu_decodable, "stream is not decodable by the GPU kernel (use decode_jpeg_host)"
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
