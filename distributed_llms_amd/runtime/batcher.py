"""ctypes binding of the native dynamic batcher (``csrc/runtime/batcher.cpp``).

``DynamicBatcher`` groups work items (pages for detection, crops for recognition) by a shape
key under a device-memory budget and a max-wait deadline.  The budget defaults to what a
288 GB MI355X leaves after weights and workspaces (``Settings.hbm_budget_gb``).
"""
from __future__ import annotations

import ctypes
import threading
from pathlib import Path
from typing import Dict, List, Optional, Tuple

LIB_PATH = Path(__file__).resolve().parents[1] / "_lib" / "libdlamd_runtime.so"
_lib: Optional[ctypes.CDLL] = None
_lock = threading.Lock()


def lib() -> ctypes.CDLL:
    global _lib
    with _lock:
        if _lib is None:
            if not LIB_PATH.exists():
                raise RuntimeError(f"{LIB_PATH} is missing: run `python -m distributed_llms_amd.utils.build`")
            l = ctypes.CDLL(str(LIB_PATH))
            u64, i64, i, vp = ctypes.c_uint64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
            l.dlrt_batcher_create.argtypes = [u64, i, i64]
            l.dlrt_batcher_create.restype = vp
            l.dlrt_batcher_destroy.argtypes = [vp]
            l.dlrt_batcher_destroy.restype = None
            l.dlrt_batcher_push.argtypes = [vp, u64, u64, u64]
            l.dlrt_batcher_close.argtypes = [vp]
            l.dlrt_batcher_close.restype = None
            l.dlrt_batcher_pop.argtypes = [vp, i, i64, ctypes.POINTER(u64), i, ctypes.POINTER(u64)]
            l.dlrt_batcher_pending.argtypes = [vp]
            l.dlrt_batcher_stats.argtypes = [vp, ctypes.POINTER(u64)]
            l.dlrt_batcher_stats.restype = None
            _lib = l
    return _lib


def shape_key(h: int, w: int) -> int:
    return (int(h) << 32) | int(w)


def page_cost_bytes(Hs: int, Ws: int, Hd: int, Wd: int, channels: int = 1) -> int:
    """Device bytes one page needs while in flight: source raster + detector activations
    (dominated by the 1/2- and 1/4-scale feature maps, ~150 B per detector input pixel in bf16)
    + post-process workspaces (2 x int32 + prob + mask per pixel)."""
    return Hs * Ws * channels + Hd * Wd * (16 + 150 + 11)


class DynamicBatcher:
    def __init__(self, max_items: int, max_wait_ms: float, budget_bytes: int = 0):
        self._h = lib().dlrt_batcher_create(int(budget_bytes), int(max_items), int(max_wait_ms * 1000))
        if not self._h:
            raise ValueError("invalid batcher parameters")
        self.max_items = max_items
        self._buf = (ctypes.c_uint64 * max_items)()
        self._items: Dict[int, object] = {}
        self._next = 1
        self._mu = threading.Lock()

    def push(self, key: int, payload: object, cost: int = 0) -> None:
        with self._mu:
            iid = self._next
            self._next += 1
            self._items[iid] = payload
        if lib().dlrt_batcher_push(self._h, key, iid, int(cost)) != 0:
            with self._mu:
                self._items.pop(iid, None)
            raise RuntimeError("batcher is closed")

    def pop(self, timeout_ms: float = -1.0, flush: bool = False, cap: int = 0) -> Tuple[Optional[int], List[object]]:
        """Next batch as (shape key, payloads); (None, []) on timeout or when closed and drained.  ``cap`` > 0 takes
        at most that many items (the rest of the bucket stays queued)."""
        key = ctypes.c_uint64()
        limit = min(self.max_items, cap) if cap > 0 else self.max_items
        n = lib().dlrt_batcher_pop(self._h, 1 if flush else 0, int(timeout_ms * 1000) if timeout_ms >= 0 else -1,
                                   self._buf, limit, ctypes.byref(key))
        if n <= 0:
            return None, []
        with self._mu:
            out = [self._items.pop(int(self._buf[k])) for k in range(n)]
        return int(key.value), out

    def close(self) -> None:
        lib().dlrt_batcher_close(self._h)

    def pending(self) -> int:
        return int(lib().dlrt_batcher_pending(self._h))

    def stats(self) -> Dict[str, int]:
        s = (ctypes.c_uint64 * 5)()
        lib().dlrt_batcher_stats(self._h, s)
        return {"pushed": s[0], "popped": s[1], "batches": s[2], "full_batches": s[3], "deadline_batches": s[4]}

    def __del__(self):
        try:
            if self._h:
                lib().dlrt_batcher_destroy(self._h)
                self._h = None
        except Exception:
            pass
