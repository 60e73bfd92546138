"""Direct (all-links) all-reduce over HIP IPC peer buffers for one xGMI node.

RCCL stays the default collective backend; this is the hand-written two-phase
reduce-scatter + all-gather kernel, for the bucket sizes where a ring is
bound by one xGMI link (7 links x ~153 GB/s per GPU, point to point): every rank reads a
distinct 1/N shard from each peer at once, so all links carry traffic, and small buckets take a
one-shot path (every rank sums all N buffers itself; one barrier, latency bound).  The kernel is
``csrc/kernels/direct_allreduce.hip``; this module owns the bootstrap: each rank allocates its
buffers, exports them with ``hipIpcGetMemHandle``, the 128-byte handle pairs travel through the
process group (``all_gather_object``), and every rank maps its peers' buffers.

The reference has no collective backend at all; nothing to cite there.

Collective contract: every rank calls ``all_reduce_`` with the same shape / dtype / algo in the same
order.  Results are bitwise identical on every rank (fixed summation order, fp32 accumulation).
"""
from __future__ import annotations

import ctypes
import os
import socket
from typing import Optional

import torch
import torch.distributed as dist

MAX_WORLD = 8
ALGOS = {"auto": 0, "one_shot": 1, "two_shot": 2}


class DirectAllReduceError(RuntimeError):
    pass


def supported(t: torch.Tensor) -> bool:
    return (t.is_cuda and t.is_contiguous() and t.dtype in (torch.bfloat16, torch.float32)
            and (t.numel() * t.element_size()) % 16 == 0 and t.data_ptr() % 16 == 0)


class DirectAllReduce:
    """One per (process group, device).  ``max_bytes`` bounds a single call (one gradient bucket)."""

    def __init__(self, max_bytes: int, group: Optional[dist.ProcessGroup] = None, blocks: int = 128,
                 device: Optional[torch.device] = None):
        if not dist.is_initialized():
            raise DirectAllReduceError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if not 2 <= self.world <= MAX_WORLD:
            raise DirectAllReduceError(f"direct all-reduce serves 2..{MAX_WORLD} ranks of one node, got {self.world}")
        if not torch.cuda.is_available():
            raise DirectAllReduceError("needs a GPU (CPU groups use the process group's own all_reduce)")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.max_bytes = (int(max_bytes) + 15) // 16 * 16
        self.blocks = int(blocks)
        from ..ops import _loader

        self._l = _loader.lib()
        self._loader = _loader
        hosts = [None] * self.world
        dist.all_gather_object(hosts, (socket.gethostname(), self.max_bytes, self.blocks), group=group)
        if len({h[0] for h in hosts}) != 1:
            raise DirectAllReduceError("ranks span several hosts: HIP IPC buffers only reach GPUs of one node")
        if len({h[1:] for h in hosts}) != 1:
            raise DirectAllReduceError(f"ranks disagree on (max_bytes, blocks): {hosts}")
        ctx = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _loader.check(self._l.dlamd_dar_create(self.rank, self.world, self.max_bytes, self.blocks,
                                                   ctypes.byref(ctx)), "dlamd_dar_create")
            self._ctx = ctx
            mine = ctypes.create_string_buffer(128)
            _loader.check(self._l.dlamd_dar_export(ctx, mine), "dlamd_dar_export")
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(mine.raw), group=group)
            for p, h in enumerate(handles):
                if p == self.rank:
                    continue
                buf = ctypes.create_string_buffer(h, 128)
                rc = self._l.dlamd_dar_import(ctx, p, buf)
                if rc != 0:
                    raise DirectAllReduceError(f"rank {self.rank}: hipIpcOpenMemHandle of rank {p}'s buffers failed ({rc}); "
                                               "is HSA_ENABLE_IPC_MODE_LEGACY=0 exported before the first HIP call?")
        dist.barrier(group=group)   # nobody launches before every rank has mapped every peer
        self.calls = 0

    @property
    def flags_uncached(self) -> bool:
        return self._l.dlamd_dar_info(self._ctx, 0) == 1

    def all_reduce_(self, t: torch.Tensor, scale: float = 1.0, algo: str = "auto",
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Sum ``t`` over the group (times ``scale``) on the current HIP stream; in place unless ``out`` is given."""
        if not supported(t):
            raise DirectAllReduceError("tensor must be a contiguous 16-byte-aligned bf16/fp32 GPU tensor whose size is a "
                                       "multiple of 16 bytes")
        nbytes = t.numel() * t.element_size()
        if nbytes > self.max_bytes:
            raise DirectAllReduceError(f"{nbytes} bytes exceed this context's max_bytes={self.max_bytes}")
        dst = t if out is None else out
        if out is not None and (out.dtype != t.dtype or out.numel() != t.numel() or not supported(out)):
            raise DirectAllReduceError("out must match the input's dtype / size / alignment")
        rc = self._l.dlamd_dar_allreduce(self._ctx, self._loader.ptr(t), self._loader.ptr(dst), t.numel(),
                                         0 if t.dtype == torch.bfloat16 else 1, float(scale), ALGOS[algo],
                                         self._loader.stream_ptr())
        self._loader.check(rc, "dlamd_dar_allreduce")
        self.calls += 1
        return dst

    def set_spin_limit(self, polls: int) -> None:
        """Polls a barrier makes before it gives up on a peer (default 64 Mi, tens of seconds; env DLAMD_DAR_SPIN_LIMIT)."""
        self._loader.check(self._l.dlamd_dar_set_spin_limit(self._ctx, int(polls)), "dlamd_dar_set_spin_limit")

    def status(self) -> int:
        """Blocking: 0 ok; 1 / 2 = some peer never reached the first / second barrier of a call (bounded spin ran out)."""
        return int(self._l.dlamd_dar_status(self._ctx))

    def check(self) -> None:
        s = self.status()
        if s != 0:
            raise DirectAllReduceError(f"rank {self.rank}: a peer never arrived at barrier {s} (results of that call are invalid)")

    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None:
            try:
                dist.barrier(group=self.group)   # peers may still be reading this rank's buffers
            except Exception:
                pass
            self._l.dlamd_dar_destroy(self._ctx)
            self._ctx = None
