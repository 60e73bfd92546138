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

Two ways in: ``all_reduce_(tensor)`` stages any tensor through the context's own exported buffers (one
device-to-device copy per call), and ``alloc`` / ``register`` / ``all_reduce_region_`` reduce buffers the caller
allocated through this module IN PLACE - what ``BucketedDDP`` does with its flat gradient buckets, so a gradient is
read by the peers from where autograd wrote it.

Collective contract: every rank calls ``all_reduce_`` / ``register`` / ``all_reduce_region_`` with the same
shape / dtype / algo in the same order.  Results are bitwise identical on every rank (fixed summation order, fp32
accumulation).
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


class _IpcAllocation:
    """Device memory from ``hipMalloc`` (exportable as a whole through ``hipIpcGetMemHandle``, which allocations carved
    out of PyTorch's caching allocator are not), presented to ``torch.as_tensor`` through the array interface.  The
    tensor keeps this object alive; the memory is released with it."""

    def __init__(self, lib, nbytes: int, device: torch.device):
        self._lib = lib
        self.nbytes = int(nbytes)
        self.device = device
        p = ctypes.c_void_p()
        with torch.cuda.device(device):
            rc = lib.dlamd_dar_alloc(self.nbytes, ctypes.byref(p))
        if rc != 0 or not p.value:
            raise DirectAllReduceError(f"hipMalloc of {self.nbytes} bytes for a registered bucket failed ({rc})")
        self.ptr = int(p.value)

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2, "strides": None}

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", 0), 0
        if ptr:
            try:
                self._lib.dlamd_dar_free(ctypes.c_void_p(ptr))
            except Exception:  # noqa: BLE001 - interpreter shutdown: the driver reclaims the memory with the process
                pass


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
        self._regions = {}          # data_ptr of a registered tensor -> (region id, bytes, allocation)

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

    # ------------------------------------------------------------------ registered buffers (reduced in place)
    def alloc(self, numel: int, dtype: torch.dtype) -> torch.Tensor:
        """A zero-filled 1-d tensor in its own IPC-exportable allocation (size rounded up to 16 bytes)."""
        esz = torch.empty((), dtype=dtype).element_size()
        nbytes = (int(numel) * esz + 15) // 16 * 16
        a = _IpcAllocation(self._l, nbytes, self.device)
        raw = torch.as_tensor(a, device=self.device)
        if raw.data_ptr() != a.ptr:
            raise DirectAllReduceError("torch copied the IPC allocation instead of aliasing it")
        t = raw.view(dtype)[:numel]
        t._dlamd_ipc = a            # found again by register()
        return t

    def register(self, tensors) -> list:
        """Collective: export each tensor's allocation (from ``alloc``) and map every peer's.  -> region ids."""
        mine, ids = [], []
        with torch.cuda.device(self.device):
            for t in tensors:
                a = getattr(t, "_dlamd_ipc", None)
                if a is None or t.data_ptr() != a.ptr:
                    raise DirectAllReduceError("register() takes tensors returned by alloc()")
                h = ctypes.create_string_buffer(64)
                rid = self._l.dlamd_dar_register(self._ctx, ctypes.c_void_p(a.ptr), a.nbytes, h)
                if rid < 0:
                    raise DirectAllReduceError(f"rank {self.rank}: registering a {a.nbytes}-byte buffer failed ({rid})")
                ids.append(rid)
                mine.append((rid, a.nbytes, bytes(h.raw)))
                self._regions[a.ptr] = (rid, a.nbytes, a)
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine, group=self.group)
            if len({tuple((r, n) for r, n, _ in regs) for regs in everyone}) != 1:
                raise DirectAllReduceError("ranks registered different buffers (ids / sizes differ)")
            for p, regs in enumerate(everyone):
                if p == self.rank:
                    continue
                for rid, _, h in regs:
                    rc = self._l.dlamd_dar_register_peer(self._ctx, rid, p, ctypes.create_string_buffer(h, 64))
                    if rc != 0:
                        raise DirectAllReduceError(f"rank {self.rank}: mapping region {rid} of rank {p} failed ({rc})")
        dist.barrier(group=self.group)
        return ids

    def region_of(self, t: torch.Tensor) -> Optional[int]:
        r = self._regions.get(t.data_ptr())
        return None if r is None else r[0]

    def all_reduce_region_(self, t: torch.Tensor, scale: float = 1.0, algo: str = "auto") -> torch.Tensor:
        """Sum a registered tensor over the group in place, on the current HIP stream, without a staging copy."""
        r = self._regions.get(t.data_ptr())
        if r is None:
            raise DirectAllReduceError("tensor is not registered (alloc + register first)")
        if not supported(t) or t.numel() * t.element_size() > r[1]:
            raise DirectAllReduceError("registered tensor must be contiguous bf16/fp32 with a multiple of 16 bytes")
        rc = self._l.dlamd_dar_allreduce_region(self._ctx, r[0], t.numel(), 0 if t.dtype == torch.bfloat16 else 1,
                                                float(scale), ALGOS[algo], self._loader.stream_ptr())
        self._loader.check(rc, "dlamd_dar_allreduce_region")
        self.calls += 1
        return t

    def set_spin_limit(self, polls: int) -> None:
        """Polls a barrier makes before it gives up on a peer (default 64 Mi, tens of seconds; env DLAMD_DAR_SPIN_LIMIT)."""
        self._loader.check(self._l.dlamd_dar_set_spin_limit(self._ctx, int(polls)), "dlamd_dar_set_spin_limit")

    def status(self) -> int:
        """Blocking: 0 ok; 1 / 2 / 3 = some peer never reached that barrier of a call (bounded spin ran out)."""
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
            self._regions = {}      # the tensors keep their allocations; only the peers' mappings are gone
