"""Bucketed, backward-overlapped gradient all-reduce for data-parallel training.

Design for MI355X: xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU), so a
collective is bound by per-link bandwidth and by launch latency, not by a switch.
Consequences taken here:

* gradients live in a few large *flat* buckets (default 64 MB: the whole CRNN is one
  or two buckets, a TrOCR-sized model ~20) - fewer, larger collectives;
* a bucket's collective is issued from the autograd hook of its last-arriving
  gradient, so communication of late layers overlaps the backward of early ones;
* ``mode="rs_ag"`` splits the all-reduce into reduce-scatter + all-gather, which maps
  to the direct (all-links) algorithm on a fully connected mesh and lets the optimizer
  run on a 1/N shard between the two halves (``ShardedFlatOptimizer``);
  ``mode="allreduce"`` leaves the algorithm choice to RCCL;
* ``mode="direct"`` runs the in-repo two-phase all-links kernel over HIP IPC peer buffers
  (parallel/direct.py, csrc/kernels/direct_allreduce.hip) on a side stream, with the 1/N
  averaging folded into the kernel; single node, 2..8 ranks, GPU only (CPU groups and
  larger worlds quietly use ``allreduce``).

Parameters and gradients are *views* into the flat buffers, so the fused AdamW kernel
(ops/optim.py) updates a whole bucket in one launch and no copy-in/copy-out happens.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


@dataclass
class Bucket:
    params: List[torch.nn.Parameter]
    flat_param: torch.Tensor
    flat_grad: torch.Tensor
    offsets: List[int]
    used: int = 0          # elements actually occupied by parameters (the rest is world-size padding)
    pending: int = 0
    handle: Optional[object] = None


def _align(n: int, a: int) -> int:
    return (n + a - 1) // a * a


class FlatBuckets:
    """Re-homes parameters (and their .grad) into contiguous flat buffers, bucketed in reverse
    registration order (= the order gradients become ready in backward)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, pad_to: int = 1,
                 grad_alloc=None):
        """``grad_alloc(numel, dtype)`` -> zero-filled flat tensor: where a bucket's gradient buffer comes from when it
        must live in memory of a particular kind (an IPC-exportable allocation for the direct all-reduce)."""
        plist = [p for p in params if p.requires_grad]
        assert plist, "no trainable parameters"
        dtype, device = plist[0].dtype, plist[0].device
        assert all(p.dtype == dtype and p.device == device for p in plist), "one dtype/device per FlatBuckets"
        self.dtype, self.device = dtype, device
        esz = plist[0].element_size()
        groups: List[List[torch.nn.Parameter]] = [[]]
        size = 0
        for p in reversed(plist):
            n = _align(p.numel(), 4)
            if groups[-1] and (size + n) * esz > bucket_bytes:
                groups.append([])
                size = 0
            groups[-1].append(p)
            size += n
        self.buckets: List[Bucket] = []
        self._bucket_of = {}
        for g in groups:
            offs, total = [], 0
            for p in g:
                offs.append(total)
                total += _align(p.numel(), 4)
            used = total
            total = _align(total, max(pad_to, 4))
            fp = torch.zeros(total, dtype=dtype, device=device)
            fg = torch.zeros(total, dtype=dtype, device=device) if grad_alloc is None else grad_alloc(total, dtype)
            assert fg.numel() == total and fg.dtype == dtype and fg.device == device
            for p, o in zip(g, offs):
                fp[o : o + p.numel()].copy_(p.data.reshape(-1))
                p.data = fp[o : o + p.numel()].view_as(p.data)
                p.grad = fg[o : o + p.numel()].view_as(p.data)
            b = Bucket(g, fp, fg, offs, used)
            for p, o in zip(g, offs):
                self._bucket_of[id(p)] = (len(self.buckets), o)   # keyed by identity: tensors compare elementwise
            self.buckets.append(b)

    def bucket_index(self, p: torch.nn.Parameter) -> int:
        return self._bucket_of[id(p)][0]

    def offset_of(self, p: torch.nn.Parameter) -> int:
        return self._bucket_of[id(p)][1]

    def zero_grad(self) -> None:
        for b in self.buckets:
            b.flat_grad.zero_()
            for p, o in zip(b.params, b.offsets):
                # autograd may have replaced .grad (e.g. after set_to_none); re-attach the view
                if p.grad is None or p.grad.data_ptr() != b.flat_grad.data_ptr() + o * b.flat_grad.element_size():
                    p.grad = b.flat_grad[o : o + p.numel()].view_as(p.data)

    def numel(self) -> int:
        return sum(b.flat_param.numel() for b in self.buckets)


class BucketedDDP:
    """Gradient averaging across the data-parallel group, overlapped with backward."""

    def __init__(self, module: torch.nn.Module, bucket_mb: float = 64.0, mode: str = "allreduce",
                 group: Optional[dist.ProcessGroup] = None, broadcast_params: bool = True,
                 register_buckets: bool = True):
        """``register_buckets`` (direct mode): the flat gradient buckets are allocated as IPC-exportable buffers and
        registered with the all-reduce, which then reduces them in place - the peers read a gradient from where
        autograd accumulated it.  False stages every bucket through the all-reduce's own buffers (one more
        device-to-device copy per bucket)."""
        assert mode in ("allreduce", "rs_ag", "direct")
        self.module = module
        self.mode = mode
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._direct = None
        self._comm_stream = None
        self.registered = False
        plist = [p for p in module.parameters() if p.requires_grad]
        device = plist[0].device if plist else torch.device("cpu")
        bucket_bytes, pad_to = int(bucket_mb * (1 << 20)), max(1, self.world) * 8
        if mode == "direct" and not (device.type == "cuda" and 2 <= self.world <= 8):
            self.mode = mode = "allreduce"
        if mode == "direct" and register_buckets:
            from .direct import DirectAllReduce

            self._direct = DirectAllReduce(4096, group=group, device=device)   # no staging buffers to speak of
            self.flat = FlatBuckets(module.parameters(), bucket_bytes, pad_to=pad_to, grad_alloc=self._direct.alloc)
            self._direct.register([b.flat_grad for b in self.flat.buckets])
            self.registered = True
        else:
            self.flat = FlatBuckets(module.parameters(), bucket_bytes, pad_to=pad_to)
            if mode == "direct":
                from .direct import DirectAllReduce

                big = max(b.flat_grad.numel() * b.flat_grad.element_size() for b in self.flat.buckets)
                self._direct = DirectAllReduce(big, group=group, device=self.flat.device)
        if self._direct is not None:
            self._comm_stream = torch.cuda.Stream(self.flat.device)
        self.comm_bytes = 0
        self.comm_calls = 0
        if self.world > 1 and broadcast_params:
            for b in self.flat.buckets:
                dist.broadcast(b.flat_param, src=0, group=group)
            for buf in module.buffers():
                if buf.is_floating_point():
                    dist.broadcast(buf, src=0, group=group)
        self._hooks = []
        self._in_backward = False
        if self.world > 1:
            for p in module.parameters():
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._reset()

    def _reset(self) -> None:
        for b in self.flat.buckets:
            b.pending = len(b.params)
            b.handle = None

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        b = self.flat.buckets[self.flat.bucket_index(p)]
        off = self.flat.offset_of(p)
        want = b.flat_grad.data_ptr() + off * b.flat_grad.element_size()
        if p.grad is not None and p.grad.data_ptr() != want:
            # autograd produced a fresh tensor instead of accumulating into our view: fold it back
            b.flat_grad[off : off + p.numel()].view_as(p.data).copy_(p.grad)
            p.grad = b.flat_grad[off : off + p.numel()].view_as(p.data)
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: Bucket) -> None:
        if self.mode == "direct":
            self._comm_stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self._comm_stream):
                if self.registered:
                    self._direct.all_reduce_region_(b.flat_grad, scale=1.0 / self.world)
                else:
                    self._direct.all_reduce_(b.flat_grad, scale=1.0 / self.world)
            self.comm_bytes += b.flat_grad.numel() * b.flat_grad.element_size()
            self.comm_calls += 1
            return
        b.flat_grad.div_(self.world)
        if self.mode == "allreduce":
            b.handle = dist.all_reduce(b.flat_grad, group=self.group, async_op=True)
            self.comm_bytes += b.flat_grad.numel() * b.flat_grad.element_size()
        else:
            n = b.flat_grad.numel() // self.world
            shard = b.flat_grad[self.rank * n : (self.rank + 1) * n]
            b.handle = dist.reduce_scatter_tensor(shard, b.flat_grad, group=self.group, async_op=True)
            self.comm_bytes += b.flat_grad.numel() * b.flat_grad.element_size() // 2
        self.comm_calls += 1

    def finish_backward(self, check: bool = True) -> None:
        """Wait for every bucket's collective (call after loss.backward(), before the optimizer).

        In ``direct`` mode the kernel's barriers are bounded spins: a peer that does not show up in time makes the
        kernel set a status word and leave, with sums over buffers the peer has not written.  ``check`` reads that
        word (one 4-byte device-to-host copy, i.e. a stream synchronisation) and raises ``DirectAllReduceError``, so
        a step never reaches the optimizer with unreduced gradients.  A loop that already synchronises every step
        (``loss.item()``) loses nothing; one that must not may pass ``check=False`` and call ``check_comm()`` at its
        own synchronisation points - but then has to discard the steps since the last check on an error."""
        if self.world > 1:
            for b in self.flat.buckets:
                if b.pending != 0:
                    # parameters that received no gradient this step (unused branch): reduce what we have
                    b.pending = 0
                    self._launch(b)
                if b.handle is not None:
                    b.handle.wait()
            if self._direct is not None:
                torch.cuda.current_stream(self.flat.device).wait_stream(self._comm_stream)
                if check:
                    self._reset()
                    self._direct.check()
        self._reset()

    def check_comm(self) -> None:
        """Raise if any direct all-reduce since the last check ran past a barrier timeout (see finish_backward)."""
        if self._direct is not None:
            self._direct.check()

    def shard_range(self, b: Bucket):
        n = b.flat_grad.numel() // max(1, self.world)
        return self.rank * n, (self.rank + 1) * n

    def all_gather_params(self) -> None:
        """Second half of rs_ag: after the optimizer updated this rank's shard of every bucket."""
        if self.world == 1 or self.mode != "rs_ag":
            return
        handles = []
        for b in self.flat.buckets:
            lo, hi = self.shard_range(b)
            handles.append(dist.all_gather_into_tensor(b.flat_param, b.flat_param[lo:hi].clone(), group=self.group,
                                                       async_op=True))
            self.comm_bytes += b.flat_param.numel() * b.flat_param.element_size() // 2
            self.comm_calls += 1
        for h in handles:
            h.wait()

    def zero_grad(self) -> None:
        self.flat.zero_grad()

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
