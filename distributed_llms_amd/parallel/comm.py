"""Process-group bootstrap: one process per GPU, ``torch.distributed`` over RCCL/xGMI.

The reference has no collective backend at all (replicas only talk to
Redis/Mongo); this is new for recognizer training and for
the bench's cross-rank timing.  ``backend="nccl"`` IS RCCL on ROCm; CPU-only
hosts (CI, tests) use gloo with the same code path.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


@dataclass
class DistEnv:
    rank: int = 0
    world: int = 1
    local_rank: int = 0
    device: torch.device = torch.device("cpu")
    backend: str = "none"

    @property
    def is_main(self) -> bool:
        return self.rank == 0


def init_distributed(force_cpu: bool = False, timeout_s: int = 600) -> DistEnv:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract) and join the group."""
    import datetime

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required by RCCL on this driver
    use_gpu = torch.cuda.is_available() and not force_cpu
    if use_gpu:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
    backend = "none"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = "nccl" if use_gpu else "gloo"
        kw = {"device_id": device} if use_gpu else {}
        dist.init_process_group(backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout_s), **kw)
    elif dist.is_initialized():
        backend = dist.get_backend()
    return DistEnv(rank, world, local_rank, device, backend)


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def shutdown() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()


def all_reduce_max_scalar(v: float, device: Optional[torch.device] = None) -> float:
    if not dist.is_initialized():
        return v
    t = torch.tensor([v], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
