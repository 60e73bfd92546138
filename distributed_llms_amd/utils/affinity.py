"""CPU placement of the per-GPU worker processes.

Every GPU worker keeps a decode pool, an engine thread and a result packer busy.  Left to the scheduler, the threads of
eight workers wander over both sockets of the host and every JPEG byte crosses the inter-socket link on its way to a
GPU that hangs off the other socket.  A worker therefore pins itself (and with it every thread it starts later) to the
cores next to its GPU: the ``local_cpulist`` of the GPU's PCI device when sysfs shows one, else an even slice of the
cores this process may use.  The reference has nothing to place (one CPU process pool per VM,
services/ocr_service/ocr_service.py:42); this is part of what replaces its VM-level scaling inside a node.
"""
from __future__ import annotations

import glob
import os
from typing import List, Optional, Sequence


def parse_cpulist(text: str) -> List[int]:
    """``"0-3,8,10-11"`` -> [0, 1, 2, 3, 8, 10, 11]."""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def allowed_cpus() -> List[int]:
    if hasattr(os, "sched_getaffinity"):
        return sorted(os.sched_getaffinity(0))
    return list(range(os.cpu_count() or 1))


def gpu_pci_cpulists(root: str = "/sys/class/drm") -> List[List[int]]:
    """``local_cpulist`` of every AMD GPU render node, ordered by PCI address (the order HIP enumerates devices in when
    no visibility mask reorders them).  Empty when sysfs shows none."""
    found = []
    for card in sorted(glob.glob(os.path.join(root, "renderD*"))):
        dev = os.path.join(card, "device")
        try:
            with open(os.path.join(dev, "vendor")) as f:
                if f.read().strip().lower() != "0x1002":
                    continue
            with open(os.path.join(dev, "local_cpulist")) as f:
                cpus = parse_cpulist(f.read())
            found.append((os.path.basename(os.path.realpath(dev)), cpus))
        except (OSError, ValueError):
            continue
    found.sort()
    return [c for _, c in found]


def cpus_for_worker(index: int, n_workers: int, allowed: Optional[Sequence[int]] = None,
                    gpu_lists: Optional[Sequence[Sequence[int]]] = None) -> List[int]:
    """Cores for worker ``index`` of ``n_workers``.

    With sysfs locality: the cores local to GPU ``index`` that this process may use, split evenly among the workers
    whose GPUs share those cores (GPUs 0-3 on socket 0 each get a quarter of it).  Without: slice ``index`` of the
    allowed cores.  Never empty: falls back to every allowed core."""
    allowed = sorted(allowed if allowed is not None else allowed_cpus())
    n_workers = max(1, n_workers)
    lists = gpu_pci_cpulists() if gpu_lists is None else [list(c) for c in gpu_lists]
    if lists and index < len(lists) and len(lists) >= n_workers:
        mine = [c for c in lists[index] if c in set(allowed)]
        if mine:
            peers = [w for w in range(n_workers) if w < len(lists) and set(lists[w]) == set(lists[index])]
            k, n = peers.index(index), len(peers)
            per = len(mine) // n
            if per >= 1:
                return mine[k * per : (k + 1) * per] if k < n - 1 else mine[k * per :]
            return mine
    per = len(allowed) // n_workers
    if per >= 1:
        k = index % n_workers
        return allowed[k * per : (k + 1) * per] if k < n_workers - 1 else allowed[k * per :]
    return allowed


def pin_process(cpus: Sequence[int]) -> bool:
    """Restrict the calling thread, and every thread it starts from now on, to ``cpus`` (Linux affinity is per thread
    and inherited at creation: call this first thing in a worker's main thread).  False when not possible."""
    if not cpus or not hasattr(os, "sched_setaffinity"):
        return False
    try:
        os.sched_setaffinity(0, set(int(c) for c in cpus))
        return True
    except OSError:
        return False
