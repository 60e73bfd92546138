"""Named shared-memory segments for handing a PDF from a front-end process to the GPU workers.

The reference pickles the whole PDF into a pool worker (services/ocr_service/ocr_service.py:274-278).  Here the upload
is *received* into a segment (``api/httpd.py`` reads the socket straight into it), the workers map it by name and the
native reader parses it in place; the bytes cross the process boundary without ever being copied.

``multiprocessing.shared_memory`` is not used: its segments are a ``memoryview`` (no ``find`` for the multipart scan),
and its resource tracker unlinks segments behind the back of processes that were not started through
``multiprocessing`` (front ends and workers are separate programs here).  A segment is a file in ``/dev/shm`` mapped
with ``mmap``: the creator unlinks it when the document is done, every user just closes its mapping.
"""
from __future__ import annotations

import itertools
import mmap
import os
import threading
from typing import Optional

SHM_DIR = os.environ.get("OCR_SHM_DIR", "/dev/shm")
_seq = itertools.count(1)
_lock = threading.Lock()


class Segment:
    """One mapping of a named segment.  ``buf`` is the ``mmap`` object: buffer protocol (``numpy.frombuffer``,
    ``memoryview``), slice assignment and ``find``."""

    __slots__ = ("name", "size", "buf", "owner")

    def __init__(self, name: str, size: int, buf: mmap.mmap, owner: bool):
        self.name, self.size, self.buf, self.owner = name, size, buf, owner

    @classmethod
    def create(cls, size: int, prefix: str = "dlamd") -> "Segment":
        size = max(1, int(size))
        while True:
            with _lock:
                name = f"{prefix}-{os.getpid()}-{next(_seq)}"
            path = os.path.join(SHM_DIR, name)
            try:
                fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            except FileExistsError:
                continue
            try:
                os.ftruncate(fd, size)
                buf = mmap.mmap(fd, size)
            except BaseException:
                os.close(fd)
                try:
                    os.unlink(path)
                except OSError:
                    pass
                raise
            os.close(fd)
            return cls(name, size, buf, True)

    @classmethod
    def attach(cls, name: str, size: Optional[int] = None) -> "Segment":
        if "/" in name or name.startswith("."):
            raise ValueError("bad segment name")
        fd = os.open(os.path.join(SHM_DIR, name), os.O_RDWR)
        try:
            n = os.fstat(fd).st_size if size is None else int(size)
            buf = mmap.mmap(fd, n)
        finally:
            os.close(fd)
        return cls(name, n, buf, False)

    def close(self) -> None:
        """Unmap.  Raises ``BufferError`` while an object (a numpy view, a memoryview) still exports the mapping."""
        if self.buf is not None:
            self.buf.close()
            self.buf = None

    def unlink(self) -> None:
        try:
            os.unlink(os.path.join(SHM_DIR, self.name))
        except OSError:
            pass

    def release(self) -> None:
        """Owner-side end of life: remove the name first (no new attach), then unmap if nothing views it any more."""
        if self.owner:
            self.unlink()
        try:
            self.close()
        except BufferError:
            pass          # a view outlives the job; the pages go back to the system when it is collected


class SegmentPool:
    """Recycles upload segments inside one front-end process.

    A fresh 120 MB tmpfs file costs ~30 000 page faults with zero-filling while the upload is received into it and the
    same again to free - a third of the receive path's CPU time.  A segment whose document finished cleanly is kept
    (up to ``max_bytes`` in total) and handed to the next upload that fits: its pages are already there.  Segments of
    failed or cancelled documents are never reused - a worker may still be reading them."""

    def __init__(self, max_bytes: int = 0, prefix: str = "dlamd"):
        self.max_bytes = int(max_bytes)
        self.prefix = prefix
        self.free: list = []
        self.held = 0
        self.mu = threading.Lock()
        self.reused = 0
        self.created = 0

    def acquire(self, size: int) -> Segment:
        size = max(1, int(size))
        with self.mu:
            best = None
            for k, sg in enumerate(self.free):
                if sg.size >= size and sg.size <= 2 * size + (1 << 20) and (best is None or sg.size < self.free[best].size):
                    best = k
            if best is not None:
                sg = self.free.pop(best)
                self.held -= sg.size
                self.reused += 1
                return sg
            self.created += 1
        # rounded up so that documents of about the same size can share it
        return Segment.create((size + (1 << 20) - 1) >> 20 << 20, self.prefix)

    def give_back(self, seg: Segment, reusable: bool) -> None:
        if reusable and seg.owner and seg.buf is not None:
            with self.mu:
                if self.held + seg.size <= self.max_bytes:
                    self.free.append(seg)
                    self.held += seg.size
                    return
        seg.release()

    def clear(self) -> None:
        with self.mu:
            segs, self.free, self.held = self.free, [], 0
        for sg in segs:
            sg.release()


def sweep_stale(prefix: str = "dlamd") -> int:
    """Remove segments whose creating process is gone (a front end that died mid-document).  Returns the count."""
    n = 0
    try:
        names = os.listdir(SHM_DIR)
    except OSError:
        return 0
    for name in names:
        parts = name.split("-")
        if len(parts) != 3 or parts[0] != prefix or not parts[1].isdigit():
            continue
        if os.path.exists(f"/proc/{parts[1]}"):
            continue
        try:
            os.unlink(os.path.join(SHM_DIR, name))
            n += 1
        except OSError:
            pass
    return n
