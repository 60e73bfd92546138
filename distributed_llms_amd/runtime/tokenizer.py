"""ctypes binding of the native byte-level BPE vocabulary (``csrc/runtime/bpe_tokenizer.cpp``).

The reference reaches the Rust ``tokenizers`` wheel through magic-pdf's recognizers
(services/ocr_service/requirements.txt:293).  Here the TrOCR-sized decoder's ids are turned into text
by in-repo C++: one call per recognised batch, all rows packed into a single byte buffer.

``ByteLevelVocab()`` is the built-in raw byte vocabulary (ids 0..3 specials, 4..259 bytes);
``ByteLevelVocab.from_files(vocab.json, merges.txt)`` loads a GPT-2 / RoBERTa style vocabulary.
"""
from __future__ import annotations

import ctypes
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np

from .batcher import lib as _runtime_lib

_bound = False


def _lib() -> ctypes.CDLL:
    global _bound
    l = _runtime_lib()
    if not _bound:
        vp, i, i64, cp = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_char_p
        l.dlbpe_create_bytes.restype = vp
        l.dlbpe_create.argtypes = [cp, i64, cp, i64]
        l.dlbpe_create.restype = vp
        l.dlbpe_destroy.argtypes = [vp]
        l.dlbpe_destroy.restype = None
        l.dlbpe_vocab_size.argtypes = [vp]
        l.dlbpe_special.argtypes = [vp, i]
        l.dlbpe_decode_batch.argtypes = [vp, vp, i, i, i, vp, i64, vp]
        l.dlbpe_decode_batch.restype = i64
        l.dlbpe_encode.argtypes = [vp, cp, i64, vp, i64]
        l.dlbpe_encode.restype = i64
        l.dlbpe_printable.argtypes = [cp, i64, vp, i64]
        l.dlbpe_printable.restype = i64
        _bound = True
    return l


class ByteLevelVocab:
    def __init__(self, _handle: Optional[int] = None):
        self._h = _handle if _handle is not None else _lib().dlbpe_create_bytes()
        if not self._h:
            raise ValueError("could not build the vocabulary")
        l = _lib()
        self.size = int(l.dlbpe_vocab_size(self._h))
        self.bos, self.pad, self.eos, self.unk = (int(l.dlbpe_special(self._h, k)) for k in range(4))

    @classmethod
    def from_bytes(cls, vocab_json: bytes, merges_txt: bytes = b"") -> "ByteLevelVocab":
        h = _lib().dlbpe_create(vocab_json, len(vocab_json), merges_txt or None, len(merges_txt))
        if not h:
            raise ValueError("vocab.json is not a {token: id} object")
        return cls(h)

    @classmethod
    def from_files(cls, vocab_json, merges_txt=None) -> "ByteLevelVocab":
        m = Path(merges_txt).read_bytes() if merges_txt and Path(merges_txt).exists() else b""
        return cls.from_bytes(Path(vocab_json).read_bytes(), m)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib().dlbpe_destroy(h)
            except Exception:
                pass

    def decode_batch(self, ids, stop_at_eos: bool = True) -> List[str]:
        """ids: [rows, cols] integer array / tensor (host). -> one string per row."""
        a = np.ascontiguousarray(np.asarray(ids), dtype=np.int32)
        if a.ndim == 1:
            a = a[None]
        rows, cols = a.shape
        if rows == 0:
            return []
        off = np.empty(rows + 1, dtype=np.int64)
        cap = max(64, rows * cols * 4)
        l = _lib()
        for _ in range(2):
            buf = ctypes.create_string_buffer(cap)
            n = l.dlbpe_decode_batch(self._h, a.ctypes.data, rows, cols, int(stop_at_eos), ctypes.addressof(buf), cap,
                                     off.ctypes.data)
            if n >= 0:
                raw = buf.raw[:n]
                return [raw[off[r]:off[r + 1]].decode("utf-8", "replace") for r in range(rows)]
            cap = -n - 1
        raise RuntimeError("decode buffer sizing failed")

    def decode(self, ids: Sequence[int], stop_at_eos: bool = True) -> str:
        return self.decode_batch(np.asarray(list(ids), dtype=np.int32)[None], stop_at_eos)[0]

    def encode(self, text: str, add_special: bool = False) -> List[int]:
        b = text.encode("utf-8")
        cap = max(16, len(b))
        out = np.empty(cap, dtype=np.int32)
        n = _lib().dlbpe_encode(self._h, b, len(b), out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("encode failed")
        ids = out[:n].tolist()
        return [self.bos] + ids + [self.eos] if add_special else ids


def printable(raw: bytes) -> str:
    """Raw bytes -> the printable byte alphabet used inside vocab.json / merges.txt."""
    cap = 4 * len(raw) + 8
    buf = ctypes.create_string_buffer(cap)
    n = _lib().dlbpe_printable(raw, len(raw), ctypes.addressof(buf), cap)
    return buf.raw[:n].decode("utf-8")
