"""ctypes bridge to the in-tree HIP kernel library (``libdlamd_kernels.so``).

The kernels expose a plain C ABI (``extern "C" int dlamd_*(..., hipStream_t)``)
and are launched on PyTorch-ROCm's *current* HIP stream, so they compose with
torch ops, stream pipelining and ``torch.cuda.graph`` capture (hipGraph).

On a machine with a GPU the library is mandatory: a missing / stale build
raises instead of silently falling back to eager PyTorch.
"""
from __future__ import annotations

import ctypes
import os
import threading
from pathlib import Path
from typing import Optional

import torch

_LIB: Optional[ctypes.CDLL] = None
_LOCK = threading.Lock()

LIB_DIR = Path(__file__).resolve().parents[1] / "_lib"
KERNEL_LIB = LIB_DIR / "libdlamd_kernels.so"


class KernelLibraryError(RuntimeError):
    pass


def kernels_available() -> bool:
    return KERNEL_LIB.exists()


def lib() -> ctypes.CDLL:
    """Load (once) and return the HIP kernel library."""
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LOCK:
        if _LIB is not None:
            return _LIB
        if not KERNEL_LIB.exists():
            raise KernelLibraryError(
                f"{KERNEL_LIB} is missing: run `python -m distributed_llms_amd.utils.build` "
                "(HIP kernels are mandatory on a GPU host; there is no eager fallback on device tensors)"
            )
        # torch is imported above, so libamdhip64.so.7 is already mapped and the
        # kernels share PyTorch's HIP runtime instance (same streams, same context).
        l = ctypes.CDLL(str(KERNEL_LIB), mode=ctypes.RTLD_GLOBAL)
        _declare(l)
        _LIB = l
    return _LIB


def stream_ptr() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def check(rc: int, name: str) -> None:
    if rc != 0:
        raise KernelLibraryError(f"{name} failed with code {rc} (host-side shape contract or launch error)")


def _declare(l: ctypes.CDLL) -> None:
    vp, i, f, ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong

    def sig(name, args):
        if hasattr(l, name):
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int

    sig("dlamd_conv2d", [vp, vp, vp, vp, vp, vp] + [i] * 20 + [f, i, vp, i, i, vp])
    sig("dlamd_gemm8_serves", [i] * 12)
    sig("dlamd_gemm8", [vp, vp, vp, vp, vp] + [i] * 18 + [vp])
    sig("dlamd_db_head_fused", [vp, vp, vp, vp, vp, vp, vp, i, i, i, f, vp])
    sig("dlamd_block_head_split", [vp, vp, vp, vp, vp, ll, vp])
    sig("dlamd_resample", [vp, i, vp, i, i, f, f, f, f, f, f, vp])
    sig("dlamd_resample_stem", [vp, i, i, i, f, f, f, f, f, f, vp, vp, vp, i, i, vp])
    sig("dlamd_db_postprocess", [vp] * 10 + [i, i, i, i, i, f, f, i, f, vp, vp, vp])
    sig("dlamd_row_argmax", [vp, i, vp, vp, ll, i, ll, vp])
    sig("dlamd_ctc_collapse", [vp, vp, vp, vp, vp, vp, vp, i, i, i, vp])
    sig("dlamd_lstm_bidir", [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, vp])
    if hasattr(l, "dlamd_lstm_set_variant"):
        l.dlamd_lstm_set_variant.argtypes = [i]
        l.dlamd_lstm_set_variant.restype = None
    sig("dlamd_layernorm", [vp, vp, vp, vp, vp, ll, i, f, i, vp])
    sig("dlamd_attention_fwd", [vp, vp, vp, vp, i, i, i, i, ll, ll, ll, ll, ll, ll, ll, ll, f, i, vp])
    sig("dlamd_attention_decode", [vp, ll, vp, vp, ll, ll, vp, vp, ll, vp, i, i, vp, ll, i, i, f, vp])
    sig("dlamd_ctc_loss", [vp, i, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, vp])
    sig("dlamd_adamw", [vp, vp, vp, vp, vp, ll, f, f, f, f, f, f, f, vp, vp])
    sig("dlamd_sumsq", [vp, ll, vp, vp])
    sig("dlamd_jpeg_decode", [vp, i, i, ll, vp])
    sig("dlamd_jpeg_sync", [vp, i, i, vp])
    sig("dlamd_jpeg_sync_decode", [vp, i, i, vp])
    sig("dlamd_jpeg_sync_finish", [vp, vp, i, ll, ll, vp])
    sig("dlamd_row_cumsum_i32", [vp, vp, vp, ll, i, i, i, vp])
    sig("dlamd_conv_wgrad", [vp, vp, vp] + [i] * 11 + [vp])
    sig("dlamd_bn_train_fwd", [vp] * 10 + [ll, i, f, f, i, vp])
    sig("dlamd_bn_train_bwd", [vp] * 9 + [ll, i, i, vp])
    sig("dlamd_lstm_train_fwd", [vp, vp, vp, vp, vp, vp, i, i, i, i, vp])
    sig("dlamd_lstm_train_bwd", [vp, vp, vp, vp, vp, vp, i, i, i, i, vp])
    sig("dlamd_dar_create", [i, i, ll, i, vp])
    sig("dlamd_dar_export", [vp, vp])
    sig("dlamd_dar_import", [vp, i, vp])
    sig("dlamd_dar_info", [vp, i])
    sig("dlamd_dar_status", [vp])
    sig("dlamd_dar_set_spin_limit", [vp, ll])
    sig("dlamd_dar_allreduce", [vp, vp, vp, ll, i, f, i, vp])
    sig("dlamd_dar_destroy", [vp])
    sig("dlamd_dar_alloc", [ll, vp])
    sig("dlamd_dar_free", [vp])
    sig("dlamd_dar_register", [vp, vp, ll, vp])
    sig("dlamd_dar_register_peer", [vp, i, i, vp])
    sig("dlamd_dar_allreduce_region", [vp, i, ll, i, f, i, vp])
    sig("dlamd_embed_rows", [vp, vp, vp, vp, vp, ll, i, i, i, i, f, vp])
    sig("dlamd_decode_bookkeep", [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, vp])
