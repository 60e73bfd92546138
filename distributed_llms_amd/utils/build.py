"""In-tree native build driver.

Every native component of the framework is compiled here, explicitly, for
gfx950 only:

* ``csrc/kernels/*.hip``  -> ``distributed_llms_amd/_lib/libdlamd_kernels.so``
  (hand-written HIP/CDNA4 kernels, C ABI, loaded with ctypes so that the
  kernels launch on the HIP stream PyTorch-ROCm is currently using)
* ``csrc/pdf/*.cpp``      -> ``distributed_llms_amd/_lib/libdlamd_pdf.so``
  (C++ PDF reader / classifier / JPEG tools; host only)
* ``csrc/runtime/*.cpp``  -> ``distributed_llms_amd/_lib/libdlamd_runtime.so``
  (C++ batcher, shared-memory page ring, host thread pool)

The shared objects live in-tree so that they travel with a source snapshot to
a GPU box (nothing is JIT-compiled at import time).  ``hipcc`` cross-compiles
without a GPU, so ``build_all()`` is also the CPU-side "does it build" check
used by ``__graft_entry__.build``.

The reference has no native code of its own; these libraries
replace the wheels it pulls in (PyMuPDF, OpenCV, pyclipper, torch-CPU ops).
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Dict, List, Sequence

REPO_ROOT = Path(__file__).resolve().parents[2]
CSRC = REPO_ROOT / "csrc"
LIB_DIR = REPO_ROOT / "distributed_llms_amd" / "_lib"
OBJ_DIR = REPO_ROOT / "build" / "obj"
GPU_ARCH = "gfx950"

HIPCC = os.environ.get("HIPCC", shutil.which("hipcc") or "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", shutil.which("g++") or "g++")

HIP_FLAGS = [
    f"--offload-arch={GPU_ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffast-math",
    "-fno-gpu-sanitize",
    "-Wno-unused-result",
    "-DNDEBUG",
    # no packed-FP32 VALU forms (v_pk_add/mul/fma_f32): hipcc -O3 otherwise forms them from plain float code, and
    # some of their encodings have produced silently wrong values on this hardware pool
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
]
CXX_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-pthread"]


class BuildError(RuntimeError):
    pass


def _sha(paths: Sequence[Path], extra: str = "") -> str:
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(str(p.relative_to(REPO_ROOT)).encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def _run(cmd: List[str]) -> None:
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise BuildError("command failed: %s\n%s" % (" ".join(cmd), proc.stdout))


def _stamp_path(name: str) -> Path:
    return LIB_DIR / f".{name}.stamp.json"


def _up_to_date(name: str, digest: str, out: Path) -> bool:
    sp = _stamp_path(name)
    if not out.exists() or not sp.exists():
        return False
    try:
        return json.loads(sp.read_text()).get("digest") == digest
    except Exception:
        return False


def _compile_objects(srcs: Sequence[Path], compiler: str, flags: List[str], jobs: int) -> List[Path]:
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    objs: List[Path] = []
    cmds = []
    for s in srcs:
        o = OBJ_DIR / (s.parent.name + "_" + s.stem + ".o")
        objs.append(o)
        cmds.append([compiler, *flags, "-c", str(s), "-o", str(o)])
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        list(ex.map(_run, cmds))
    return objs


def build_kernels(force: bool = False, jobs: int = 8, verbose: bool = False) -> Path:
    """Compile all HIP kernels into one shared object for gfx950."""
    srcs = sorted((CSRC / "kernels").glob("*.hip"))
    hdrs = sorted((CSRC / "kernels").glob("*.h"))
    out = LIB_DIR / "libdlamd_kernels.so"
    digest = _sha(list(srcs) + list(hdrs), " ".join(HIP_FLAGS))
    if not force and _up_to_date("kernels", digest, out):
        return out
    if not srcs:
        raise BuildError("no HIP sources found under csrc/kernels")
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    flags = HIP_FLAGS + ["-I", str(CSRC / "kernels")]
    objs = _compile_objects(srcs, HIPCC, flags, jobs)
    _run([HIPCC, f"--offload-arch={GPU_ARCH}", "-shared", "-fPIC", "-o", str(out), *map(str, objs)])
    _stamp_path("kernels").write_text(json.dumps({"digest": digest}))
    if verbose:
        print(f"[build] {out} ({len(srcs)} HIP sources)")
    return out


def _build_cpp(name: str, subdir: str, libs: List[str], force: bool, jobs: int, verbose: bool) -> Path:
    srcs = sorted((CSRC / subdir).glob("*.cpp"))
    # the PDF library includes the JPEG decode core it shares with the kernels (csrc/kernels/jpeg_core.h)
    hdrs = sorted((CSRC / subdir).glob("*.h")) + sorted((CSRC / "kernels").glob("*core.h"))
    out = LIB_DIR / f"libdlamd_{name}.so"
    digest = _sha(list(srcs) + list(hdrs), " ".join(CXX_FLAGS + libs))
    if not force and _up_to_date(name, digest, out):
        return out
    if not srcs:
        raise BuildError(f"no C++ sources found under csrc/{subdir}")
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    flags = CXX_FLAGS + ["-I", str(CSRC / subdir)]
    objs = _compile_objects(srcs, CXX, flags, jobs)
    _run([CXX, "-shared", "-fPIC", "-pthread", "-o", str(out), *map(str, objs), *libs])
    _stamp_path(name).write_text(json.dumps({"digest": digest}))
    if verbose:
        print(f"[build] {out} ({len(srcs)} C++ sources)")
    return out


def build_pdf(force: bool = False, jobs: int = 8, verbose: bool = False) -> Path:
    return _build_cpp("pdf", "pdf", ["-lz"], force, jobs, verbose)


def build_runtime(force: bool = False, jobs: int = 8, verbose: bool = False) -> Path:
    return _build_cpp("runtime", "runtime", ["-lrt"], force, jobs, verbose)


def build_all(force: bool = False, jobs: int = 8, verbose: bool = False) -> Dict[str, Path]:
    out = {}
    out["kernels"] = build_kernels(force, jobs, verbose)
    if (CSRC / "pdf").exists() and any((CSRC / "pdf").glob("*.cpp")):
        out["pdf"] = build_pdf(force, jobs, verbose)
    if (CSRC / "runtime").exists() and any((CSRC / "runtime").glob("*.cpp")):
        out["runtime"] = build_runtime(force, jobs, verbose)
    return out


def lib_path(name: str) -> Path:
    return LIB_DIR / f"libdlamd_{name}.so"


if __name__ == "__main__":
    force = "--force" in sys.argv
    res = build_all(force=force, verbose=True)
    for k, v in res.items():
        print(k, v)
