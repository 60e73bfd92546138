"""Debug build target: host C++ under AddressSanitizer + UBSan.

Builds ``csrc/tests/sanitize_main.cpp`` together with the PDF reader, batcher, BPE and layout-analysis sources into
``build/sanitize/dlamd_san`` (``-fsanitize=address,undefined``, runtimes linked statically so a preloaded
library cannot displace them) and runs it on freshly generated text / scanned PDFs plus thousands of
mutated copies, and the layout analysis on generated pages (columns, tables with wrapped and missing cells) and on
boxes that make no sense.  GPU code is not covered: device-side sanitizers are unavailable on this pool, the HIP
kernels are guarded by host-side shape contracts and canary tests instead (tests/test_kernels_gpu.py).

    python -m distributed_llms_amd.utils.sanitize [--iters N]
"""
from __future__ import annotations

import argparse
import subprocess
import sys
import tempfile
from pathlib import Path

from .build import CSRC, CXX, REPO_ROOT, BuildError

SAN_FLAGS = ["-g", "-O1", "-std=c++17", "-pthread", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
             "-fno-sanitize-recover=undefined", "-static-libasan", "-static-libubsan"]


def build_sanitized(verbose: bool = False) -> Path:
    out_dir = REPO_ROOT / "build" / "sanitize"
    out_dir.mkdir(parents=True, exist_ok=True)
    exe = out_dir / "dlamd_san"
    srcs = [CSRC / "tests" / "sanitize_main.cpp", *sorted((CSRC / "pdf").glob("*.cpp")), *sorted((CSRC / "runtime").glob("*.cpp"))]
    cmd = [CXX, *SAN_FLAGS, "-I", str(CSRC / "pdf"), "-I", str(CSRC / "runtime"), *map(str, srcs), "-o", str(exe), "-lz", "-lrt"]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise BuildError("sanitized build failed:\n" + proc.stdout)
    if verbose:
        print(f"[sanitize] built {exe}")
    return exe


def seed_documents(tmp: Path):
    from ..pdf import synth

    paths = []
    texts = [["Sanitizer seed page %d" % i, "The quick brown fox (jumps) over \\ the lazy dog", "Total: 1,250.00"] for i in range(3)]
    for name, kw in (("text_plain", dict(compress=False, xref_stream=False, use_tj_arrays=False)),
                     ("text_xrefstream", dict(compress=True, xref_stream=True, use_tj_arrays=True))):
        p = tmp / f"{name}.pdf"
        p.write_bytes(synth.make_text_pdf(texts, **kw))
        paths.append(p)
    pages = [synth.render_page(s, size=(310, 438), font_px=12, margin=16)[0] for s in (1, 2)]
    for codec in ("jpeg", "flate", "ccitt"):
        p = tmp / f"scan_{codec}.pdf"
        p.write_bytes(synth.make_scanned_pdf(pages, codec=codec))
        paths.append(p)
    return paths


def run(iters: int = 1500, verbose: bool = True) -> int:
    exe = build_sanitized(verbose)
    with tempfile.TemporaryDirectory() as td:
        seeds = seed_documents(Path(td))
        proc = subprocess.run([str(exe), f"--iters={iters}", *map(str, seeds)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True,
                              env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1",
                                   "PATH": "/usr/bin:/bin"})
    if verbose or proc.returncode != 0:
        print(proc.stdout)
    return proc.returncode


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--iters", type=int, default=1500)
    sys.exit(run(ap.parse_args().iters))
