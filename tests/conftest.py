import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (-m "not gpu") runs over a few worker processes when pytest-xdist is there and the command line
    does not say otherwise (8 min in one process, under 3 with four).  GPU runs stay in ONE process: the box allows
    few processes on its card, and the multi-process GPU tests start their own ranks."""
    opt = config.option
    if os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput"):
        return                      # already one of the workers: they inherit the command line
    if (getattr(opt, "markexpr", "") == "not gpu" and hasattr(opt, "numprocesses") and opt.numprocesses is None
            and not os.environ.get("DLAMD_TESTS_SERIAL") and not getattr(opt, "collectonly", False)):
        opt.numprocesses = max(1, min(4, len(os.sched_getaffinity(0)) // 2 if hasattr(os, "sched_getaffinity") else 2))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running test")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
