"""Host C++ (PDF reader, JPEG scanner, inflater, batcher, BPE vocabulary, layout analysis) under ASan + UBSan on mutated
and nonsensical inputs."""
import shutil

import pytest

from distributed_llms_amd.utils import sanitize


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs a host compiler")
def test_native_host_code_is_clean_under_asan_ubsan():
    assert sanitize.run(iters=120, verbose=False) == 0
