import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_err(a, b):
    """max|a-b| / max|b|  (the parity metric of SURVEY.md 8(d))"""
    import torch
    a = torch.as_tensor(a).detach().double()
    b = torch.as_tensor(b).detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(autouse=True)
def _release_device_memory(request):
    """Collect garbage and hand the allocator's cache back after every GPU test (a model and its engine reference each other,
    so a test's workspaces go with the cycle collector, which counts Python allocations, not device bytes).  Round 6 found
    the real leak behind the suite's growing live set - the training Functions hung autograd's node on tensors the
    engine keeps, a cycle through the C++ graph no collector sees (fixed in train.py: they return aliases) - with
    AMMC_TEST_MEMLOG=<file>: live device bytes after every test."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
            if os.environ.get("AMMC_TEST_MEMLOG"):            # (diagnostics: live device bytes after every GPU test)
                with open(os.environ["AMMC_TEST_MEMLOG"], "a") as fp:
                    fp.write(f"{torch.cuda.memory_allocated() / 2**30:8.2f} GiB live  {torch.cuda.memory_reserved() / 2**30:8.2f} reserved  {request.node.nodeid}\n")
