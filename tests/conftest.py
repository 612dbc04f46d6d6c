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
    """A model and its engine reference each other (module -> engine -> module), so a test's 60-GB training workspace is
    only freed by the cycle collector - which counts Python allocations, not device bytes: by the end of the -m gpu suite
    the live set had grown to 284 of 288 GB and the batch-32 fp64 truth ran out of memory (round 6).  Collect after every
    GPU test and hand the cache back."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
