import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(1)       # the oracle's tiny-N loops are fastest single-threaded
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ensure_built():
    """Build libgpimhip.so if it is missing or stale (hipcc cross-compiles without a GPU)."""
    from gpim_amd import _build
    return _build.build()
