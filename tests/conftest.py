import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _gpu_available():
        pytest.skip("no GPU")
    return True


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The oracle's C restatement and libswfr.so must exist before any test loads them."""
    from oracle import oracle_backend
    oracle_backend.build()
    import swf_renderer_amd.build as b
    if b.needs_build():
        b.build()
    yield
