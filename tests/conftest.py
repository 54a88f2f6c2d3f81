import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running statistical test")


def _have_gpu():
    try:
        from commpy_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run on the HIP path: fail (not skip) when selected with -m gpu but no device/library."""
    from commpy_amd import _lib
    _lib.load()
    if _lib.device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests need the MI355X and libcommpy_amd.so")
    return True
