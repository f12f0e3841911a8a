import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import candle_birefnet_amd as cb
        return cb.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _has_gpu():
        pytest.skip("no HIP device")
    return 0
