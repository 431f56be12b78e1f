import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE = os.path.join(ROOT, "oracle")
if ORACLE not in sys.path:
    sys.path.insert(0, ORACLE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """A libtc_amd.so context on GPU 0.  GPU tests must run the native HIP path: a missing
    library or device is a hard failure, never a skip or a fallback."""
    from threshold_crypto_amd.engine import Engine
    return Engine(0)
