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


def pytest_sessionstart(session):
    """The built library is git-ignored: in a fresh checkout that has a compiler, build it (and the C oracle) once, the way
    __graft_entry__.build() does, instead of failing the first test that loads it.  A box without hipcc still fails loudly there."""
    import shutil
    from threshold_crypto_amd import _native
    if os.path.exists(_native.LIB_PATH) or not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        return
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session")
def engine():
    """A libtc_amd.so context on GPU 0.  GPU tests must run the native HIP path: a missing
    library or device is a hard failure, never a skip or a fallback."""
    from threshold_crypto_amd.engine import Engine
    return Engine(0)


import contextlib


@contextlib.contextmanager
def engine_with_env(**env):
    """A fresh context created while `env` is in the environment: TC_DUO_MIN / TC_PAIRING_FORM / TC_PAIRING_BUDGET are read ONCE, by
    tc_ctx_create (csrc/tc_launch.h Tuning), so a test that forces a form builds its own context instead of toggling the variable
    under a live one.  The environment is restored before the context is used."""
    from threshold_crypto_amd.engine import Engine
    saved = {k: os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        eng = Engine(0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        yield eng
    finally:
        eng.close()
