"""The C-ABI shared library loads in a GPU-less container, exports every symbol the public
header declares, and refuses to run without a HIP device (no CPU fallback).  No compute calls."""
import ctypes
import os
import re

import pytest

from threshold_crypto_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "tc_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    assert os.path.exists(_native.LIB_PATH), "build with python -m threshold_crypto_amd.build"
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export " + n


def test_python_binding_covers_header():
    assert sorted(_native.ALL_SYMBOLS) == _declared()


def test_header_cites_reference_lines():
    text = open(os.path.join(ROOT, "include", "tc_amd.h")).read()
    assert text.count("src/lib.rs:") >= 15


def test_no_device_is_a_hard_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _native.load()
    ctx = ctypes.c_void_p()
    rc = lib.tc_ctx_create(ctypes.byref(ctx), 0)
    assert rc == _native.TC_ERR_NO_DEVICE and not ctx.value
    from threshold_crypto_amd.engine import Engine, TcError
    with pytest.raises(TcError):
        Engine(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "threshold_crypto_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "tc_oracle" not in src and "c_oracle" not in src and "hostsim" not in src.replace(
                    "tests/hostsim", ""), f
    # developer tools stay oracle-free too (the fixture generator is the one sanctioned exception);
    # everything that checks against the oracle lives under tests/
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith(".py") and f != "gen_golden.py":
            src = open(os.path.join(ROOT, "tools", f), errors="ignore").read()
            assert "tc_oracle" not in src and "c_oracle" not in src and "\"oracle\"" not in src, f
