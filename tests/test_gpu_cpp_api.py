"""Builds tests/cpp/test_api.cpp against include/threshold_crypto.hpp + libtc_amd.so and runs it on
the GPU: the reference's threshold-signature / threshold-encryption / bytes tests through the C++
host mirror.  Key material and expected values are produced here with the oracle."""
import os
import random
import struct
import subprocess

import pytest

import tc_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_api.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_api")
LIBDIR = os.path.join(ROOT, "threshold_crypto_amd")


def _build():
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), SRC, "-o", EXE, "-L" + LIBDIR,
                    "-ltc_amd", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"], check=True)


def test_cpp_header_compiles_and_links():
    """CPU-side: the C++ mirror compiles against the C ABI and links the shared library."""
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_api_replays_reference_tests(tmp_path):
    _build()
    rnd = random.Random(77)
    t, n = 3, 12
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    msg = b"Totally real news"
    fx = struct.pack("<II", t, n)
    fx += b"".join(o.fr_to_bytes(o.secret_key_share(poly, i)) for i in range(n))
    commit = o.commitment(poly)
    fx += b"".join(o.g1_uncompressed(c) for c in commit)
    fx += struct.pack("<I", len(msg)) + msg + o.g2_uncompressed(o.sign(poly[0], msg))
    plain = b"Muffins in the canteen today! Don't tell Eve!"
    u, v, w = o.encrypt_with_r(commit[0], rnd.randrange(1, o.R), plain)
    fx += o.g1_uncompressed(u) + struct.pack("<I", len(v)) + v + o.g2_uncompressed(w)
    fx += struct.pack("<I", len(plain)) + plain
    for ix in (-1, -(2 ** 40), 7, -(2 ** 63), 9):       # shares at i64 indices (IntoFr for i64: negative = -(|x|) mod r); t + 2 of them
        fx += struct.pack("<q", ix) + o.fr_to_bytes(o.poly_evaluate(poly, (ix + 1) % o.R))
    path = tmp_path / "fixture.bin"
    path.write_bytes(fx)
    r = subprocess.run([EXE, str(path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CPP-API-OK" in r.stdout, r.stdout + r.stderr
