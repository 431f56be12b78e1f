"""ctypes loader for Oracle B (oracle/libtc_oracle.so, built from oracle/c/tc_oracle.c).
TEST INFRASTRUCTURE ONLY -- see the header of c/tc_oracle.c."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtc_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "c", "tc_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True)
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        path = os.environ.get("TC_ORACLE_LIB") or LIB_PATH   # tests: a sanitizer build of the same source (tests/test_hostsim.py)
        if path == LIB_PATH and not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(path)
        _lib.or_fq_mul_count.restype = ctypes.c_uint64
        for name in ("or_hash_g2", "or_sha3_256", "or_fq_mul_count_reset"):
            getattr(_lib, name).restype = None
        sz = ctypes.c_size_t
        cp = ctypes.c_char_p
        vp = ctypes.c_void_p
        _lib.or_set_hspec.argtypes = [ctypes.c_int]
        _lib.or_set_hspec.restype = None
        _lib.or_hash_g2.argtypes = [cp, sz, cp]
        _lib.or_hash_g1_g2.argtypes = [cp, cp, sz, cp]
        _lib.or_xor_with_hash.argtypes = [cp, cp, sz, cp]
        _lib.or_g2_mul.argtypes = [cp, cp, cp]
        _lib.or_g1_mul.argtypes = [cp, cp, cp]
        _lib.or_sign.argtypes = [cp, cp, sz, cp]
        _lib.or_lagrange.argtypes = [sz, vp, cp]
        _lib.or_combine_g2.argtypes = [sz, sz, vp, cp, cp]
        _lib.or_combine_g1.argtypes = [sz, sz, vp, cp, cp]
        _lib.or_threshold_decrypt.argtypes = [sz, sz, vp, cp, cp, sz, cp]
        _lib.or_pairing_check.argtypes = [cp, cp, cp, cp]
        _lib.or_pairing_gt.argtypes = [cp, cp, cp]
        _lib.or_verify_g2.argtypes = [cp, cp, cp]
        _lib.or_verify.argtypes = [cp, cp, cp, sz]
        _lib.or_ciphertext_verify.argtypes = [cp, cp, sz, cp]
        _lib.or_verify_decryption_share.argtypes = [cp, cp, cp, cp, sz, cp]
        _lib.or_g1_compress.argtypes = [cp, cp]
        _lib.or_g2_compress.argtypes = [cp, cp]
        _lib.or_sha3_256.argtypes = [cp, sz, cp]
        _lib.or_g1_decompress.argtypes = [cp, cp]
        _lib.or_g2_decompress.argtypes = [cp, cp]
        _lib.or_combine_g2_batch.argtypes = [sz, sz, vp, vp, sz, vp, vp, ctypes.c_int]
        _lib.or_combine_g2_batch.restype = None
        _lib.or_verify_g2_batch.argtypes = [cp, vp, vp, sz, vp, ctypes.c_int]
        _lib.or_verify_g2_batch.restype = None
        _lib.or_ciphertext_verify_batch.argtypes = [vp, vp, sz, vp, sz, vp, ctypes.c_int]
        _lib.or_ciphertext_verify_batch.restype = None
        _lib.or_threshold_decrypt_batch.argtypes = [sz, sz, vp, vp, vp, sz, sz, vp, vp, ctypes.c_int]
        _lib.or_threshold_decrypt_batch.restype = None
        _lib.or_combine_signatures_wire.argtypes = [sz, sz, vp, cp, cp]
        _lib.or_decrypt_wire.argtypes = [sz, sz, vp, cp, cp, sz, cp]
        _lib.or_combine_signatures_wire_batch.argtypes = [sz, sz, vp, vp, sz, vp, vp, ctypes.c_int]
        _lib.or_combine_signatures_wire_batch.restype = None
        _lib.or_sign_combine_batch.argtypes = [sz, sz, sz, vp, vp, vp, sz, vp, vp, ctypes.c_int]
        _lib.or_sign_combine_batch.restype = None
        _lib.or_combine_g1_batch.argtypes = [sz, sz, vp, vp, sz, vp, vp, ctypes.c_int]
        _lib.or_combine_g1_batch.restype = None
        _lib.or_sign_shares_batch.argtypes = [sz, sz, vp, vp, vp, sz, vp, vp, ctypes.c_int]
        _lib.or_sign_shares_batch.restype = None
    return _lib


def set_hspec(v):
    """H-spec alternative bits (oracle/c/tc_oracle.c g_hspec; the same bits as tc_oracle.HSPEC); 0 = the recalled behaviour"""
    load().or_set_hspec(int(v))


def _buf(n):
    return ctypes.create_string_buffer(n)


def _idx(ids):
    return (ctypes.c_uint64 * len(ids))(*ids)


def hash_g2(msg):
    out = _buf(192)
    load().or_hash_g2(bytes(msg), len(msg), out)
    return out.raw


def hash_g1_g2(g1, msg):
    out = _buf(192)
    rc = load().or_hash_g1_g2(g1, bytes(msg), len(msg), out)
    return rc, out.raw


def xor_with_hash(g1, data):
    out = _buf(max(1, len(data)))
    rc = load().or_xor_with_hash(g1, bytes(data), len(data), out)
    return rc, out.raw[: len(data)]


def g2_mul(fr, pt):
    out = _buf(192)
    return load().or_g2_mul(fr, pt, out), out.raw


def g1_mul(fr, pt):
    out = _buf(96)
    return load().or_g1_mul(fr, pt, out), out.raw


def sign(fr, msg):
    out = _buf(192)
    return load().or_sign(fr, bytes(msg), len(msg), out), out.raw


def lagrange(t, ids):
    out = _buf(32 * (t + 1))
    rc = load().or_lagrange(t, _idx(ids), out)
    return rc, [int.from_bytes(out.raw[32 * i: 32 * i + 32], "little") for i in range(t + 1)]


def combine_g2(t, ids, shares):
    out = _buf(192)
    return load().or_combine_g2(t, len(ids), _idx(ids), b"".join(shares), out), out.raw


def combine_g1(t, ids, shares):
    out = _buf(96)
    return load().or_combine_g1(t, len(ids), _idx(ids), b"".join(shares), out), out.raw


def threshold_decrypt(t, ids, shares, v):
    out = _buf(max(1, len(v)))
    rc = load().or_threshold_decrypt(t, len(ids), _idx(ids), b"".join(shares), bytes(v), len(v), out)
    return rc, out.raw[: len(v)]


def pairing_check(a, b, c, d):
    return load().or_pairing_check(a, b, c, d)


def pairing_gt(a, b):
    out = _buf(576)
    rc = load().or_pairing_gt(a, b, out)
    return rc, out.raw


def verify_g2(pk, sig, h):
    return load().or_verify_g2(pk, sig, h)


def verify(pk, sig, msg):
    return load().or_verify(pk, sig, bytes(msg), len(msg))


def ciphertext_verify(u, v, w):
    return load().or_ciphertext_verify(u, bytes(v), len(v), w)


def verify_decryption_share(pk_share, share, u, v, w):
    return load().or_verify_decryption_share(pk_share, share, u, bytes(v), len(v), w)


def g1_compress(p):
    out = _buf(48)
    return load().or_g1_compress(p, out), out.raw


def g2_compress(p):
    out = _buf(96)
    return load().or_g2_compress(p, out), out.raw


def sha3_256(msg):
    out = _buf(32)
    load().or_sha3_256(bytes(msg), len(msg), out)
    return out.raw


def fq_mul_count(reset=False):
    lib = load()
    n = lib.or_fq_mul_count()
    if reset:
        lib.or_fq_mul_count_reset()
    return n


def combine_g2_batch(t, idx, shares, nthreads):
    """idx: (B, n) uint64 numpy, shares: (B, n, 192) uint8 numpy -> (out (B,192) uint8, rc (B,) int32)"""
    import numpy as np
    B, n = idx.shape
    out = np.zeros((B, 192), dtype=np.uint8)
    rc = np.zeros(B, dtype=np.int32)
    idx = np.ascontiguousarray(idx)
    shares = np.ascontiguousarray(shares)
    load().or_combine_g2_batch(t, n, idx.ctypes.data, shares.ctypes.data, B, out.ctypes.data, rc.ctypes.data, nthreads)
    return out, rc


def verify_g2_batch(pk, sigs, hashes, nthreads):
    import numpy as np
    B = sigs.shape[0]
    rc = np.zeros(B, dtype=np.int32)
    sigs = np.ascontiguousarray(sigs)
    hashes = np.ascontiguousarray(hashes)
    load().or_verify_g2_batch(bytes(pk), sigs.ctypes.data, hashes.ctypes.data, B, rc.ctypes.data, nthreads)
    return rc


def ciphertext_verify_batch(u, v, length, w, nthreads):
    """u (B,96), v (B*length,) uint8, w (B,192) -> rc (B,) int32 (1 = valid)"""
    import numpy as np
    B = u.shape[0]
    rc = np.zeros(B, dtype=np.int32)
    u, v, w = np.ascontiguousarray(u), np.ascontiguousarray(v), np.ascontiguousarray(w)
    load().or_ciphertext_verify_batch(u.ctypes.data, v.ctypes.data, length, w.ctypes.data, B, rc.ctypes.data, nthreads)
    return rc


def threshold_decrypt_batch(t, idx, shares, v, length, nthreads):
    """idx (B,n) uint64, shares (B,n,96), v (B*length,) -> (plain (B*length,) uint8, rc (B,) int32)"""
    import numpy as np
    B, n = idx.shape
    out = np.zeros(B * length, dtype=np.uint8)
    rc = np.zeros(B, dtype=np.int32)
    idx, shares, v = np.ascontiguousarray(idx), np.ascontiguousarray(shares), np.ascontiguousarray(v)
    load().or_threshold_decrypt_batch(t, n, idx.ctypes.data, shares.ctypes.data, v.ctypes.data, length, B, out.ctypes.data,
                                      rc.ctypes.data, nthreads)
    return out, rc


def combine_signatures_wire(t, ids, shares96):
    """(rc, 96 B Signature::to_bytes) from the first t+1 of the 96-byte compressed shares (checked decode)"""
    out = _buf(96)
    return load().or_combine_signatures_wire(t, len(ids), _idx(ids), b"".join(bytes(s) for s in shares96), out), out.raw


def decrypt_wire(t, ids, shares48, v):
    out = _buf(max(len(v), 1))
    rc = load().or_decrypt_wire(t, len(ids), _idx(ids), b"".join(bytes(s) for s in shares48), bytes(v), len(v), out)
    return rc, out.raw[:len(v)]


def combine_signatures_wire_batch(t, idx, shares96, nthreads):
    """idx: (B, n) uint64, shares96: (B, n, 96) uint8 -> (out (B, 96) uint8, rc (B,) int32)"""
    import numpy as np
    B, n = idx.shape
    out = np.zeros((B, 96), dtype=np.uint8)
    rc = np.zeros(B, dtype=np.int32)
    idx, shares96 = np.ascontiguousarray(idx), np.ascontiguousarray(shares96)
    load().or_combine_signatures_wire_batch(t, n, idx.ctypes.data, shares96.ctypes.data, B, out.ctypes.data, rc.ctypes.data, nthreads)
    return out, rc


def sign_combine_batch(t, sk_table, idx, hashes, nthreads):
    """B threshold signatures from scratch on `nthreads` host threads: job k signs hashes[k] with the key shares
    sk_table[idx[k][*]] (SecretKeyShare::sign_g2) and combines them (combine_signatures)."""
    import numpy as np
    sk_table = np.ascontiguousarray(sk_table, dtype=np.uint8)
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    hashes = np.ascontiguousarray(hashes, dtype=np.uint8)
    B, n = idx.shape
    out = np.zeros((B, 192), dtype=np.uint8)
    rc = np.zeros(B, dtype=np.int32)
    load().or_sign_combine_batch(t, n, sk_table.shape[0], sk_table.ctypes.data, idx.ctypes.data, hashes.ctypes.data, B, out.ctypes.data,
                                 rc.ctypes.data, nthreads)
    return out, rc


def combine_g1_batch(t, idx, shares, nthreads):
    import numpy as np
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    shares = np.ascontiguousarray(shares, dtype=np.uint8)
    B, n = idx.shape
    out = np.zeros((B, 96), dtype=np.uint8)
    rc = np.zeros(B, dtype=np.int32)
    load().or_combine_g1_batch(t, n, idx.ctypes.data, shares.ctypes.data, B, out.ctypes.data, rc.ctypes.data, nthreads)
    return out, rc


def sign_shares_batch(sk_table, idx, hashes, nthreads):
    """out[k][s] = [sk_table[idx[k][s]]] hashes[k] for B messages x n signers"""
    import numpy as np
    sk_table = np.ascontiguousarray(sk_table, dtype=np.uint8)
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    hashes = np.ascontiguousarray(hashes, dtype=np.uint8)
    B, n = idx.shape
    out = np.zeros((B, n, 192), dtype=np.uint8)
    rc = np.zeros(B, dtype=np.int32)
    load().or_sign_shares_batch(n, sk_table.shape[0], sk_table.ctypes.data, idx.ctypes.data, hashes.ctypes.data, B, out.ctypes.data,
                                rc.ctypes.data, nthreads)
    return out, rc


def host_threads():
    """threads the batch drivers can really use: the CPU affinity mask capped by the cgroup quota"""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def g1_decompress(b):
    out = _buf(96)
    return load().or_g1_decompress(bytes(b), out), out.raw


def g2_decompress(b):
    out = _buf(192)
    return load().or_g2_decompress(bytes(b), out), out.raw
