"""HostSimEngine: the Engine methods the multi-rank flows use (bench.py, threshold_crypto_amd/config5.py), executed by
tests/hostsim -- the SAME per-lane job bodies the kernels run (threshold_crypto_amd/csrc/*.h), compiled by g++.

TEST HARNESS ONLY.  It exists so that the N-rank paths (rank spawn, rendezvous, key-set broadcast, sharding from global
job indices, all-reduce / gather, the bench line's bookkeeping) can run on `gloo` ranks in the GPU-less build
container; nothing in the product imports it, and a bench line produced with it says so ("test_harness").
Arrays may be numpy or torch CPU tensors; results come back in the kind the inputs had."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "threshold_crypto_amd", "csrc")
SRC = os.path.join(HERE, "hostsim", "hostsim.cpp")
LIB = os.path.join(HERE, "hostsim", "libtc_hostsim.so")
_G1_GEN = None


def build():
    newest = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(newest, os.path.getmtime(SRC)):
        tmp = LIB + ".%d.tmp" % os.getpid()   # several ranks may get here together: build aside, rename atomically
        subprocess.run(["g++", "-O2", "-std=c++17", "-DTC_TEST_HOOKS", "-shared", "-fPIC", "-I" + CSRC, SRC, "-o", tmp], check=True)
        os.replace(tmp, LIB)
    return LIB


def _np(a):
    return a.numpy() if type(a).__module__.startswith("torch") else np.asarray(a)


class HostSimEngine:
    is_test_harness = True

    def __init__(self, device=None):
        self.L = ctypes.CDLL(build())
        self.ct = ctypes
        self.L.hs_hash_g2.restype = None
        self.device = device

    # -- plumbing the bench expects of an Engine --------------------------------------------------------------
    def last_kernel_ms(self):
        return 0.0

    def set_timing(self, on):
        pass

    def sync(self):
        pass

    def close(self):
        pass

    def version(self):
        return "tests/hostsim (g++ build of the device source; TEST HARNESS)"

    def _buf(self, n):
        return self.ct.create_string_buffer(n)

    @staticmethod
    def _like(ref, arr):
        if type(ref).__module__.startswith("torch"):
            import torch
            return torch.from_numpy(arr)
        return arr

    # -- the job bodies ---------------------------------------------------------------------------------------
    def g1_commitment(self, fr):
        f = _np(fr)
        out = np.zeros((f.shape[0], 96), np.uint8)
        st = np.zeros(f.shape[0], np.uint8)
        for i in range(f.shape[0]):
            b = self._buf(96)
            st[i] = self.L.hs_g1_fixed_base_mul(bytes(f[i]), b)
            out[i] = np.frombuffer(b.raw, np.uint8)
        return self._like(fr, out), self._like(fr, st)

    def hash_g2(self, flat, off):
        fl, of = _np(flat), _np(off)
        B = of.shape[0] - 1
        out = np.zeros((B, 192), np.uint8)
        for j in range(B):
            m = bytes(fl[int(of[j]): int(of[j + 1])])
            b = self._buf(192)
            self.L.hs_hash_g2(m, self.ct.c_size_t(len(m)), b)
            out[j] = np.frombuffer(b.raw, np.uint8)
        return self._like(flat, out)

    def _mul(self, fn, pb, fr, pts):
        f, p = _np(fr), _np(pts)
        S, B = f.shape[0], p.shape[0]
        out = np.zeros((B, S, pb), np.uint8)
        st = np.zeros((B, S), np.uint8)
        for j in range(B):
            for s in range(S):
                b = self._buf(pb)
                st[j, s] = fn(bytes(f[s]), bytes(p[j]), b)
                out[j, s] = np.frombuffer(b.raw, np.uint8)
        return self._like(pts, out), self._like(pts, st)

    def g2_mul(self, fr, pts):
        return self._mul(self.L.hs_g2_mul, 192, fr, pts)

    def g1_mul(self, fr, pts):
        return self._mul(self.L.hs_g1_mul, 96, fr, pts)

    def sign_shares_g2(self, sk_table, idx, hashes):
        sk, ix, hs = _np(sk_table), _np(idx), _np(hashes)
        B, n = ix.shape
        out = np.zeros((B, n, 192), np.uint8)
        st = np.zeros((B, n), np.uint8)
        for j in range(B):
            for k in range(n):
                b = self._buf(192)
                st[j, k] = self.L.hs_g2_mul(bytes(sk[int(ix[j, k])]), bytes(hs[j]), b)
                out[j, k] = np.frombuffer(b.raw, np.uint8)
        return self._like(hashes, out), self._like(hashes, st)

    def combine_g2(self, t, idx, shares):
        ix, sh = _np(idx), _np(shares)
        B, n = ix.shape
        out = np.zeros((B, 192), np.uint8)
        st = np.zeros(B, np.uint8)
        for j in range(B):
            b = self._buf(192)
            ids = (self.ct.c_uint64 * n)(*[int(v) & 0xFFFFFFFFFFFFFFFF for v in ix[j]])
            st[j] = self.L.hs_combine_g2(int(t), ids, sh[j].tobytes(), b)
            out[j] = np.frombuffer(b.raw, np.uint8)
        return self._like(shares, out), self._like(shares, st)

    def verify_g2(self, pk, sig, hashes):
        global _G1_GEN
        if _G1_GEN is None:
            from threshold_crypto_amd import api
            _G1_GEN = api._G1_GEN
        p, s, h = _np(pk), _np(sig), _np(hashes)
        ok = np.array([self.L.hs_pairing_check(bytes(p if p.ndim == 1 else p[j]), bytes(h[j]), _G1_GEN, bytes(s[j]))
                       for j in range(s.shape[0])], np.uint8)
        return self._like(sig, ok)
