"""Engine: one context of libtc_amd.so bound to one MI355X, with array-level batch calls.

Arrays are either numpy (host memory: the library stages through its own device buffers)
or torch CUDA tensors (device-resident: nothing crosses PCIe; torch is used only as the
owner of device memory).  All compute happens in the HIP kernels behind the C ABI.
"""
import ctypes

import numpy as np

from . import _native

G1_BYTES = 96
G2_BYTES = 192
FR_BYTES = 32


class TcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libtc_amd call failed (%d): %s" % (code, msg))
        self.code = code


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    if x is None:
        return None
    if _is_torch(x):
        return ctypes.c_void_p(x.data_ptr())
    return ctypes.c_void_p(x.ctypes.data)


def pack_messages(msgs):
    """list of bytes -> (flat uint8 array, uint64 offsets[B+1])"""
    off = np.zeros(len(msgs) + 1, dtype=np.uint64)
    if len(msgs):
        off[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64)
    flat = np.frombuffer(b"".join(msgs), dtype=np.uint8).copy() if len(msgs) else np.zeros(0, np.uint8)
    if flat.size == 0:
        flat = np.zeros(1, np.uint8)
    return flat, off


class Engine:
    """Owns a tc_ctx.  Raises if the native library or a HIP device is missing."""

    def __init__(self, device=0):
        self._lib = _native.load()
        ctx = ctypes.c_void_p()
        rc = self._lib.tc_ctx_create(ctypes.byref(ctx), int(device))
        if rc != _native.TC_OK:
            raise TcError(rc, "tc_ctx_create failed on device %d (no gfx950 HIP device? there is no CPU fallback)"
                          % device)
        self._ctx = ctx
        self.device = int(device)
        self._device_io = False

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.tc_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing ---------------------------------------------------------------------------
    def set_timing(self, on):
        self._lib.tc_ctx_set_timing(self._ctx, 1 if on else 0)

    def last_kernel_ms(self):
        return float(self._lib.tc_last_kernel_ms(self._ctx))

    def set_stream(self, stream_ptr):
        self._lib.tc_ctx_set_stream(self._ctx, ctypes.c_void_p(stream_ptr) if stream_ptr else None)

    def sync(self):
        self._lib.tc_sync(self._ctx)

    def version(self):
        return self._lib.tc_version().decode()

    def _mode(self, *arrays):
        dev = [a for a in arrays if a is not None and _is_torch(a)]
        if dev and len(dev) != len([a for a in arrays if a is not None]):
            raise ValueError("mixing host (numpy) and device (torch) arrays in one call")
        want = bool(dev)
        if want != self._device_io:
            self._lib.tc_ctx_set_device_io(self._ctx, 1 if want else 0)
            self._device_io = want
        return want

    def _empty(self, like_device, shape, dtype=np.uint8, ref=None):
        if like_device:
            import torch
            tdt = {np.uint8: torch.uint8, np.uint64: torch.int64}[dtype]
            return torch.empty(shape, dtype=tdt, device=ref.device)
        return np.empty(shape, dtype=dtype)

    def _call(self, name, *args):
        rc = getattr(self._lib, name)(self._ctx, *args)
        if rc != _native.TC_OK:
            raise TcError(rc, self._lib.tc_last_error(self._ctx).decode())

    @staticmethod
    def _check(a, shape_tail, name):
        if tuple(a.shape[-len(shape_tail):]) != tuple(shape_tail):
            raise ValueError("%s: expected trailing shape %s, got %s" % (name, shape_tail, tuple(a.shape)))
        if not (a.is_contiguous() if _is_torch(a) else a.flags["C_CONTIGUOUS"]):
            raise ValueError("%s must be contiguous" % name)

    # -- hashing ------------------------------------------------------------------------------
    def hash_g2(self, msgs, off):
        dev = self._mode(msgs, off)
        B = off.shape[0] - 1
        out = self._empty(dev, (B, G2_BYTES), ref=msgs)
        self._call("tc_hash_g2_batch", _ptr(msgs), _ptr(off), B, _ptr(out))
        return out

    def hash_g1_g2(self, g1, msgs, off):
        dev = self._mode(g1, msgs, off)
        B = off.shape[0] - 1
        self._check(g1, (G1_BYTES,), "g1")
        out = self._empty(dev, (B, G2_BYTES), ref=g1)
        st = self._empty(dev, (B,), ref=g1)
        self._call("tc_hash_g1_g2_batch", _ptr(g1), _ptr(msgs), _ptr(off), B, _ptr(out), _ptr(st))
        return out, st

    # -- scalar multiplication -------------------------------------------------------------------
    def _mul(self, name, pb, fr, pts):
        dev = self._mode(fr, pts)
        self._check(fr, (FR_BYTES,), "fr")
        self._check(pts, (pb,), "pts")
        S, B = fr.shape[0], pts.shape[0]
        out = self._empty(dev, (B, S, pb), ref=pts)
        st = self._empty(dev, (B, S), ref=pts)
        self._call(name, _ptr(fr), _ptr(pts), S, B, _ptr(out), _ptr(st))
        return out, st

    def g2_mul(self, fr, pts):
        """out[j, s] = fr[s] * pts[j]"""
        return self._mul("tc_g2_mul_batch", G2_BYTES, fr, pts)

    def g1_mul(self, fr, pts):
        return self._mul("tc_g1_mul_batch", G1_BYTES, fr, pts)

    def sign(self, fr, msgs, off):
        dev = self._mode(fr, msgs, off)
        S, B = fr.shape[0], off.shape[0] - 1
        out = self._empty(dev, (B, S, G2_BYTES), ref=fr)
        st = self._empty(dev, (B, S), ref=fr)
        self._call("tc_sign_batch", _ptr(fr), _ptr(msgs), _ptr(off), S, B, _ptr(out), _ptr(st))
        return out, st

    # -- combination --------------------------------------------------------------------------------
    def _combine(self, name, pb, t, idx, shares):
        dev = self._mode(idx, shares)
        B, n = idx.shape
        self._check(shares, (n, pb), "shares")
        out = self._empty(dev, (B, pb), ref=shares)
        st = self._empty(dev, (B,), ref=shares)
        self._call(name, int(t), int(n), _ptr(idx), _ptr(shares), B, _ptr(out), _ptr(st))
        return out, st

    def combine_g2(self, t, idx, shares):
        return self._combine("tc_combine_g2_batch", G2_BYTES, t, idx, shares)

    def combine_g1(self, t, idx, shares):
        return self._combine("tc_combine_g1_batch", G1_BYTES, t, idx, shares)

    def _lincomb(self, name, pb, scalars, points):
        dev = self._mode(scalars, points)
        B, n = scalars.shape[0], scalars.shape[1]
        self._check(scalars, (n, FR_BYTES), "scalars")
        self._check(points, (n, pb), "points")
        out = self._empty(dev, (B, pb), ref=points)
        st = self._empty(dev, (B,), ref=points)
        self._call(name, int(n), _ptr(scalars), _ptr(points), B, _ptr(out), _ptr(st))
        return out, st

    def lincomb_g1(self, scalars, points):
        """out[j] = sum_k scalars[j, k] * points[j, k]"""
        return self._lincomb("tc_g1_lincomb_batch", G1_BYTES, scalars, points)

    def lincomb_g2(self, scalars, points):
        return self._lincomb("tc_g2_lincomb_batch", G2_BYTES, scalars, points)

    def decrypt(self, t, idx, shares_g1, v, off):
        dev = self._mode(idx, shares_g1, v, off)
        B, n = idx.shape
        self._check(shares_g1, (n, G1_BYTES), "shares")
        out = self._empty(dev, tuple(v.shape), ref=v)
        st = self._empty(dev, (B,), ref=v)
        self._call("tc_decrypt_batch", int(t), int(n), _ptr(idx), _ptr(shares_g1), _ptr(v), _ptr(off), B,
                   _ptr(out), _ptr(st))
        return out, st

    def xor_with_hash(self, g1, data, off):
        dev = self._mode(g1, data, off)
        B = off.shape[0] - 1
        out = self._empty(dev, tuple(data.shape), ref=data)
        st = self._empty(dev, (B,), ref=data)
        self._call("tc_xor_with_hash_batch", _ptr(g1), _ptr(data), _ptr(off), B, _ptr(out), _ptr(st))
        return out, st

    # -- pairing checks -------------------------------------------------------------------------------
    @staticmethod
    def _stride(a, nbytes):
        return 0 if a.ndim == 1 else nbytes

    def pairing_check(self, a, b, c, d, B=None):
        """ok[j] = e(a[j], b[j]) == e(c[j], d[j]); 1-D operands are broadcast to every job."""
        dev = self._mode(a, b, c, d)
        if B is None:
            B = max(x.shape[0] if x.ndim == 2 else 1 for x in (a, b, c, d))
        ok = self._empty(dev, (B,), ref=a)
        self._call("tc_pairing_check_batch", _ptr(a), self._stride(a, G1_BYTES), _ptr(b), self._stride(b, G2_BYTES),
                   _ptr(c), self._stride(c, G1_BYTES), _ptr(d), self._stride(d, G2_BYTES), B, _ptr(ok))
        return ok

    def verify_g2(self, pk, sig, hash_):
        dev = self._mode(pk, sig, hash_)
        B = sig.shape[0]
        ok = self._empty(dev, (B,), ref=sig)
        self._call("tc_verify_g2_batch", _ptr(pk), self._stride(pk, G1_BYTES), _ptr(sig), _ptr(hash_), B, _ptr(ok))
        return ok

    def verify_sig(self, pk, sig, msgs, off):
        dev = self._mode(pk, sig, msgs, off)
        B = sig.shape[0]
        ok = self._empty(dev, (B,), ref=sig)
        self._call("tc_verify_sig_batch", _ptr(pk), self._stride(pk, G1_BYTES), _ptr(sig), _ptr(msgs), _ptr(off), B,
                   _ptr(ok))
        return ok

    def ciphertext_verify(self, u, v, off, w):
        dev = self._mode(u, v, off, w)
        B = u.shape[0]
        ok = self._empty(dev, (B,), ref=u)
        self._call("tc_ciphertext_verify_batch", _ptr(u), _ptr(v), _ptr(off), _ptr(w), B, _ptr(ok))
        return ok

    def verify_decryption_share(self, pk_share, share, u, v, off, w):
        dev = self._mode(pk_share, share, u, v, off, w)
        B = share.shape[0]
        ok = self._empty(dev, (B,), ref=share)
        self._call("tc_verify_decryption_share_batch", _ptr(pk_share), self._stride(pk_share, G1_BYTES), _ptr(share),
                   _ptr(u), _ptr(v), _ptr(off), _ptr(w), B, _ptr(ok))
        return ok

    # -- wire formats ------------------------------------------------------------------------------------
    def g1_compress(self, pts):
        dev = self._mode(pts)
        B = pts.shape[0]
        out = self._empty(dev, (B, 48), ref=pts)
        st = self._empty(dev, (B,), ref=pts)
        self._call("tc_g1_compress_batch", _ptr(pts), B, _ptr(out), _ptr(st))
        return out, st

    def g2_compress(self, pts):
        dev = self._mode(pts)
        B = pts.shape[0]
        out = self._empty(dev, (B, 96), ref=pts)
        st = self._empty(dev, (B,), ref=pts)
        self._call("tc_g2_compress_batch", _ptr(pts), B, _ptr(out), _ptr(st))
        return out, st

    def g1_decompress(self, comp):
        """checked decode of 48-byte compressed G1 (from_bytes): status 3 = invalid"""
        dev = self._mode(comp)
        B = comp.shape[0]
        out = self._empty(dev, (B, G1_BYTES), ref=comp)
        st = self._empty(dev, (B,), ref=comp)
        self._call("tc_g1_decompress_batch", _ptr(comp), B, _ptr(out), _ptr(st))
        return out, st

    def g2_decompress(self, comp):
        dev = self._mode(comp)
        B = comp.shape[0]
        out = self._empty(dev, (B, G2_BYTES), ref=comp)
        st = self._empty(dev, (B,), ref=comp)
        self._call("tc_g2_decompress_batch", _ptr(comp), B, _ptr(out), _ptr(st))
        return out, st

    def encrypt(self, pk, r, msgs, off):
        """PublicKey::encrypt_with_rng with the Fr draws supplied: returns (u, v, w, status)."""
        dev = self._mode(pk, r, msgs, off)
        B = off.shape[0] - 1
        self._check(r, (FR_BYTES,), "r")
        u = self._empty(dev, (B, G1_BYTES), ref=r)
        v = self._empty(dev, tuple(msgs.shape), ref=r)
        w = self._empty(dev, (B, G2_BYTES), ref=r)
        st = self._empty(dev, (B,), ref=r)
        self._call("tc_encrypt_batch", _ptr(pk), self._stride(pk, G1_BYTES), _ptr(r), _ptr(msgs), _ptr(off), B, _ptr(u),
                   _ptr(v), _ptr(w), _ptr(st))
        return u, v, w, st

    def public_key_shares(self, commit, idx):
        """Commitment::evaluate(idx + 1) for every index: (M, 96)."""
        dev = self._mode(commit, idx)
        M = idx.shape[0]
        t = commit.shape[0] - 1
        out = self._empty(dev, (M, G1_BYTES), ref=commit)
        st = self._empty(dev, (M,), ref=commit)
        self._call("tc_public_key_share_batch", _ptr(commit), int(t), _ptr(idx), M, _ptr(out), _ptr(st))
        return out, st
