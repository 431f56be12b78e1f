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
        # device-I/O calls return before their kernels have run: the operand tensors of the calls since the last sync() are
        # kept alive here, so that `eng.verify_sig(pk.to(dev), ...)` with temporaries is not a use-after-free (torch would hand a
        # dead tensor's memory to the next allocation while the kernels still read it)
        self._keep = {}

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.tc_ctx_destroy(self._ctx)     # (waits for the context's streams)
            self._ctx = None
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing ---------------------------------------------------------------------------
    def set_timing(self, on):
        self._lib.tc_ctx_set_timing(self._ctx, 1 if on else 0)

    def set_input_checks(self, on):
        """checked-input mode (the context's default): every point operand is tested for order-r membership on the
        device first.  False is the explicit opt-out for operands known to be members (this library's own outputs)."""
        self._lib.tc_ctx_set_input_checks(self._ctx, 1 if on else 0)

    def input_checks(self):
        return bool(self._lib.tc_ctx_get_input_checks(self._ctx))

    def trim(self):
        """give the context's staging / table buffers back to the device (tc_ctx_trim)"""
        self._lib.tc_ctx_trim(self._ctx)

    def transfer_bytes(self):
        """(host-to-device, device-to-host) bytes this context's staging copies have moved over PCIe"""
        up, down = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._lib.tc_ctx_transfer_bytes(self._ctx, ctypes.byref(up), ctypes.byref(down))
        return int(up.value), int(down.value)

    def tuning(self):
        """the form choices of this context (tc_ctx_get_tuning): jobs from which a checked G2 decode / a hash takes two jobs per
        lane pair, the pairing form (0 = by batch size), the line-buffer budget (0 = a third of the free HBM), whether the membership
        tests of checked-input mode run beside the main kernels on a second stream.  Fixed when the context was created (defaults, or
        TC_DUO_MIN / TC_PAIRING_FORM / TC_PAIRING_BUDGET / TC_CHECKS_BESIDE / TC_MSM_BUDGET / TC_PRIVATE_RESERVE in the environment at that
        moment)."""
        out = (ctypes.c_uint64 * 8)()
        rc = self._lib.tc_ctx_get_tuning(self._ctx, out)
        if rc != 0:
            raise TcError(rc, "tc_ctx_get_tuning")
        return {"duo_min_decode": int(out[0]), "duo_min_hash": int(out[1]), "pairing_form": int(out[2]), "pairing_budget": int(out[3]),
                "checks_beside": int(out[4]), "msm_budget": int(out[5]), "private_reserve": int(out[6])}

    def last_kernel_ms(self):
        return float(self._lib.tc_last_kernel_ms(self._ctx))

    def set_stream(self, stream_ptr):
        self._lib.tc_ctx_set_stream(self._ctx, ctypes.c_void_p(stream_ptr) if stream_ptr else None)

    def sync(self):
        self._lib.tc_sync(self._ctx)
        self._keep.clear()

    def version(self):
        return self._lib.tc_version().decode()

    def _mode(self, *arrays):
        dev = [a for a in arrays if a is not None and _is_torch(a)]
        if dev and len(dev) != len([a for a in arrays if a is not None]):
            raise ValueError("mixing host (numpy) and device (torch) arrays in one call")
        want = bool(dev)
        for a in dev:
            self._keep[id(a)] = a
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
    def _arg(a, shape, kind, name):
        """Every operand is validated before its raw pointer crosses the C ABI: dtype (`kind` "u8" or
        "u64"), C-contiguity and the FULL expected shape (None = any extent).  A wrong dtype or a sliced
        view would otherwise be misread, or read out of bounds, by the library."""
        if a is None:
            raise ValueError("%s is required" % name)
        torch_arr = _is_torch(a)
        dt = str(a.dtype)
        good = {"u8": ("uint8", "torch.uint8"), "u64": ("uint64", "int64", "torch.int64", "torch.uint64")}[kind]
        if dt not in good:
            raise TypeError("%s: expected dtype %s, got %s" % (name, good[0], dt))
        if len(a.shape) != len(shape) or any(w is not None and int(g) != int(w) for g, w in zip(a.shape, shape)):
            raise ValueError("%s: expected shape %s, got %s" % (name, tuple(shape), tuple(a.shape)))
        if not (a.is_contiguous() if torch_arr else a.flags["C_CONTIGUOUS"]):
            raise ValueError("%s must be C-contiguous" % name)
        return a

    def _msgs(self, msgs, off, B=None):
        """a message blob + its offsets[B+1]; returns B"""
        self._arg(off, (None,), "u64", "off")
        if off.shape[0] < 1:
            raise ValueError("off must hold B+1 entries")
        if B is not None and off.shape[0] != B + 1:
            raise ValueError("off holds %d entries for %d jobs (expected B+1)" % (off.shape[0], B))
        self._arg(msgs, (None,), "u8", "msgs")
        return off.shape[0] - 1

    def _point(self, a, nbytes, B, name):
        """a per-job (B, nbytes) operand or ONE (nbytes,) operand broadcast to every job; returns the stride"""
        if len(a.shape) == 1:
            self._arg(a, (nbytes,), "u8", name)
            return 0
        self._arg(a, (B, nbytes), "u8", name)
        return nbytes

    # -- hashing ------------------------------------------------------------------------------
    def hash_g2(self, msgs, off):
        dev = self._mode(msgs, off)
        B = self._msgs(msgs, off)
        out = self._empty(dev, (B, G2_BYTES), ref=msgs)
        self._call("tc_hash_g2_batch", _ptr(msgs), _ptr(off), B, _ptr(out))
        return out

    def hash_g1_g2(self, g1, msgs, off):
        dev = self._mode(g1, msgs, off)
        B = self._msgs(msgs, off)
        self._arg(g1, (B, G1_BYTES), "u8", "g1")
        out = self._empty(dev, (B, G2_BYTES), ref=g1)
        st = self._empty(dev, (B,), ref=g1)
        self._call("tc_hash_g1_g2_batch", _ptr(g1), _ptr(msgs), _ptr(off), B, _ptr(out), _ptr(st))
        return out, st

    # -- scalar multiplication -------------------------------------------------------------------
    def _mul(self, name, pb, fr, pts):
        dev = self._mode(fr, pts)
        self._arg(fr, (None, FR_BYTES), "u8", "fr")
        self._arg(pts, (None, pb), "u8", "pts")
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

    def sign_shares_g2(self, sk_table, idx, hashes):
        """out[j, k] = sk_table[idx[j, k]] * hashes[j]: the shares of each message by its selected signers"""
        dev = self._mode(sk_table, idx, hashes)
        self._arg(sk_table, (None, FR_BYTES), "u8", "sk_table")
        self._arg(idx, (None, None), "u64", "idx")
        B, n = idx.shape
        self._arg(hashes, (B, G2_BYTES), "u8", "hashes")
        out = self._empty(dev, (B, n, G2_BYTES), ref=hashes)
        st = self._empty(dev, (B, n), ref=hashes)
        self._call("tc_sign_shares_g2_batch", _ptr(sk_table), sk_table.shape[0], _ptr(idx), _ptr(hashes), n, B, _ptr(out), _ptr(st))
        return out, st

    def sign(self, fr, msgs, off):
        dev = self._mode(fr, msgs, off)
        self._arg(fr, (None, FR_BYTES), "u8", "fr")
        S, B = fr.shape[0], self._msgs(msgs, off)
        out = self._empty(dev, (B, S, G2_BYTES), ref=fr)
        st = self._empty(dev, (B, S), ref=fr)
        self._call("tc_sign_batch", _ptr(fr), _ptr(msgs), _ptr(off), S, B, _ptr(out), _ptr(st))
        return out, st

    # -- combination --------------------------------------------------------------------------------
    def _combine(self, name, pb, t, idx, shares):
        dev = self._mode(idx, shares)
        self._arg(idx, (None, None), "u64", "idx")
        B, n = idx.shape
        self._arg(shares, (B, n, pb), "u8", "shares")
        out = self._empty(dev, (B, pb), ref=shares)
        st = self._empty(dev, (B,), ref=shares)
        self._call(name, int(t), int(n), _ptr(idx), _ptr(shares), B, _ptr(out), _ptr(st))
        return out, st

    def combine_g2(self, t, idx, shares):
        return self._combine("tc_combine_g2_batch", G2_BYTES, t, idx, shares)

    def combine_g1(self, t, idx, shares):
        return self._combine("tc_combine_g1_batch", G1_BYTES, t, idx, shares)

    def _combine_fr(self, name, pb, t, idx_fr, shares):
        """`T: IntoFr` abscissae as (B, n, 32) little-endian Fr values (src/into_fr.rs)"""
        dev = self._mode(idx_fr, shares)
        self._arg(idx_fr, (None, None, FR_BYTES), "u8", "idx_fr")
        B, n = idx_fr.shape[0], idx_fr.shape[1]
        self._arg(shares, (B, n, pb), "u8", "shares")
        out = self._empty(dev, (B, pb), ref=shares)
        st = self._empty(dev, (B,), ref=shares)
        self._call(name, int(t), int(n), _ptr(idx_fr), _ptr(shares), B, _ptr(out), _ptr(st))
        return out, st

    def combine_g2_fr(self, t, idx_fr, shares):
        return self._combine_fr("tc_combine_g2_fr_batch", G2_BYTES, t, idx_fr, shares)

    def combine_g1_fr(self, t, idx_fr, shares):
        return self._combine_fr("tc_combine_g1_fr_batch", G1_BYTES, t, idx_fr, shares)

    def combine_signatures_wire(self, t, idx, shares96):
        """PublicKeySet::combine_signatures on the wire forms: (B, n, 96) compressed shares in (checked decode of from_bytes on
        the device), (B, 96) Signature::to_bytes out"""
        dev = self._mode(idx, shares96)
        self._arg(idx, (None, None), "u64", "idx")
        B, n = idx.shape
        self._arg(shares96, (B, n, G2_BYTES // 2), "u8", "shares96")
        out = self._empty(dev, (B, G2_BYTES // 2), ref=shares96)
        st = self._empty(dev, (B,), ref=shares96)
        self._call("tc_combine_signatures_wire_batch", int(t), int(n), _ptr(idx), _ptr(shares96), B, _ptr(out), _ptr(st))
        return out, st

    def _lincomb(self, name, pb, scalars, points):
        dev = self._mode(scalars, points)
        self._arg(scalars, (None, None, FR_BYTES), "u8", "scalars")
        B, n = scalars.shape[0], scalars.shape[1]
        self._arg(points, (B, n, pb), "u8", "points")
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
        self._arg(idx, (None, None), "u64", "idx")
        B, n = idx.shape
        self._arg(shares_g1, (B, n, G1_BYTES), "u8", "shares")
        self._msgs(v, off, B)
        out = self._empty(dev, tuple(v.shape), ref=v)
        st = self._empty(dev, (B,), ref=v)
        self._call("tc_decrypt_batch", int(t), int(n), _ptr(idx), _ptr(shares_g1), _ptr(v), _ptr(off), B,
                   _ptr(out), _ptr(st))
        return out, st

    def decrypt_fr(self, t, idx_fr, shares_g1, v, off):
        dev = self._mode(idx_fr, shares_g1, v, off)
        self._arg(idx_fr, (None, None, FR_BYTES), "u8", "idx_fr")
        B, n = idx_fr.shape[0], idx_fr.shape[1]
        self._arg(shares_g1, (B, n, G1_BYTES), "u8", "shares")
        self._msgs(v, off, B)
        out = self._empty(dev, tuple(v.shape), ref=v)
        st = self._empty(dev, (B,), ref=v)
        self._call("tc_decrypt_fr_batch", int(t), int(n), _ptr(idx_fr), _ptr(shares_g1), _ptr(v), _ptr(off), B, _ptr(out), _ptr(st))
        return out, st

    def decrypt_wire(self, t, idx, shares48, v, off):
        """PublicKeySet::decrypt with the decryption shares in their 48-byte compressed form (checked decode on the device)"""
        dev = self._mode(idx, shares48, v, off)
        self._arg(idx, (None, None), "u64", "idx")
        B, n = idx.shape
        self._arg(shares48, (B, n, G1_BYTES // 2), "u8", "shares48")
        self._msgs(v, off, B)
        out = self._empty(dev, tuple(v.shape), ref=v)
        st = self._empty(dev, (B,), ref=v)
        self._call("tc_decrypt_wire_batch", int(t), int(n), _ptr(idx), _ptr(shares48), _ptr(v), _ptr(off), B, _ptr(out), _ptr(st))
        return out, st

    def xor_with_hash(self, g1, data, off):
        dev = self._mode(g1, data, off)
        B = self._msgs(data, off)
        self._arg(g1, (B, G1_BYTES), "u8", "g1")
        out = self._empty(dev, tuple(data.shape), ref=data)
        st = self._empty(dev, (B,), ref=data)
        self._call("tc_xor_with_hash_batch", _ptr(g1), _ptr(data), _ptr(off), B, _ptr(out), _ptr(st))
        return out, st

    # -- pairing checks -------------------------------------------------------------------------------
    def pairing_check(self, a, b, c, d, B=None):
        """ok[j] = e(a[j], b[j]) == e(c[j], d[j]); 1-D operands are broadcast to every job."""
        dev = self._mode(a, b, c, d)
        if B is None:
            B = max(x.shape[0] if len(x.shape) == 2 else 1 for x in (a, b, c, d))
        sa, sb = self._point(a, G1_BYTES, B, "a"), self._point(b, G2_BYTES, B, "b")
        sc, sd = self._point(c, G1_BYTES, B, "c"), self._point(d, G2_BYTES, B, "d")
        ok = self._empty(dev, (B,), ref=a)
        self._call("tc_pairing_check_batch", _ptr(a), sa, _ptr(b), sb, _ptr(c), sc, _ptr(d), sd, B, _ptr(ok))
        return ok

    def verify_g2(self, pk, sig, hash_):
        dev = self._mode(pk, sig, hash_)
        self._arg(sig, (None, G2_BYTES), "u8", "sig")
        B = sig.shape[0]
        self._arg(hash_, (B, G2_BYTES), "u8", "hash")
        ok = self._empty(dev, (B,), ref=sig)
        self._call("tc_verify_g2_batch", _ptr(pk), self._point(pk, G1_BYTES, B, "pk"), _ptr(sig), _ptr(hash_), B, _ptr(ok))
        return ok

    def verify_sig(self, pk, sig, msgs, off):
        dev = self._mode(pk, sig, msgs, off)
        self._arg(sig, (None, G2_BYTES), "u8", "sig")
        B = sig.shape[0]
        self._msgs(msgs, off, B)
        ok = self._empty(dev, (B,), ref=sig)
        self._call("tc_verify_sig_batch", _ptr(pk), self._point(pk, G1_BYTES, B, "pk"), _ptr(sig), _ptr(msgs), _ptr(off), B,
                   _ptr(ok))
        return ok

    def verify_shares_rlc(self, pk_shares, sig_shares, msgs, off, seed=None):
        """Share validation by one random linear combination per message (opt-in; see tc_amd.h): returns
        (ok (B, N), number of messages that fell back to per-share checks).  `seed`: 32 secret random bytes
        (os.urandom when omitted)."""
        import os
        dev = self._mode(pk_shares, sig_shares, msgs, off)
        self._arg(pk_shares, (None, G1_BYTES), "u8", "pk_shares")
        N = pk_shares.shape[0]
        B = self._msgs(msgs, off)
        self._arg(sig_shares, (B, N, G2_BYTES), "u8", "sig_shares")
        seed = bytes(seed) if seed is not None else os.urandom(32)
        if len(seed) != 32:
            raise ValueError("seed: 32 bytes")
        ok = self._empty(dev, (B, N), ref=sig_shares)
        nfb = ctypes.c_uint64(0)
        self._call("tc_verify_shares_rlc_batch", _ptr(pk_shares), N, _ptr(sig_shares), _ptr(msgs), _ptr(off), B, seed, _ptr(ok),
                   ctypes.byref(nfb))
        return ok, int(nfb.value)

    def verify_g2_rlc(self, pk, sig, hash_, group=0, seed=None):
        """Same-key signature batch by random linear combination (opt-in; see tc_amd.h): returns (ok (B,), number of jobs
        that fell back to per-job checks).  `seed`: 32 secret random bytes (os.urandom when omitted)."""
        import os
        dev = self._mode(pk, sig, hash_)
        self._arg(pk, (G1_BYTES,), "u8", "pk")
        self._arg(sig, (None, G2_BYTES), "u8", "sig")
        B = sig.shape[0]
        self._arg(hash_, (B, G2_BYTES), "u8", "hash")
        seed = bytes(seed) if seed is not None else os.urandom(32)
        if len(seed) != 32:
            raise ValueError("seed: 32 bytes")
        ok = self._empty(dev, (B,), ref=sig)
        nfb = ctypes.c_uint64(0)
        self._call("tc_verify_g2_rlc_batch", _ptr(pk), _ptr(sig), _ptr(hash_), B, int(group), seed, _ptr(ok), ctypes.byref(nfb))
        return ok, int(nfb.value)

    def verify_sig_rlc(self, pk, sig, msgs, off, group=0, seed=None):
        import os
        dev = self._mode(pk, sig, msgs, off)
        self._arg(pk, (G1_BYTES,), "u8", "pk")
        self._arg(sig, (None, G2_BYTES), "u8", "sig")
        B = sig.shape[0]
        self._msgs(msgs, off, B)
        seed = bytes(seed) if seed is not None else os.urandom(32)
        if len(seed) != 32:
            raise ValueError("seed: 32 bytes")
        ok = self._empty(dev, (B,), ref=sig)
        nfb = ctypes.c_uint64(0)
        self._call("tc_verify_sig_rlc_batch", _ptr(pk), _ptr(sig), _ptr(msgs), _ptr(off), B, int(group), seed, _ptr(ok), ctypes.byref(nfb))
        return ok, int(nfb.value)

    def ciphertext_verify(self, u, v, off, w):
        dev = self._mode(u, v, off, w)
        self._arg(u, (None, G1_BYTES), "u8", "u")
        B = u.shape[0]
        self._msgs(v, off, B)
        self._arg(w, (B, G2_BYTES), "u8", "w")
        ok = self._empty(dev, (B,), ref=u)
        self._call("tc_ciphertext_verify_batch", _ptr(u), _ptr(v), _ptr(off), _ptr(w), B, _ptr(ok))
        return ok

    def decrypt_share(self, fr, u, v, off, w):
        """SecretKeyShare::decrypt_share (src/lib.rs:452-457) for one key share and B ciphertexts -> (shares (B, 96), ok (B,)):
        ok[j] = Ciphertext::verify, the share of a ciphertext that fails it is the identity (the reference returns None)."""
        dev = self._mode(fr, u, v, off, w)
        self._arg(fr, (FR_BYTES,), "u8", "fr")
        self._arg(u, (None, G1_BYTES), "u8", "u")
        B = u.shape[0]
        self._msgs(v, off, B)
        self._arg(w, (B, G2_BYTES), "u8", "w")
        out = self._empty(dev, (B, G1_BYTES), ref=u)
        ok = self._empty(dev, (B,), ref=u)
        self._call("tc_decrypt_share_batch", _ptr(fr), _ptr(u), _ptr(v), _ptr(off), _ptr(w), B, _ptr(out), _ptr(ok))
        return out, ok

    def secret_key_decrypt(self, fr, u, v, off, w):
        """SecretKey::decrypt (src/lib.rs:384-391) for B ciphertexts -> (plaintext bytes laid out like v, ok (B,)); the bytes
        of a ciphertext that fails Ciphertext::verify are zeros (None)."""
        dev = self._mode(fr, u, v, off, w)
        self._arg(fr, (FR_BYTES,), "u8", "fr")
        self._arg(u, (None, G1_BYTES), "u8", "u")
        B = u.shape[0]
        self._msgs(v, off, B)
        self._arg(w, (B, G2_BYTES), "u8", "w")
        out = self._empty(dev, tuple(v.shape), ref=v)
        ok = self._empty(dev, (B,), ref=u)
        self._call("tc_secret_key_decrypt_batch", _ptr(fr), _ptr(u), _ptr(v), _ptr(off), _ptr(w), B, _ptr(out), _ptr(ok))
        return out, ok

    def verify_decryption_share(self, pk_share, share, u, v, off, w):
        dev = self._mode(pk_share, share, u, v, off, w)
        self._arg(share, (None, G1_BYTES), "u8", "share")
        B = share.shape[0]
        self._arg(u, (B, G1_BYTES), "u8", "u")
        self._arg(w, (B, G2_BYTES), "u8", "w")
        self._msgs(v, off, B)
        ok = self._empty(dev, (B,), ref=share)
        self._call("tc_verify_decryption_share_batch", _ptr(pk_share), self._point(pk_share, G1_BYTES, B, "pk_share"),
                   _ptr(share), _ptr(u), _ptr(v), _ptr(off), _ptr(w), B, _ptr(ok))
        return ok

    def verify_decryption_shares_rlc(self, pk_shares, shares, u, v, off, w, seed=None):
        """Decryption-share validation by one random linear combination per ciphertext (opt-in; see tc_amd.h): returns
        (ok (B, N), number of ciphertexts that fell back to per-share checks)."""
        import os
        dev = self._mode(pk_shares, shares, u, v, off, w)
        self._arg(pk_shares, (None, G1_BYTES), "u8", "pk_shares")
        N = pk_shares.shape[0]
        self._arg(u, (None, G1_BYTES), "u8", "u")
        B = u.shape[0]
        self._arg(shares, (B, N, G1_BYTES), "u8", "shares")
        self._arg(w, (B, G2_BYTES), "u8", "w")
        self._msgs(v, off, B)
        seed = bytes(seed) if seed is not None else os.urandom(32)
        if len(seed) != 32:
            raise ValueError("seed: 32 bytes")
        ok = self._empty(dev, (B, N), ref=shares)
        nfb = ctypes.c_uint64(0)
        self._call("tc_verify_decryption_shares_rlc_batch", _ptr(pk_shares), N, _ptr(shares), _ptr(u), _ptr(v), _ptr(off), _ptr(w), B, seed,
                   _ptr(ok), ctypes.byref(nfb))
        return ok, int(nfb.value)

    # -- membership tests -----------------------------------------------------------------------------
    def g1_subgroup_check(self, pts):
        """ok[j] = pts[j] is a valid encoding of a point of G1 (on the curve, order r)"""
        dev = self._mode(pts)
        self._arg(pts, (None, G1_BYTES), "u8", "pts")
        ok = self._empty(dev, (pts.shape[0],), ref=pts)
        self._call("tc_g1_subgroup_check_batch", _ptr(pts), pts.shape[0], _ptr(ok))
        return ok

    def g2_subgroup_check(self, pts):
        dev = self._mode(pts)
        self._arg(pts, (None, G2_BYTES), "u8", "pts")
        ok = self._empty(dev, (pts.shape[0],), ref=pts)
        self._call("tc_g2_subgroup_check_batch", _ptr(pts), pts.shape[0], _ptr(ok))
        return ok

    # -- wire formats ------------------------------------------------------------------------------------
    def _recode(self, name, a, nin, nout):
        dev = self._mode(a)
        self._arg(a, (None, nin), "u8", "points")
        B = a.shape[0]
        out = self._empty(dev, (B, nout), ref=a)
        st = self._empty(dev, (B,), ref=a)
        self._call(name, _ptr(a), B, _ptr(out), _ptr(st))
        return out, st

    def g1_compress(self, pts):
        return self._recode("tc_g1_compress_batch", pts, G1_BYTES, 48)

    def g2_compress(self, pts):
        return self._recode("tc_g2_compress_batch", pts, G2_BYTES, 96)

    def g1_decompress(self, comp):
        """checked decode of 48-byte compressed G1 (from_bytes): status 3 = invalid"""
        return self._recode("tc_g1_decompress_batch", comp, 48, G1_BYTES)

    def g2_decompress(self, comp):
        return self._recode("tc_g2_decompress_batch", comp, 96, G2_BYTES)

    def encrypt(self, pk, r, msgs, off):
        """PublicKey::encrypt_with_rng with the Fr draws supplied: returns (u, v, w, status)."""
        dev = self._mode(pk, r, msgs, off)
        B = self._msgs(msgs, off)
        self._arg(r, (B, FR_BYTES), "u8", "r")
        u = self._empty(dev, (B, G1_BYTES), ref=r)
        v = self._empty(dev, tuple(msgs.shape), ref=r)
        w = self._empty(dev, (B, G2_BYTES), ref=r)
        st = self._empty(dev, (B,), ref=r)
        self._call("tc_encrypt_batch", _ptr(pk), self._point(pk, G1_BYTES, B, "pk"), _ptr(r), _ptr(msgs), _ptr(off), B, _ptr(u),
                   _ptr(v), _ptr(w), _ptr(st))
        return u, v, w, st

    def public_key_shares(self, commit, idx):
        """Commitment::evaluate(idx + 1) for every index: (M, 96)."""
        dev = self._mode(commit, idx)
        self._arg(commit, (None, G1_BYTES), "u8", "commit")
        self._arg(idx, (None,), "u64", "idx")
        M = idx.shape[0]
        t = commit.shape[0] - 1
        out = self._empty(dev, (M, G1_BYTES), ref=commit)
        st = self._empty(dev, (M,), ref=commit)
        self._call("tc_public_key_share_batch", _ptr(commit), int(t), _ptr(idx), M, _ptr(out), _ptr(st))
        return out, st

    # -- DKG algebra (src/poly.rs) ------------------------------------------------------------------------
    def g1_commitment(self, coeff_fr):
        """out[i] = coeff_fr[i] * g1 (Poly::commitment / BivarPoly::commitment): fixed-base, LDS window table"""
        dev = self._mode(coeff_fr)
        self._arg(coeff_fr, (None, FR_BYTES), "u8", "coeff_fr")
        M = coeff_fr.shape[0]
        out = self._empty(dev, (M, G1_BYTES), ref=coeff_fr)
        st = self._empty(dev, (M,), ref=coeff_fr)
        self._call("tc_g1_commitment_batch", _ptr(coeff_fr), M, _ptr(out), _ptr(st))
        return out, st

    def bivar_commitment_rows(self, commit, degree, xs):
        """BivarCommitment::row(xs[m]) for every m: (M, degree+1, 96)"""
        dev = self._mode(commit, xs)
        n = (degree + 1) * (degree + 2) // 2
        self._arg(commit, (n, G1_BYTES), "u8", "commit")
        self._arg(xs, (None,), "u64", "xs")
        M = xs.shape[0]
        out = self._empty(dev, (M, degree + 1, G1_BYTES), ref=commit)
        st = self._empty(dev, (M, degree + 1), ref=commit)
        self._call("tc_bivar_commitment_row_batch", _ptr(commit), int(degree), _ptr(xs), M, _ptr(out), _ptr(st))
        return out, st

    def fr_interpolate(self, xs, ys):
        """Poly::interpolate for B jobs of n samples: xs, ys (B, n, 32) -> coefficients (B, n, 32), status (B,)"""
        dev = self._mode(xs, ys)
        self._arg(xs, (None, None, FR_BYTES), "u8", "xs")
        B, n = xs.shape[0], xs.shape[1]
        self._arg(ys, (B, n, FR_BYTES), "u8", "ys")
        out = self._empty(dev, (B, n, FR_BYTES), ref=xs)
        st = self._empty(dev, (B,), ref=xs)
        self._call("tc_fr_interpolate_batch", int(n), _ptr(xs), _ptr(ys), B, _ptr(out), _ptr(st))
        return out, st


class Group:
    """tc_group: several GPUs of one node driven from this process (one worker thread per GPU inside the library),
    host-memory batches sharded contiguously, key-set broadcast and valid-count all-reduce on RCCL."""

    def __init__(self, devices):
        self._lib = _native.load()
        g = ctypes.c_void_p()
        arr = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._lib.tc_group_create(ctypes.byref(g), arr, len(devices))
        if rc != _native.TC_OK:
            raise TcError(rc, "tc_group_create failed for devices %s" % (list(devices),))
        self._g = g
        self.t = None

    def close(self):
        if getattr(self, "_g", None):
            self._lib.tc_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        rc = getattr(self._lib, name)(self._g, *args)
        if rc != _native.TC_OK:
            raise TcError(rc, self._lib.tc_group_last_error(self._g).decode())

    def size(self):
        return int(self._lib.tc_group_size(self._g))

    def uses_rccl(self):
        return bool(self._lib.tc_group_uses_rccl(self._g))

    def shard(self, B, rank):
        s, c = ctypes.c_size_t(), ctypes.c_size_t()
        self._call("tc_group_shard", int(B), int(rank), ctypes.byref(s), ctypes.byref(c))
        return int(s.value), int(c.value)

    def set_keyset(self, commit):
        Engine._arg(commit, (None, G1_BYTES), "u8", "commit")
        self._call("tc_group_set_keyset", commit.shape[0] - 1, _ptr(commit))
        self.t = commit.shape[0] - 1

    def get_keyset(self, rank):
        out = np.empty((self.t + 1, G1_BYTES), dtype=np.uint8)
        self._call("tc_group_get_keyset", int(rank), _ptr(out))
        return out

    def combine_signatures(self, idx, shares):
        Engine._arg(idx, (None, None), "u64", "idx")
        B, n = idx.shape
        Engine._arg(shares, (B, n, G2_BYTES), "u8", "shares")
        out = np.empty((B, G2_BYTES), dtype=np.uint8)
        st = np.empty(B, dtype=np.uint8)
        self._call("tc_group_combine_signatures", n, _ptr(idx), _ptr(shares), B, _ptr(out), _ptr(st))
        return out, st

    def verify_g2(self, sig, hashes):
        Engine._arg(sig, (None, G2_BYTES), "u8", "sig")
        B = sig.shape[0]
        Engine._arg(hashes, (B, G2_BYTES), "u8", "hashes")
        ok = np.empty(B, dtype=np.uint8)
        nv = ctypes.c_uint64(0)
        self._call("tc_group_verify_g2", _ptr(sig), _ptr(hashes), B, _ptr(ok), ctypes.byref(nv))
        return ok, int(nv.value)

    def transfer_bytes(self):
        """(host-to-device, device-to-host) bytes the group and its contexts have moved over PCIe"""
        up, down = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._call("tc_group_transfer_bytes", ctypes.byref(up), ctypes.byref(down))
        return int(up.value), int(down.value)

    def sign_combine_verify(self, sk_table, idx, msgs, off):
        Engine._arg(sk_table, (None, FR_BYTES), "u8", "sk_table")
        Engine._arg(idx, (None, None), "u64", "idx")
        B, n = idx.shape
        # the same operand validation as the single-GPU Engine: dtype, contiguity, B + 1 offsets
        Engine._arg(off, (B + 1,), "u64", "off")
        Engine._arg(msgs, (None,), "u8", "msgs")
        if int(off[0]) != 0 or (np.diff(off.astype(np.int64)) < 0).any() or int(off[-1]) > msgs.shape[0]:
            raise ValueError("off must start at 0, be non-decreasing and end inside msgs")
        sig = np.empty((B, G2_BYTES), dtype=np.uint8)
        ok = np.empty(B, dtype=np.uint8)
        nv = ctypes.c_uint64(0)
        self._call("tc_group_sign_combine_verify", _ptr(sk_table), sk_table.shape[0], _ptr(idx), n, _ptr(msgs), _ptr(off), B, _ptr(sig),
                   _ptr(ok), ctypes.byref(nv))
        return sig, ok, int(nv.value)
