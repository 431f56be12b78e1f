"""Synthetic (t, N, batch) workloads of BASELINE.json, generated ON the GPU through the product
path itself (so a 65 536-job batch needs no CPU curve arithmetic) and deterministic in the seed
(SURVEY.md 8d).  Used by bench.py, __graft_entry__.smoke() and the full-size property tests.

Key set:   a_k = LE(SHA3-256("tc/key" || seed_le64 || k_le32)) mod r,  sk_i = f(i+1)
Messages:  m_j = "tc/msg" || j_le64
Subsets:   per job j a Fisher-Yates shuffle of [0, N) driven by splitmix64(seed ^ j), first t+1
           entries, sorted ascending (BTreeMap iteration order).
"""
import hashlib

import numpy as np

from .api import SecretKeySet
from .engine import pack_messages

SEED = 0x7C5EED
_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def key_set(t, seed=SEED):
    coeffs = []
    for k in range(t + 1):
        d = hashlib.sha3_256(b"tc/key" + seed.to_bytes(8, "little") + k.to_bytes(4, "little")).digest()
        coeffs.append(int.from_bytes(d, "little") % _R)
    return SecretKeySet(coeffs)


def messages(B, start=0):
    return [b"tc/msg" + int(j).to_bytes(8, "little") for j in range(start, start + B)]


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x, z ^ (z >> 31)


def signer_subsets(B, N, t, seed=SEED, start=0):
    """(B, t+1) uint64 ascending signer indices per job."""
    out = np.empty((B, t + 1), dtype=np.uint64)
    for j in range(B):
        state = seed ^ (start + j)
        perm = list(range(N))
        for i in range(N - 1, N - 2 - t, -1):  # partial Fisher-Yates: last t+1 slots
            state, r = _splitmix64(state)
            k = r % (i + 1)
            perm[i], perm[k] = perm[k], perm[i]
        out[j] = sorted(perm[N - 1 - t:])
    return out


class ThresholdSigWorkload:
    """BASELINE config "t, N, batch threshold signatures": hash points, all N shares of every
    message and the per-job (t+1)-subsets, resident in host numpy arrays (moved to HBM by the
    caller).  `prehashed=True` replaces hash_g2(m_j) by h_j * G2 with h_j = j+1 scrambled, which
    decouples the determined arithmetic from the H-spec (SURVEY.md 8c "Decoupling")."""

    def __init__(self, engine, t, N, B, seed=SEED, start=0, chunk=8192, index_offset=0, hashes=None):
        """index_offset: the N signers are the nodes index_offset .. index_offset + N - 1 (share indices of 65 535 and
        above leave the combiner's small-index fast path: bench.py's general-path leg); hashes: the hash points of
        the same messages when the caller already has them."""
        self.t, self.N, self.B = t, N, B
        self.sks = key_set(t, seed)
        self.shares_sk = [self.sks.secret_key_share(index_offset + i) for i in range(N)]
        fr = np.stack([np.frombuffer(s._bytes(), dtype=np.uint8) for s in self.shares_sk])
        self.msgs = messages(B, start)
        flat, off = pack_messages(self.msgs)
        self.msg_flat, self.msg_off = flat, off
        self.hashes = engine.hash_g2(flat, off) if hashes is None else hashes   # (B, 192)
        self.idx = signer_subsets(B, N, t, seed, start)              # (B, t+1): positions in the signer table
        # only the t+1 selected shares of each job are needed by combine; sign all N per message in
        # chunks (S x B lanes) and gather the selected ones
        sel = np.empty((B, t + 1, 192), dtype=np.uint8)
        for lo in range(0, B, chunk):
            hi = min(B, lo + chunk)
            allsh, st = engine.g2_mul(fr, np.ascontiguousarray(self.hashes[lo:hi]))   # (b, N, 192)
            assert not st.any()
            rows = np.arange(hi - lo)[:, None]
            sel[lo:hi] = allsh[rows, self.idx[lo:hi].astype(np.int64)]
        self.shares = sel
        self.idx = self.idx + np.uint64(index_offset)                # the share indices the combiner sees
        pk, st = engine.g1_mul(np.frombuffer(self.sks.poly[0].to_bytes(32, "little"), dtype=np.uint8)[None].copy(),
                               np.frombuffer(_g1_gen(), dtype=np.uint8)[None].copy())
        self.master_pk = np.ascontiguousarray(pk[0, 0])
        self.master_sk_fr = np.frombuffer(self.sks.poly[0].to_bytes(32, "little"), dtype=np.uint8).copy()


def _g1_gen():
    from .api import _G1_GEN
    return _G1_GEN


def _sha3_scalars(tag, B, seed, start=0):
    out = np.empty((B, 32), dtype=np.uint8)
    for j in range(B):
        d = hashlib.sha3_256(tag + seed.to_bytes(8, "little") + (start + j).to_bytes(8, "little")).digest()
        out[j] = np.frombuffer((int.from_bytes(d, "little") % _R).to_bytes(32, "little"), dtype=np.uint8)
    return out


class ThresholdEncWorkload:
    """BASELINE config "t, N, batch threshold decryptions": ciphertexts under the master key
    (PublicKey::encrypt_with_rng, src/lib.rs:128-137, composed from the batch entry points with
    r_j = LE(SHA3-256("tc/enc" || seed || j)) mod r and 32-byte plaintexts SHA3-256("tc/pt" || j)),
    and the t+1 selected decryption shares of every job."""

    def __init__(self, engine, t, N, B, seed=SEED, start=0, chunk=16384):
        self.t, self.N, self.B = t, N, B
        self.sks = key_set(t, seed)
        sk_shares = [self.sks.secret_key_share(i) for i in range(N)]
        fr = np.stack([np.frombuffer(s._bytes(), dtype=np.uint8) for s in sk_shares])
        g1 = np.frombuffer(_g1_gen(), dtype=np.uint8)
        pk, _ = engine.g1_mul(np.frombuffer(self.sks.poly[0].to_bytes(32, "little"), dtype=np.uint8)[None].copy(), g1[None].copy())
        self.master_pk = np.ascontiguousarray(pk[0, 0])
        r = _sha3_scalars(b"tc/enc", B, seed, start)[:, None, :]                       # (B, 1, 32)
        self.plain = [hashlib.sha3_256(b"tc/pt" + (start + j).to_bytes(8, "little")).digest() for j in range(B)]
        pt_flat, off = pack_messages(self.plain)
        self.off = off
        tile = lambda p: np.ascontiguousarray(np.broadcast_to(p[None, None, :], (B, 1, p.shape[0])))
        self.u, st = engine.lincomb_g1(np.ascontiguousarray(r), tile(g1))              # u = r * g1
        g, st2 = engine.lincomb_g1(np.ascontiguousarray(r), tile(self.master_pk))      # g = r * pk
        assert not st.any() and not st2.any()
        self.v, st3 = engine.xor_with_hash(g, pt_flat, off)                            # v = m ^ H(g)
        h, st4 = engine.hash_g1_g2(self.u, self.v, off)
        self.w, st5 = engine.lincomb_g2(np.ascontiguousarray(r), np.ascontiguousarray(h[:, None, :]))  # w = r * H(u, v)
        assert not st3.any() and not st4.any() and not st5.any()
        self.idx = signer_subsets(B, N, t, seed ^ 0x5EED, start)
        sel = np.empty((B, t + 1, 96), dtype=np.uint8)
        for lo in range(0, B, chunk):
            hi = min(B, lo + chunk)
            allsh, st6 = engine.g1_mul(fr, np.ascontiguousarray(self.u[lo:hi]))       # (b, N, 96)
            assert not st6.any()
            rows = np.arange(hi - lo)[:, None]
            sel[lo:hi] = allsh[rows, self.idx[lo:hi].astype(np.int64)]
        self.shares = sel
        self.plain_flat = pt_flat
