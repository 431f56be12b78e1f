"""Host-side mirror of the threshold_crypto 0.4.0 public API for the accelerated path.

Same names, argument meaning and error behaviour as the Rust crate (file:line cited per
method, relative to the reference repository), plus `*_batch` forms that hand a whole batch to
one kernel launch.  The single-item methods are batches of one, so a reference test reads the
same here.  Every group/pairing/hash computation goes through libtc_amd.so (HIP, gfx950);
this module only packs bytes, orders shares (BTreeMap order) and maps status codes to the
reference's error types.  It never imports the oracle and has no CPU fallback.

Values are held in the reference's own canonical encodings: G1 96 B / G2 192 B uncompressed
(`into_affine().into_uncompressed()`), Fr 32 B little-endian.
"""
import numpy as np

from .engine import Engine, pack_messages

PK_SIZE = 48   # src/lib.rs:71
SIG_SIZE = 96  # src/lib.rs:75

# Fr modulus: only used for key-set bookkeeping on secret polynomials (SecretKeySet), which the
# reference also does on the CPU outside the hot path (src/poly.rs:358-369).
_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001

_G1_GEN = bytes.fromhex(
    "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
    "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")


class Error(Exception):
    """threshold_crypto::error::Error (src/error.rs:7-17)."""


class NotEnoughShares(Error):
    """Error::NotEnoughShares (src/error.rs:9-10)."""


class DuplicateEntry(Error):
    """Error::DuplicateEntry (src/error.rs:12-13)."""


class FromBytesError(Exception):
    """FromBytesError::Invalid (src/error.rs:37-41)."""


_STATUS_EXC = {1: NotEnoughShares, 2: DuplicateEntry, 3: FromBytesError}

_default_engine = None


def default_engine():
    """The process-wide Engine on GPU 0 (created on first use; raises without a GPU)."""
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


def set_default_engine(engine):
    global _default_engine
    _default_engine = engine


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def _stack(items, width):
    out = np.empty((len(items), width), dtype=np.uint8)
    for i, it in enumerate(items):
        out[i] = np.frombuffer(bytes(it), dtype=np.uint8)
    return out


def _raise_status(st):
    st = int(st)
    if st:
        raise _STATUS_EXC.get(st, Error)("job status %d" % st)


def into_fr_plus_1(i):
    """src/lib.rs:769-773 for every IntoFr impl (src/into_fr.rs:10-56)."""
    return ((i.v if isinstance(i, Fr) else int(i)) + 1) % _R


class Fr:
    """A scalar-field element as an index type: `impl IntoFr for Fr` (src/into_fr.rs:10-14).  Ordered by its canonical value,
    like the derived Ord of pairing's Fr, so a dict keyed by Fr iterates as the reference's BTreeMap does."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = int(v) % _R

    def __int__(self):
        return self.v

    def __eq__(self, o):
        return isinstance(o, Fr) and o.v == self.v

    def __lt__(self, o):
        return self.v < o.v

    def __hash__(self):
        return hash(("Fr", self.v))

    def __repr__(self):
        return "Fr(%#x)" % self.v


def into_fr(i):
    """IntoFr (src/into_fr.rs:10-56): Fr by value, u64 / usize by value, negative i32 / i64 as -(|x|) mod r."""
    return i.v if isinstance(i, Fr) else int(i) % _R


def _index_arrays(ordered, n):
    """(u64 array, None) when every index is a plain integer in [0, 2^64) -- the u64 entries --, else (None, (B, n, 32) Fr
    bytes) for the `T: IntoFr` entries (Fr keys, negative integers)."""
    flat = [i for o in ordered for i, _ in o]
    if all(not isinstance(i, Fr) and 0 <= int(i) < 2 ** 64 for i in flat):
        return np.array([[int(i) for i, _ in o] for o in ordered], dtype=np.uint64).reshape(len(ordered), n), None
    fr = np.zeros((len(ordered), n, 32), dtype=np.uint8)
    for j, o in enumerate(ordered):
        for k, (i, _) in enumerate(o):
            fr[j, k] = np.frombuffer(into_fr(i).to_bytes(32, "little"), dtype=np.uint8)
    return None, fr


def _require_members(engine, g2, rows):
    """The reference only holds G1/G2 values that passed the checked decode (on the curve AND in the
    order-r subgroup, src/lib.rs:140-146, 246-252); the kernels rely on it.  Raw uncompressed bytes that
    did not come out of this library are therefore tested on the device before they become a value."""
    e = engine or default_engine()
    ok = (e.g2_subgroup_check if g2 else e.g1_subgroup_check)(rows)
    if not ok.all():
        raise FromBytesError("point is not a valid member of the order-r subgroup")


class _G1Value:
    """A group element in its uncompressed encoding.  `_trusted=True` is for values this library produced
    itself (kernel outputs); anything else is validated like from_bytes validates (see _require_members)."""
    __slots__ = ("raw",)
    SIZE = 96
    _G2 = False

    def __init__(self, raw, _trusted=False):
        raw = bytes(raw)
        if len(raw) != self.SIZE:
            raise ValueError("expected %d bytes" % self.SIZE)
        if not _trusted:
            _require_members(None, self._G2, _u8(raw)[None])
        self.raw = raw

    def __eq__(self, other):
        return type(other) is type(self) and other.raw == self.raw

    def __hash__(self):
        return hash(self.raw)

    def to_bytes(self):
        """Compressed form (to_bytes, src/lib.rs:149-153)."""
        out, st = default_engine().g1_compress(_u8(self.raw)[None])
        _raise_status(st[0])
        return bytes(out[0])


class _G2Value(_G1Value):
    SIZE = 192
    _G2 = True

    def to_bytes(self):
        """Compressed form (to_bytes, src/lib.rs:255-259)."""
        out, st = default_engine().g2_compress(_u8(self.raw)[None])
        _raise_status(st[0])
        return bytes(out[0])


def _from_bytes(cls, data, size, fn_name):
    data = bytes(data)
    if len(data) != size:
        raise FromBytesError("expected %d bytes" % size)
    out, st = getattr(default_engine(), fn_name)(_u8(data)[None])
    if int(st[0]):
        raise FromBytesError("invalid encoding")
    return cls(out[0], _trusted=True)  # the checked decode ran on the device


class Signature(_G2Value):
    """struct Signature(G2) (src/lib.rs:202)."""

    @classmethod
    def from_bytes(cls, data):
        """Signature::from_bytes (src/lib.rs:246-252): checked decode of the 96-byte form."""
        return _from_bytes(cls, data, SIG_SIZE, "g2_decompress")

    def parity(self):
        """Signature::parity (src/lib.rs:237-243): xor-fold of the uncompressed bytes."""
        x = 0
        for b in self.raw:
            x ^= b
        return bin(x).count("1") % 2 != 0


class SignatureShare(Signature):
    """struct SignatureShare(pub Signature) (src/lib.rs:266)."""


class DecryptionShare(_G1Value):
    """struct DecryptionShare(G1) (src/lib.rs:517)."""


class Ciphertext:
    """struct Ciphertext(G1, Vec<u8>, G2) (src/lib.rs:473-478)."""

    def __init__(self, u, v, w, _trusted=False):
        self.u, self.v, self.w = bytes(u), bytes(v), bytes(w)
        if len(self.u) != 96 or len(self.w) != 192:
            raise ValueError("Ciphertext(u: 96 bytes, v, w: 192 bytes)")
        if not _trusted:
            _require_members(None, False, _u8(self.u)[None])
            _require_members(None, True, _u8(self.w)[None])

    def verify(self):
        """Ciphertext::verify (src/lib.rs:508-512)."""
        return bool(Ciphertext.verify_batch([self])[0])

    @staticmethod
    def verify_batch(cts, engine=None):
        e = engine or default_engine()
        v, off = pack_messages([c.v for c in cts])
        return e.ciphertext_verify(_stack([c.u for c in cts], 96), v, off, _stack([c.w for c in cts], 192)).astype(bool)


class PublicKey(_G1Value):
    """struct PublicKey(G1) (src/lib.rs:79)."""

    @classmethod
    def from_bytes(cls, data):
        """PublicKey::from_bytes (src/lib.rs:140-146): checked decode of the 48-byte form."""
        return _from_bytes(cls, data, PK_SIZE, "g1_decompress")

    def encrypt_with_r(self, r, msg):
        """PublicKey::encrypt_with_rng (src/lib.rs:128-137) with the rng's Fr draw `r` given."""
        return self.encrypt_with_r_batch([r], [msg])[0]

    def encrypt_with_r_batch(self, rs, msgs, engine=None):
        e = engine or default_engine()
        flat, off = pack_messages([bytes(m) for m in msgs])
        rr = _stack([(int(x) % _R).to_bytes(32, "little") for x in rs], 32)
        u, v, w, st = e.encrypt(_u8(self.raw), rr, flat, off)
        for s_ in st:
            _raise_status(s_)
        return [Ciphertext(u[j], bytes(v[int(off[j]): int(off[j + 1])]), w[j], _trusted=True) for j in range(len(msgs))]

    def verify_g2(self, sig, hash_g2_point):
        """PublicKey::verify_g2 (src/lib.rs:108-110)."""
        return bool(self.verify_g2_batch([sig], [hash_g2_point])[0])

    def verify(self, sig, msg):
        """PublicKey::verify (src/lib.rs:115-117)."""
        return bool(self.verify_batch([sig], [msg])[0])

    def verify_g2_batch(self, sigs, hashes, engine=None):
        e = engine or default_engine()
        if len(sigs) != len(hashes):
            raise ValueError("one hash point per signature")
        return e.verify_g2(_u8(self.raw), _stack([s.raw for s in sigs], 192), _stack(hashes, 192)).astype(bool)

    def verify_batch(self, sigs, msgs, engine=None, rlc=False, seed=None):
        """PublicKey::verify for many (signature, message) pairs under THIS key.  rlc=True: the opt-in random linear
        combination per group of 64 jobs (tc_verify_sig_rlc_batch; groups that fail are re-checked job by job, so the
        booleans are the per-job ones up to 2^-63); `seed`: 32 secret random bytes drawn after the signatures arrived."""
        e = engine or default_engine()
        if len(sigs) != len(msgs):
            raise ValueError("one message per signature")
        flat, off = pack_messages([bytes(m) for m in msgs])
        if rlc:
            return e.verify_sig_rlc(_u8(self.raw), _stack([s.raw for s in sigs], 192), flat, off, seed=seed)[0].astype(bool)
        return e.verify_sig(_u8(self.raw), _stack([s.raw for s in sigs], 192), flat, off).astype(bool)


class PublicKeyShare(PublicKey):
    """struct PublicKeyShare(PublicKey) (src/lib.rs:159)."""

    def verify_decryption_share(self, share, ct):
        """PublicKeyShare::verify_decryption_share (src/lib.rs:182-186)."""
        return bool(PublicKeyShare.verify_decryption_share_batch([self], [share], [ct])[0])

    @staticmethod
    def verify_decryption_share_batch(pk_shares, shares, cts, engine=None):
        e = engine or default_engine()
        if not (len(pk_shares) == len(shares) == len(cts)):
            raise ValueError("one key share, one decryption share and one ciphertext per job")
        v, off = pack_messages([c.v for c in cts])
        return e.verify_decryption_share(_stack([p.raw for p in pk_shares], 96), _stack([s.raw for s in shares], 96),
                                         _stack([c.u for c in cts], 96), v, off,
                                         _stack([c.w for c in cts], 192)).astype(bool)

    @staticmethod
    def verify_decryption_shares_rlc(pk_shares, shares, cts, engine=None, seed=None):
        """The loop of examples/threshold_enc.rs over verify_decryption_share for B ciphertexts x N nodes by ONE random
        linear combination per ciphertext (opt-in, tc_verify_decryption_shares_rlc_batch): pk_shares: the N key shares,
        shares[j][i]: node i's decryption share of cts[j]; returns a (B, N) boolean array."""
        e = engine or default_engine()
        N = len(pk_shares)
        if len(shares) != len(cts) or any(len(row) != N for row in shares):
            raise ValueError("one row of N decryption shares per ciphertext")
        v, off = pack_messages([c.v for c in cts])
        sh = np.stack([_stack([s.raw for s in row], 96) for row in shares])
        ok, _ = e.verify_decryption_shares_rlc(_stack([p.raw for p in pk_shares], 96), sh, _stack([c.u for c in cts], 96), v, off,
                                               _stack([c.w for c in cts], 192), seed=seed)
        return ok.astype(bool)

    @staticmethod
    def verify_batch_shares(pk_shares, sig_shares, msgs, engine=None):
        """PublicKeyShare::verify (src/lib.rs:177-179) for many (share key, share, msg) triples."""
        e = engine or default_engine()
        if not (len(pk_shares) == len(sig_shares) == len(msgs)):
            raise ValueError("one key share, one signature share and one message per job")
        flat, off = pack_messages([bytes(m) for m in msgs])
        return e.verify_sig(_stack([p.raw for p in pk_shares], 96), _stack([s.raw for s in sig_shares], 192), flat,
                            off).astype(bool)


def hash_g2(msg, engine=None):
    """pub fn hash_g2 (src/lib.rs:691-694): 192-byte uncompressed G2 point."""
    return hash_g2_batch([msg], engine)[0]


def hash_g2_batch(msgs, engine=None):
    e = engine or default_engine()
    flat, off = pack_messages([bytes(m) for m in msgs])
    return [bytes(x) for x in e.hash_g2(flat, off)]


class SecretKey:
    """struct SecretKey(Box<Fr>) (src/lib.rs:302)."""

    def __init__(self, fr):
        self.fr = int(fr) % _R

    def _bytes(self):
        return self.fr.to_bytes(32, "little")

    def public_key(self):
        """SecretKey::public_key (src/lib.rs:367-369)."""
        out, st = default_engine().g1_mul(_u8(self._bytes())[None], _u8(_G1_GEN)[None])
        _raise_status(st[0, 0])
        return PublicKey(out[0, 0], _trusted=True)

    def sign_g2(self, hash_g2_point):
        """SecretKey::sign_g2 (src/lib.rs:372-374)."""
        out, st = default_engine().g2_mul(_u8(self._bytes())[None], _u8(hash_g2_point)[None])
        _raise_status(st[0, 0])
        return Signature(out[0, 0], _trusted=True)

    def sign(self, msg):
        """SecretKey::sign (src/lib.rs:379-381)."""
        return self.sign_batch([msg])[0]

    def sign_batch(self, msgs, engine=None, cls=Signature):
        e = engine or default_engine()
        flat, off = pack_messages([bytes(m) for m in msgs])
        out, st = e.sign(_u8(self._bytes())[None], flat, off)
        for s_ in st.reshape(-1):
            _raise_status(s_)
        return [cls(out[j, 0], _trusted=True) for j in range(len(msgs))]

    def decrypt(self, ct):
        """SecretKey::decrypt (src/lib.rs:384-391): None if the ciphertext is invalid."""
        return self.decrypt_batch([ct])[0]

    def decrypt_batch(self, cts, engine=None):
        """SecretKey::decrypt for B ciphertexts in ONE call (verify + [sk] u + xor_with_hash on the device): the plaintext,
        or None where Ciphertext::verify fails."""
        e = engine or default_engine()
        v, off = pack_messages([ct.v for ct in cts])
        out, ok = e.secret_key_decrypt(_u8(self._bytes()), _stack([ct.u for ct in cts], 96), v, off, _stack([ct.w for ct in cts], 192))
        return [bytes(out[int(off[j]):int(off[j + 1])]) if ok[j] else None for j in range(len(cts))]


class SecretKeyShare(SecretKey):
    """struct SecretKeyShare(SecretKey) (src/lib.rs:408)."""

    def public_key_share(self):
        """SecretKeyShare::public_key_share (src/lib.rs:437-439)."""
        return PublicKeyShare(self.public_key().raw, _trusted=True)

    def sign_g2(self, hash_g2_point):
        """SecretKeyShare::sign_g2 (src/lib.rs:442-444)."""
        return SignatureShare(SecretKey.sign_g2(self, hash_g2_point).raw, _trusted=True)

    def sign(self, msg):
        """SecretKeyShare::sign (src/lib.rs:447-449)."""
        return self.sign_batch([msg], cls=SignatureShare)[0]

    def decrypt_share(self, ct):
        """SecretKeyShare::decrypt_share (src/lib.rs:452-457)."""
        return self.decrypt_share_batch([ct])[0]

    def decrypt_share_batch(self, cts, engine=None):
        """decrypt_share for B ciphertexts in ONE call (verify + [sk] u on the device): None where Ciphertext::verify fails."""
        e = engine or default_engine()
        v, off = pack_messages([ct.v for ct in cts])
        out, ok = e.decrypt_share(_u8(self._bytes()), _stack([ct.u for ct in cts], 96), v, off, _stack([ct.w for ct in cts], 192))
        return [DecryptionShare(out[j], _trusted=True) if ok[j] else None for j in range(len(cts))]

    def decrypt_share_no_verify(self, ct):
        """SecretKeyShare::decrypt_share_no_verify (src/lib.rs:460-462)."""
        out, st = default_engine().g1_mul(_u8(self._bytes())[None], _u8(ct.u)[None])
        _raise_status(st[0, 0])
        return DecryptionShare(out[0, 0], _trusted=True)


def sign_shares_batch(secret_key_shares, msgs, engine=None):
    """S signers x B messages in one launch: result[j][s] = shares[s].sign(msgs[j])."""
    e = engine or default_engine()
    flat, off = pack_messages([bytes(m) for m in msgs])
    fr = _stack([s._bytes() for s in secret_key_shares], 32)
    out, st = e.sign(fr, flat, off)
    for s_ in st.reshape(-1):
        _raise_status(s_)
    return [[SignatureShare(out[j, s], _trusted=True) for s in range(len(secret_key_shares))] for j in range(len(msgs))]


def _ordered(shares):
    """The reference iterates a BTreeMap / any IntoIterator of (index, share): dicts are taken in
    ascending index order (BTreeMap), sequences of pairs in the order given."""
    if isinstance(shares, dict):
        keys = list(shares)
        # a BTreeMap has ONE key type: plain integers order by their signed value (BTreeMap<i64, _>: negative keys first),
        # Fr keys by their canonical value (BTreeMap<Fr, _>).  Mixing the two has no counterpart in the reference.
        if any(isinstance(k, Fr) for k in keys) and not all(isinstance(k, Fr) for k in keys):
            raise TypeError("index keys of one share map must all be integers or all be Fr (a BTreeMap has one key type)")
        return sorted(shares.items(), key=lambda kv: kv[0])
    return list(shares)


class PublicKeySet:
    """struct PublicKeySet { commit: Commitment } (src/lib.rs:539-543); commit = list of G1 (96 B)."""

    def __init__(self, commit, _trusted=False):
        self.commit = [bytes(c) for c in commit]
        if any(len(c) != 96 for c in self.commit) or not self.commit:
            raise ValueError("commit: a non-empty list of 96-byte G1 points")
        if not _trusted:
            _require_members(None, False, _stack(self.commit, 96))

    def threshold(self):
        """PublicKeySet::threshold (src/lib.rs:560-562) = commit.degree()."""
        return len(self.commit) - 1

    def public_key(self):
        """PublicKeySet::public_key (src/lib.rs:565-567)."""
        return PublicKey(self.commit[0], _trusted=True)

    def public_key_share(self, i):
        """PublicKeySet::public_key_share (src/lib.rs:570-573) = Commitment::evaluate(i + 1)
        (src/poly.rs:497-508): sum_k (i+1)^k * commit[k], evaluated as one G1 linear combination."""
        return self.public_key_shares([i])[0]

    def public_key_shares(self, indices, engine=None):
        e = engine or default_engine()
        if all(0 <= int(i) < 2 ** 64 - 1 for i in indices):
            out, st = e.public_key_shares(_stack(self.commit, 96), np.array([int(i) for i in indices], dtype=np.uint64))
            for s_ in st:
                _raise_status(s_)
            return [PublicKeyShare(out[j], _trusted=True) for j in range(len(indices))]
        n = len(self.commit)
        B = len(indices)
        scal = np.zeros((B, n, 32), dtype=np.uint8)
        for j, i in enumerate(indices):
            x = into_fr_plus_1(i)
            p = 1
            for k in range(n):
                scal[j, k] = np.frombuffer(p.to_bytes(32, "little"), dtype=np.uint8)
                p = p * x % _R
        pts = np.broadcast_to(_stack(self.commit, 96)[None], (B, n, 96)).copy()
        out, st = e.lincomb_g1(scal, pts)
        for s in st:
            _raise_status(s)
        return [PublicKeyShare(out[j], _trusted=True) for j in range(B)]

    def combine_signatures(self, shares):
        """PublicKeySet::combine_signatures (src/lib.rs:608-615)."""
        sig, st = self.combine_signatures_batch([shares])
        _raise_status(st[0])
        return sig[0]

    def combine_signatures_batch(self, jobs, engine=None):
        """jobs: list of share sets (dict idx->SignatureShare or sequence of pairs), all with the
        same number of samples.  Returns ([Signature], status[])."""
        e = engine or default_engine()
        t = self.threshold()
        ordered = [_ordered(j) for j in jobs]
        n = len(ordered[0])
        if any(len(o) != n for o in ordered):
            raise ValueError("all jobs of one batch must supply the same number of shares")
        idx, idx_fr = _index_arrays(ordered, n)
        sh = np.empty((len(jobs), max(n, 1), 192), dtype=np.uint8)
        for j, o in enumerate(ordered):
            for k, (_, s) in enumerate(o):
                sh[j, k] = np.frombuffer(s.raw, dtype=np.uint8)
        sh = sh[:, :n].copy() if n else sh[:, :0].copy()
        # `T: IntoFr` (src/lib.rs:608-611): u64-range integers take the u64 entry, Fr keys / negative integers the Fr entry
        out, st = e.combine_g2(t, idx, sh) if idx_fr is None else e.combine_g2_fr(t, idx_fr, sh)
        return [Signature(out[j], _trusted=True) for j in range(len(jobs))], st

    def combine_signatures_wire_batch(self, jobs, engine=None):
        """The same on the wire forms: jobs hold (index, 96-byte SignatureShare::to_bytes) pairs; the shares pass the checked
        decode of from_bytes (src/lib.rs:246-252) on the device and the result comes back as Signature::to_bytes
        (src/lib.rs:255-259).  Returns ([bytes], status[])."""
        e = engine or default_engine()
        t = self.threshold()
        ordered = [_ordered(j) for j in jobs]
        n = len(ordered[0])
        if any(len(o) != n for o in ordered):
            raise ValueError("all jobs of one batch must supply the same number of shares")
        idx = np.array([[int(i) for i, _ in o] for o in ordered], dtype=np.uint64).reshape(len(jobs), n)
        sh = np.zeros((len(jobs), n, 96), dtype=np.uint8)
        for j, o in enumerate(ordered):
            for k, (_, s) in enumerate(o):
                sh[j, k] = np.frombuffer(bytes(s), dtype=np.uint8)
        out, st = e.combine_signatures_wire(t, idx, sh)
        return [bytes(out[j]) for j in range(len(jobs))], st

    def decrypt(self, shares, ct):
        """PublicKeySet::decrypt (src/lib.rs:618-626)."""
        out, st = self.decrypt_batch([shares], [ct])
        _raise_status(st[0])
        return out[0]

    def decrypt_batch(self, jobs, cts, engine=None):
        e = engine or default_engine()
        t = self.threshold()
        if len(jobs) != len(cts):
            raise ValueError("one ciphertext per share set")
        ordered = [_ordered(j) for j in jobs]
        n = len(ordered[0])
        if any(len(o) != n for o in ordered):
            raise ValueError("all jobs of one batch must supply the same number of shares")
        idx, idx_fr = _index_arrays(ordered, n)
        sh = np.empty((len(jobs), n, 96), dtype=np.uint8)
        for j, o in enumerate(ordered):
            for k, (_, s) in enumerate(o):
                sh[j, k] = np.frombuffer(s.raw, dtype=np.uint8)
        v, off = pack_messages([c.v for c in cts])
        out, st = e.decrypt(t, idx, sh, v, off) if idx_fr is None else e.decrypt_fr(t, idx_fr, sh, v, off)
        res = [bytes(out[int(off[j]): int(off[j + 1])]) for j in range(len(jobs))]
        return res, st


class SecretKeySet:
    """struct SecretKeySet { poly: Poly } (src/lib.rs:631-635); poly = Fr coefficients."""

    def __init__(self, poly):
        self.poly = [int(c) % _R for c in poly]

    def threshold(self):
        """SecretKeySet::threshold (src/lib.rs:664-666)."""
        return len(self.poly) - 1

    def secret_key_share(self, i):
        """SecretKeySet::secret_key_share (src/lib.rs:670-673): Poly::evaluate(i + 1), Horner in Fr
        (src/poly.rs:358-369) -- key generation, host side as in the reference."""
        x = into_fr_plus_1(i)
        res = 0
        for c in reversed(self.poly):
            res = (res * x + c) % _R
        return SecretKeyShare(res)

    def public_keys(self, engine=None):
        """SecretKeySet::public_keys (src/lib.rs:676-680) = Poly::commitment (src/poly.rs:372-377)."""
        e = engine or default_engine()
        fr = _stack([c.to_bytes(32, "little") for c in self.poly], 32)
        out, st = e.g1_mul(fr, _u8(_G1_GEN)[None])
        return PublicKeySet([bytes(out[0, k]) for k in range(len(self.poly))], _trusted=True)
