"""Oracle A -- pure-Python big-int restatement of the threshold_crypto 0.4.0 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (threshold_crypto_amd/) may
import this module; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
and there only as the checker.

PARITY STATUS: "parity unpinned" w.r.t. the Rust crate for the implementation-defined
parts (hash_g2 / hash_g1_g2 / xor_with_hash RNG consumption, "H-spec"):
the reference tests hold no BLS12-381 known-answer vectors and the Rust toolchain and the
pairing/ff/rand_chacha crates are absent from this environment.  Everything else (group
law, Lagrange combination, encodings, pairing booleans) is mathematically determined: a
unique affine point has a unique Zcash encoding.  Public anchors checked in
tests/test_oracle.py: curve constants, generator encodings, r*G = 0, bilinearity,
h2 polynomial, FIPS-202 SHA3 (hashlib), ChaCha20 djb zero-key keystream words.

Reference (cited relative to /root/reference):
  src/lib.rs:691-694   hash_g2
  src/lib.rs:697-707   hash_g1_g2
  src/lib.rs:710-715   xor_with_hash
  src/lib.rs:719-773   interpolate, into_fr_plus_1
  src/lib.rs:108-117   PublicKey::verify_g2 / verify
  src/lib.rs:182-186   PublicKeyShare::verify_decryption_share
  src/lib.rs:372-391   SecretKey::sign_g2 / sign / decrypt
  src/lib.rs:452-462   SecretKeyShare::decrypt_share(_no_verify)
  src/lib.rs:508-512   Ciphertext::verify
  src/lib.rs:128-137   PublicKey::encrypt_with_rng
  src/lib.rs:560-626   PublicKeySet::{threshold,public_key,public_key_share,combine_signatures,decrypt}
  src/poly.rs:358-377  Poly::evaluate / commitment;  src/poly.rs:497-508 Commitment::evaluate
  src/into_fr.rs       IntoFr
  src/util.rs:3-9      sha3_256
Third-party behaviour restated from the published algorithms of pairing 0.16.0 /
ff 0.6.0 / group 0.6.0 / rand_chacha 0.2.2 / tiny-keccak 2.0.1.
"""
import hashlib
import struct

# ----------------------------------------------------------------------------------------
# constants
# ----------------------------------------------------------------------------------------
Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
BLS_X = 0xd201000000010000  # |x|; x is negative
BLS_X_IS_NEGATIVE = True
H2 = 0x5d543a95414e7f1091d50792876a202cd91de4547085abaa68a205b2e5a7ddfa628f1cb4d9e82ef21537e293a6691ae1616ec6e786f0c70cf1c38e31c7238e5
H1 = 0x396c8c005555e1568c00aaab0000aaab

G1_X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
G1_Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
G2_X = (352701069587466618187139116011060144890029952792775240219908644239793785735715026873347600343865175952761926303160,
        3059144344244213709971259814753781636986470325476647558659373206291635324768958432433509563104347017837885763365758)
G2_Y = (1985150602287291935568054521177171638300868978215655730859378665066344726373823718423869104263333984641494340347905,
        927553665492332455747201965776037880757740193453592970025027978793976877002675564980949289727957565575433344219582)

PK_SIZE = 48   # src/lib.rs:71
SIG_SIZE = 96  # src/lib.rs:75

# ----------------------------------------------------------------------------------------
# Fq2 = Fq[u]/(u^2+1), elements are tuples (c0, c1)
# ----------------------------------------------------------------------------------------
F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (1, 1)  # u + 1, the Fq6 non-residue


def f2_add(a, b):
    return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)


def f2_sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def f2_neg(a):
    return ((-a[0]) % Q, (-a[1]) % Q)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def f2_sqr(a):
    return ((a[0] + a[1]) * (a[0] - a[1]) % Q, 2 * a[0] * a[1] % Q)


def f2_scale(a, k):
    return (a[0] * k % Q, a[1] * k % Q)


def f2_conj(a):
    return (a[0], (-a[1]) % Q)


def f2_inv(a):
    t = pow((a[0] * a[0] + a[1] * a[1]) % Q, Q - 2, Q)
    return (a[0] * t % Q, (-a[1] * t) % Q)


def f2_mul_xi(a):  # times (1 + u)
    return ((a[0] - a[1]) % Q, (a[0] + a[1]) % Q)


def f2_pow(a, e):
    r = F2_ONE
    for bit in bin(e)[2:]:
        r = f2_sqr(r)
        if bit == '1':
            r = f2_mul(r, a)
    return r


def f2_is_zero(a):
    return a[0] == 0 and a[1] == 0


def f2_sqrt(a):
    """Square root in Fq2 (q = 3 mod 4), Algorithm 9 of eprint 2012/685 as used by
    pairing 0.16 Fq2::sqrt.  Returns None for a non-square.  Which root is returned is
    not observable (callers re-select by lexicographic order)."""
    if f2_is_zero(a):
        return F2_ZERO
    a1 = f2_pow(a, (Q - 3) // 4)
    alpha = f2_mul(f2_sqr(a1), a)
    a0 = f2_mul(f2_conj(alpha), alpha)  # alpha^(q+1)
    if a0 == ((Q - 1), 0):
        return None
    a1 = f2_mul(a1, a)
    if alpha == ((Q - 1), 0):
        return f2_mul(a1, (0, 1))
    alpha = f2_add(alpha, F2_ONE)
    alpha = f2_pow(alpha, (Q - 1) // 2)
    return f2_mul(alpha, a1)


def f2_lex_gt(a, b):
    """pairing Fq2 Ord: compare c1 then c0 as canonical integers."""
    return (a[1], a[0]) > (b[1], b[0])


# ----------------------------------------------------------------------------------------
# Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v)
# ----------------------------------------------------------------------------------------
F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)


def f6_add(a, b):
    return (f2_add(a[0], b[0]), f2_add(a[1], b[1]), f2_add(a[2], b[2]))


def f6_sub(a, b):
    return (f2_sub(a[0], b[0]), f2_sub(a[1], b[1]), f2_sub(a[2], b[2]))


def f6_neg(a):
    return (f2_neg(a[0]), f2_neg(a[1]), f2_neg(a[2]))


def f6_mul(a, b):
    t0 = f2_mul(a[0], b[0])
    t1 = f2_mul(a[1], b[1])
    t2 = f2_mul(a[2], b[2])
    c0 = f2_add(t0, f2_mul_xi(f2_sub(f2_sub(f2_mul(f2_add(a[1], a[2]), f2_add(b[1], b[2])), t1), t2)))
    c1 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a[0], a[1]), f2_add(b[0], b[1])), t0), t1), f2_mul_xi(t2))
    c2 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a[0], a[2]), f2_add(b[0], b[2])), t0), t2), t1)
    return (c0, c1, c2)


def f6_mul_by_v(a):
    return (f2_mul_xi(a[2]), a[0], a[1])


def f6_inv(a):
    c0, c1, c2 = a
    t0 = f2_sub(f2_sqr(c0), f2_mul_xi(f2_mul(c1, c2)))
    t1 = f2_sub(f2_mul_xi(f2_sqr(c2)), f2_mul(c0, c1))
    t2 = f2_sub(f2_sqr(c1), f2_mul(c0, c2))
    d = f2_add(f2_mul(c0, t0), f2_mul_xi(f2_add(f2_mul(c2, t1), f2_mul(c1, t2))))
    di = f2_inv(d)
    return (f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di))


F12_ONE = (F6_ONE, F6_ZERO)


def f12_mul(a, b):
    t0 = f6_mul(a[0], b[0])
    t1 = f6_mul(a[1], b[1])
    c1 = f6_sub(f6_sub(f6_mul(f6_add(a[0], a[1]), f6_add(b[0], b[1])), t0), t1)
    c0 = f6_add(t0, f6_mul_by_v(t1))
    return (c0, c1)


def f12_sqr(a):
    return f12_mul(a, a)


def f12_conj(a):
    return (a[0], f6_neg(a[1]))


def f12_inv(a):
    t = f6_sub(f6_mul(a[0], a[0]), f6_mul_by_v(f6_mul(a[1], a[1])))
    ti = f6_inv(t)
    return (f6_mul(a[0], ti), f6_neg(f6_mul(a[1], ti)))


def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_sqr(r)
        if bit == '1':
            r = f12_mul(r, a)
    return r


# Frobenius constants: gamma_k[i] = xi^(i*(q^k-1)/6)
def _frob_consts(k):
    e = (Q ** k - 1) // 6
    g = f2_pow(XI, e)
    out = [F2_ONE]
    for _ in range(5):
        out.append(f2_mul(out[-1], g))
    return out


_FROB = {k: _frob_consts(k) for k in (1, 2, 3)}


def f12_frobenius(a, k):
    """a^(q^k) for k in {1,2,3}.  Basis of Fq12 over Fq2: w^j with w^2 = v, v^3 = xi;
    a = sum_{i,j} a[j][i] v^i w^j = sum a[j][i] w^(2i+j)."""
    g = _FROB[k]
    conj = (k % 2 == 1)

    def c(x, idx):
        y = f2_conj(x) if conj else x
        return f2_mul(y, g[idx])
    (a00, a01, a02), (a10, a11, a12) = a
    return ((c(a00, 0), c(a01, 2), c(a02, 4)), (c(a10, 1), c(a11, 3), c(a12, 5)))


# ----------------------------------------------------------------------------------------
# curves: affine points are None (infinity) or (x, y); generic over the coordinate field
# ----------------------------------------------------------------------------------------
class _Fq:
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % Q)
    sub = staticmethod(lambda a, b: (a - b) % Q)
    mul = staticmethod(lambda a, b: a * b % Q)
    sqr = staticmethod(lambda a: a * a % Q)
    neg = staticmethod(lambda a: (-a) % Q)
    inv = staticmethod(lambda a: pow(a, Q - 2, Q))
    is_zero = staticmethod(lambda a: a == 0)
    b = 4


class _Fq2:
    zero, one = F2_ZERO, F2_ONE
    add, sub, mul, sqr, neg, inv = map(staticmethod, (f2_add, f2_sub, f2_mul, f2_sqr, f2_neg, f2_inv))
    is_zero = staticmethod(f2_is_zero)
    b = (4, 4)


class Curve:
    """Short-Weierstrass y^2 = x^3 + b with textbook affine group law (independent of the
    Jacobian formulas used by Oracle B and by the HIP kernels)."""

    def __init__(self, F):
        self.F = F

    def on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.sqr(y) == F.add(F.mul(F.sqr(x), x), F.b)

    def neg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    def add(self, P, Pq):
        F = self.F
        if P is None:
            return Pq
        if Pq is None:
            return P
        x1, y1 = P
        x2, y2 = Pq
        if x1 == x2:
            if y1 == y2:
                return self.dbl(P)
            return None
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        x3 = F.sub(F.sub(F.sqr(lam), x1), x2)
        y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
        return (x3, y3)

    def dbl(self, P):
        F = self.F
        if P is None:
            return None
        x1, y1 = P
        if F.is_zero(y1):
            return None
        xx = F.sqr(x1)
        lam = F.mul(F.add(F.add(xx, xx), xx), F.inv(F.add(y1, y1)))
        x3 = F.sub(F.sub(F.sqr(lam), x1), x1)
        y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
        return (x3, y3)

    # Jacobian internals to keep the Python oracle fast (one inversion per scalar-mul)
    def _jdbl(self, P):
        F = self.F
        X, Y, Z = P
        if F.is_zero(Z):
            return P
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        D = F.sub(F.sub(F.sqr(F.add(X, B)), A), C)
        D = F.add(D, D)
        E = F.add(F.add(A, A), A)
        Fv = F.sqr(E)
        X3 = F.sub(Fv, F.add(D, D))
        C8 = F.add(C, C)
        C8 = F.add(C8, C8)
        C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        Z3 = F.mul(F.add(Y, Y), Z)
        return (X3, Y3, Z3)

    def _jadd_affine(self, P, A):
        F = self.F
        if A is None:
            return P
        X1, Y1, Z1 = P
        if F.is_zero(Z1):
            return (A[0], A[1], F.one)
        Z1Z1 = F.sqr(Z1)
        U2 = F.mul(A[0], Z1Z1)
        S2 = F.mul(F.mul(A[1], Z1), Z1Z1)
        if U2 == X1:
            if S2 == Y1:
                return self._jdbl(P)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, X1)
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        r = F.sub(S2, Y1)
        V = F.mul(X1, HH)
        X3 = F.sub(F.sub(F.sqr(r), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(r, F.sub(V, X3)), F.mul(Y1, HHH))
        Z3 = F.mul(Z1, H)
        return (X3, Y3, Z3)

    def _to_affine(self, P):
        F = self.F
        X, Y, Z = P
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(F.mul(Y, zi2), zi))

    def mul(self, P, k):
        """k*P, k any non-negative integer (not reduced: cofactor clearing uses k > r)."""
        F = self.F
        if P is None or k == 0:
            return None
        acc = (F.one, F.one, F.zero)
        for bit in bin(k)[2:]:
            acc = self._jdbl(acc)
            if bit == '1':
                acc = self._jadd_affine(acc, P)
        return self._to_affine(acc)


E1 = Curve(_Fq)
E2 = Curve(_Fq2)
G1_GEN = (G1_X, G1_Y)
G2_GEN = (G2_X, G2_Y)

# ----------------------------------------------------------------------------------------
# Zcash BLS12-381 encodings (pairing 0.16 EncodedPoint impls)
# ----------------------------------------------------------------------------------------
def _fq_be(x):
    return x.to_bytes(48, 'big')


def g1_uncompressed(P):
    if P is None:
        return bytes([0x40]) + bytes(95)
    return _fq_be(P[0]) + _fq_be(P[1])


def g2_uncompressed(P):
    if P is None:
        return bytes([0x40]) + bytes(191)
    (x0, x1), (y0, y1) = P
    return _fq_be(x1) + _fq_be(x0) + _fq_be(y1) + _fq_be(y0)


def g1_compressed(P):
    if P is None:
        return bytes([0xc0]) + bytes(47)
    x, y = P
    b = bytearray(_fq_be(x))
    b[0] |= 0x80
    if y > (Q - y) % Q:
        b[0] |= 0x20
    return bytes(b)


def g2_compressed(P):
    if P is None:
        return bytes([0xc0]) + bytes(95)
    (x0, x1), y = P
    b = bytearray(_fq_be(x1) + _fq_be(x0))
    b[0] |= 0x80
    if f2_lex_gt(y, f2_neg(y)):
        b[0] |= 0x20
    return bytes(b)


class DecodeError(ValueError):
    """Mirrors FromBytesError::Invalid (src/error.rs:37-41)."""


def _fq_from_be(b, strip_flags):
    b = bytearray(b)
    if strip_flags:
        b[0] &= 0x1f
    v = int.from_bytes(b, 'big')
    if v >= Q:
        raise DecodeError("coordinate not in field")
    return v


def g1_from_uncompressed(b, check=True):
    assert len(b) == 96
    if b[0] & 0x80:
        raise DecodeError("unexpected compression flag")
    if b[0] & 0x40:
        if any(b[1:]) or (b[0] & 0x3f):
            raise DecodeError("non-canonical infinity")
        return None
    if b[0] & 0x20:
        raise DecodeError("unexpected sort flag")
    P = (_fq_from_be(b[:48], True), _fq_from_be(b[48:], False))
    if check:
        if not E1.on_curve(P):
            raise DecodeError("not on curve")
        if E1.mul(P, R) is not None:
            raise DecodeError("not in subgroup")
    return P


def g2_from_uncompressed(b, check=True):
    assert len(b) == 192
    if b[0] & 0x80:
        raise DecodeError("unexpected compression flag")
    if b[0] & 0x40:
        if any(b[1:]) or (b[0] & 0x3f):
            raise DecodeError("non-canonical infinity")
        return None
    if b[0] & 0x20:
        raise DecodeError("unexpected sort flag")
    x1 = _fq_from_be(b[0:48], True)
    x0 = _fq_from_be(b[48:96], False)
    y1 = _fq_from_be(b[96:144], False)
    y0 = _fq_from_be(b[144:192], False)
    P = ((x0, x1), (y0, y1))
    if check:
        if not E2.on_curve(P):
            raise DecodeError("not on curve")
        if E2.mul(P, R) is not None:
            raise DecodeError("not in subgroup")
    return P


def g1_from_compressed(b, check=True):
    """PublicKey::from_bytes (src/lib.rs:140-146): checked decode."""
    assert len(b) == 48
    if not b[0] & 0x80:
        raise DecodeError("missing compression flag")
    if b[0] & 0x40:
        if any(b[1:]) or (b[0] & 0x3f):
            raise DecodeError("non-canonical infinity")
        return None
    greatest = bool(b[0] & 0x20)
    x = _fq_from_be(b, True)
    rhs = (x * x * x + 4) % Q
    y = pow(rhs, (Q + 1) // 4, Q)
    if y * y % Q != rhs:
        raise DecodeError("not on curve")
    if (y > Q - y) != greatest:
        y = (Q - y) % Q
    P = (x, y)
    if check and E1.mul(P, R) is not None:
        raise DecodeError("not in subgroup")
    return P


def g2_from_compressed(b, check=True):
    """Signature::from_bytes (src/lib.rs:246-252): checked decode."""
    assert len(b) == 96
    if not b[0] & 0x80:
        raise DecodeError("missing compression flag")
    if b[0] & 0x40:
        if any(b[1:]) or (b[0] & 0x3f):
            raise DecodeError("non-canonical infinity")
        return None
    greatest = bool(b[0] & 0x20)
    x1 = _fq_from_be(b[:48], True)
    x0 = _fq_from_be(b[48:], False)
    x = (x0, x1)
    rhs = f2_add(f2_mul(f2_sqr(x), x), _Fq2.b)
    y = f2_sqrt(rhs)
    if y is None:
        raise DecodeError("not on curve")
    if f2_lex_gt(y, f2_neg(y)) != greatest:
        y = f2_neg(y)
    P = (x, y)
    if check and E2.mul(P, R) is not None:
        raise DecodeError("not in subgroup")
    return P


def fr_to_bytes(s):
    """Fr wire form: 4 x u64 little-endian canonical limbs = 32 B LE (src/serde_impl.rs:296)."""
    return (s % R).to_bytes(32, 'little')


def fr_from_bytes(b):
    v = int.from_bytes(b, 'little')
    if v >= R:
        raise DecodeError("scalar not canonical")
    return v


# ----------------------------------------------------------------------------------------
# optimal ate pairing
# ----------------------------------------------------------------------------------------
def _untwist(Qp):
    """E'(Fq2) -> E(Fq12): (x', y') -> (x'/w^2, y'/w^3); w^2 = v, w^6 = xi (M-type twist)."""
    x, y = Qp
    # 1/w^2 = 1/v = v^2/xi ; 1/w^3 = w^3/xi... use explicit elements and an inversion-free form:
    # w^-2 = v^-1 = xi^-1 * v^2      -> Fq12 element with c0 = (0,0,xi^-1), c1 = 0
    # w^-3 = w^-2 * w^-1, w^-1 = w / v = xi^-1 v^2 w -> w^-3 = xi^-1 * v^-1 * ... compute numerically
    xi_inv = f2_inv(XI)
    w_m2 = ((F2_ZERO, F2_ZERO, xi_inv), F6_ZERO)          # v^2/xi
    w_m1 = (F6_ZERO, (F2_ZERO, F2_ZERO, xi_inv))          # v^2 w / xi
    w_m3 = f12_mul(w_m2, w_m1)
    X = f12_mul(((x, F2_ZERO, F2_ZERO), F6_ZERO), w_m2)
    Y = f12_mul(((y, F2_ZERO, F2_ZERO), F6_ZERO), w_m3)
    return X, Y


def _f12_from_fq(a):
    return (((a % Q, 0), F2_ZERO, F2_ZERO), F6_ZERO)


def f12_add(a, b):
    return (f6_add(a[0], b[0]), f6_add(a[1], b[1]))


def f12_sub(a, b):
    return (f6_sub(a[0], b[0]), f6_sub(a[1], b[1]))


def miller_loop_textbook(P, Qp):
    """f_{|x|,Q}(P) with affine arithmetic on E(Fq12) -- slow, obviously-correct variant
    (denominators/vertical lines omitted: they lie in a proper subfield and die in the
    final exponentiation)."""
    if P is None or Qp is None:
        return F12_ONE
    xP, yP = _f12_from_fq(P[0]), _f12_from_fq(P[1])
    xQ, yQ = _untwist(Qp)
    xT, yT = xQ, yQ
    f = F12_ONE
    three = _f12_from_fq(3)
    two = _f12_from_fq(2)
    bits = bin(BLS_X)[3:]
    for bit in bits:
        lam = f12_mul(f12_mul(three, f12_sqr(xT)), f12_inv(f12_mul(two, yT)))
        line = f12_sub(f12_sub(yP, yT), f12_mul(lam, f12_sub(xP, xT)))
        f = f12_mul(f12_sqr(f), line)
        x3 = f12_sub(f12_sub(f12_sqr(lam), xT), xT)
        y3 = f12_sub(f12_mul(lam, f12_sub(xT, x3)), yT)
        xT, yT = x3, y3
        if bit == '1':
            lam = f12_mul(f12_sub(yQ, yT), f12_inv(f12_sub(xQ, xT)))
            line = f12_sub(f12_sub(yP, yT), f12_mul(lam, f12_sub(xP, xT)))
            f = f12_mul(f, line)
            x3 = f12_sub(f12_sub(f12_sqr(lam), xT), xQ)
            y3 = f12_sub(f12_mul(lam, f12_sub(xT, x3)), yT)
            xT, yT = x3, y3
    if BLS_X_IS_NEGATIVE:
        f = f12_conj(f)
    return f


# -- fast variant: pairing 0.16 style (G2Prepared line coefficients + mul_by_014) ----------
def _doubling_step(r):
    rx, ry, rz = r
    tmp0 = f2_sqr(rx)
    tmp1 = f2_sqr(ry)
    tmp2 = f2_sqr(tmp1)
    tmp3 = f2_sub(f2_sub(f2_sqr(f2_add(tmp1, rx)), tmp0), tmp2)
    tmp3 = f2_add(tmp3, tmp3)
    tmp4 = f2_add(f2_add(tmp0, tmp0), tmp0)
    tmp6 = f2_add(rx, tmp4)
    tmp5 = f2_sqr(tmp4)
    zsq = f2_sqr(rz)
    nx = f2_sub(f2_sub(tmp5, tmp3), tmp3)
    nz = f2_sub(f2_sub(f2_sqr(f2_add(rz, ry)), tmp1), zsq)
    ny = f2_mul(f2_sub(tmp3, nx), tmp4)
    t2 = f2_scale(tmp2, 8)
    ny = f2_sub(ny, t2)
    c1 = f2_neg(f2_scale(f2_mul(tmp4, zsq), 2))
    c2 = f2_sub(f2_sub(f2_sqr(tmp6), tmp0), tmp5)
    c2 = f2_sub(c2, f2_scale(tmp1, 4))
    c0 = f2_scale(f2_mul(nz, zsq), 2)
    return (nx, ny, nz), (c0, c1, c2)


def _addition_step(r, q):
    rx, ry, rz = r
    qx, qy = q
    zsq = f2_sqr(rz)
    ysq = f2_sqr(qy)
    t0 = f2_mul(zsq, qx)
    t1 = f2_mul(f2_sub(f2_sub(f2_sqr(f2_add(qy, rz)), ysq), zsq), zsq)
    t2 = f2_sub(t0, rx)
    t3 = f2_sqr(t2)
    t4 = f2_scale(t3, 4)
    t5 = f2_mul(t4, t2)
    t6 = f2_sub(f2_sub(t1, ry), ry)
    t9 = f2_mul(t6, qx)
    t7 = f2_mul(t4, rx)
    nx = f2_sub(f2_sub(f2_sub(f2_sqr(t6), t5), t7), t7)
    nz = f2_sub(f2_sub(f2_sqr(f2_add(rz, t2)), zsq), t3)
    t10 = f2_add(qy, nz)
    t8 = f2_mul(f2_sub(t7, nx), t6)
    t0 = f2_scale(f2_mul(ry, t5), 2)
    ny = f2_sub(t8, t0)
    t10 = f2_sub(f2_sub(f2_sqr(t10), ysq), f2_sqr(nz))
    t9 = f2_sub(f2_scale(t9, 2), t10)
    t10 = f2_scale(nz, 2)
    t6 = f2_neg(t6)
    t1 = f2_scale(t6, 2)
    return (nx, ny, nz), (t10, t1, t9)


def g2_prepare(Qp):
    """G2Prepared::from_affine: 68 line-coefficient triples."""
    if Qp is None:
        return None
    coeffs = []
    r = (Qp[0], Qp[1], F2_ONE)
    for bit in bin(BLS_X >> 1)[3:]:
        r, c = _doubling_step(r)
        coeffs.append(c)
        if bit == '1':
            r, c = _addition_step(r, Qp)
            coeffs.append(c)
    r, c = _doubling_step(r)
    coeffs.append(c)
    assert len(coeffs) == 68
    return coeffs


def f12_mul_by_014(f, c0, c1, c4):
    """Sparse multiplication by (c0 + c1 v) + (c4 v) w."""
    g = ((c0, c1, F2_ZERO), (F2_ZERO, c4, F2_ZERO))
    return f12_mul(f, g)


def _ell(f, coeffs, P):
    c0, c1, c2 = coeffs
    c0 = f2_scale(c0, P[1])
    c1 = f2_scale(c1, P[0])
    return f12_mul_by_014(f, c2, c1, c0)


def miller_loop(pairs):
    """pairing 0.16 Bls12::miller_loop over [(G1Affine, G2Affine)]; infinity pairs skipped."""
    prep = [(P, g2_prepare(Qp)) for (P, Qp) in pairs if P is not None and Qp is not None]
    f = F12_ONE
    idx = 0
    for bit in bin(BLS_X >> 1)[3:]:
        for (P, co) in prep:
            f = _ell(f, co[idx], P)
        idx += 1
        if bit == '1':
            for (P, co) in prep:
                f = _ell(f, co[idx], P)
            idx += 1
        f = f12_sqr(f)
    for (P, co) in prep:
        f = _ell(f, co[idx], P)
    if BLS_X_IS_NEGATIVE:
        f = f12_conj(f)
    return f


def final_exponentiation(f):
    """f^((q^12-1)/r): easy part then the defining hard-part exponent (plain pow)."""
    f1 = f12_conj(f)
    f2 = f12_inv(f)
    r = f12_mul(f1, f2)              # f^(q^6-1)
    r = f12_mul(f12_frobenius(r, 2), r)  # ^(q^2+1)
    return f12_pow(r, (Q ** 4 - Q ** 2 + 1) // R)


def final_exponentiation_chain(f):
    """pairing 0.16 hard-part addition chain (exp_by_x with generic squarings)."""
    def exp_by_x(a, x):
        a = f12_pow(a, x)
        return f12_conj(a) if BLS_X_IS_NEGATIVE else a
    f1 = f12_conj(f)
    f2 = f12_inv(f)
    r = f12_mul(f1, f2)
    f2 = r
    r = f12_mul(f12_frobenius(r, 2), f2)
    x = BLS_X
    y0 = f12_sqr(r)
    y1 = exp_by_x(y0, x)
    x >>= 1
    y2 = exp_by_x(y1, x)
    x <<= 1
    y3 = f12_conj(r)
    y1 = f12_mul(y1, y3)
    y1 = f12_conj(y1)
    y1 = f12_mul(y1, y2)
    y2 = exp_by_x(y1, x)
    y3 = exp_by_x(y2, x)
    y1 = f12_conj(y1)
    y3 = f12_mul(y3, y1)
    y1 = f12_conj(y1)
    y1 = f12_frobenius(y1, 3)
    y2 = f12_frobenius(y2, 2)
    y1 = f12_mul(y1, y2)
    y2 = exp_by_x(y3, x)
    y2 = f12_mul(y2, y0)
    y2 = f12_mul(y2, r)
    y1 = f12_mul(y1, y2)
    y2 = f12_frobenius(y3, 1)
    y1 = f12_mul(y1, y2)
    return y1


def pairing(P, Qp):
    """Bls12::pairing(p, q) (the only pairing entry the reference calls: src/lib.rs:109,185,511)."""
    return final_exponentiation_chain(miller_loop([(P, Qp)]))


def pairing_check(a, b, c, d):
    """e(a,b) == e(c,d), evaluated as the reference does: two pairings + Fq12 compare."""
    return pairing(a, b) == pairing(c, d)


# ----------------------------------------------------------------------------------------
# SHA3-256, ChaCha20 RNG (rand_chacha 0.2.2 ChaChaRng == ChaCha20, djb layout), H-spec
# ----------------------------------------------------------------------------------------
def sha3_256(data):
    """src/util.rs:3-9 (tiny-keccak Sha3::v256 == FIPS-202)."""
    return hashlib.sha3_256(bytes(data)).digest()


def _rotl32(v, n):
    return ((v << n) & 0xffffffff) | (v >> (32 - n))


def chacha20_block(key_words, counter, stream=0):
    s = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(key_words) + \
        [counter & 0xffffffff, (counter >> 32) & 0xffffffff, stream & 0xffffffff, (stream >> 32) & 0xffffffff]
    w = list(s)

    def qr(a, b, c, d):
        w[a] = (w[a] + w[b]) & 0xffffffff; w[d] = _rotl32(w[d] ^ w[a], 16)
        w[c] = (w[c] + w[d]) & 0xffffffff; w[b] = _rotl32(w[b] ^ w[c], 12)
        w[a] = (w[a] + w[b]) & 0xffffffff; w[d] = _rotl32(w[d] ^ w[a], 8)
        w[c] = (w[c] + w[d]) & 0xffffffff; w[b] = _rotl32(w[b] ^ w[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(w[i] + s[i]) & 0xffffffff for i in range(16)]


# H-spec alternatives (the implementation-defined sampling order behind hash_g2 / xor_with_hash / key draws that
# no vector of the real crate pins yet).  HSPEC = 0 is the recalled behaviour of rand_chacha 0.2 / ff_derive 0.6 /
# pairing 0.16; each bit switches ONE item to its documented alternative -- the same bits as g_hspec in oracle/c/tc_oracle.c
# (or_set_hspec) and TC_HSPEC in threshold_crypto_amd/csrc/tc_hash.h.  tests/ref_fixtures.py diagnose() names the setting
# that reproduces a set of reference vectors.
HSPEC_U64_HI_FIRST = 1        # next_u64 = high word then low word
HSPEC_COMPARE_THEN_MASK = 2   # Fq/Fr::random accept iff the UNMASKED draw is below the modulus
HSPEC_GREATEST_MSB = 4        # greatest = top bit of next_u32 (rand's bool sampling) instead of next_u32 % 2
HSPEC_KEYSTREAM_BYTES = 8     # xor_with_hash uses consecutive keystream bytes (fill_bytes), not one word per byte
HSPEC_CANONICAL_DRAW = 16     # the accepted pattern is the canonical value, not the Montgomery representation
HSPEC_NAMES = {1: "next_u64 takes the HIGH word first", 2: "Fq::random compares BEFORE masking the top limb",
               4: "G2::random takes `greatest` from the TOP bit of next_u32", 8: "xor_with_hash uses consecutive keystream BYTES",
               16: "Fq::random's accepted pattern is the CANONICAL value (not Montgomery)"}
HSPEC = 0


def set_hspec(v):
    global HSPEC
    HSPEC = int(v)


def describe_hspec(v):
    return "the recalled H-spec (0)" if not v else "; ".join(n for b, n in HSPEC_NAMES.items() if v & b) + " (HSPEC = %d)" % v


class ChaChaRng:
    """rand_chacha 0.2.2 ChaChaRng::from_seed(seed): ChaCha20, key = seed, 64-bit block
    counter 0, 64-bit stream 0; output = keystream as sequential LE u32 words
    (BlockRng; next_u64 = low word then high word).  H-spec item 2."""

    def __init__(self, seed):
        assert len(seed) == 32
        self.key = struct.unpack('<8I', bytes(seed))
        self.counter = 0
        self.buf = []
        self.words_used = 0

    def next_u32(self):
        if not self.buf:
            self.buf = chacha20_block(self.key, self.counter)
            self.counter += 1
        self.words_used += 1
        return self.buf.pop(0)

    def next_u64(self):
        a = self.next_u32()
        b = self.next_u32()
        return (b | (a << 32)) if HSPEC & HSPEC_U64_HI_FIRST else (a | (b << 32))


FQ_R = (1 << 384) % Q
FQ_RINV = pow(FQ_R, Q - 2, Q)
FR_R = (1 << 256) % R
FR_RINV = pow(FR_R, R - 2, R)


def fq_random(rng):
    """ff_derive 0.6 PrimeField::random for Fq: 6 x next_u64 (limb 0 first), top limb masked to
    61 bits, accept if < q; the accepted pattern IS the Montgomery representation.  H-spec item 4."""
    while True:
        limbs = [rng.next_u64() for _ in range(6)]
        if not HSPEC & HSPEC_COMPARE_THEN_MASK:
            limbs[5] &= 0xffffffffffffffff >> 3
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < Q:
            return v if HSPEC & HSPEC_CANONICAL_DRAW else v * FQ_RINV % Q


def fr_random(rng):
    """ff_derive 0.6 random for Fr: 4 x next_u64, top limb masked to 63 bits.  H-spec item 5."""
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        if not HSPEC & HSPEC_COMPARE_THEN_MASK:
            limbs[3] &= 0xffffffffffffffff >> 1
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < R:
            return v if HSPEC & HSPEC_CANONICAL_DRAW else v * FR_RINV % R


def g2_get_point_from_x(x, greatest):
    rhs = f2_add(f2_mul(f2_sqr(x), x), _Fq2.b)
    y = f2_sqrt(rhs)
    if y is None:
        return None
    negy = f2_neg(y)
    y_lt_negy = f2_lex_gt(negy, y)
    return (x, y if (y_lt_negy ^ greatest) else negy)


def g2_random(rng, stats=None):
    """pairing 0.16 G2::random.  H-spec item 3."""
    attempts = 0
    while True:
        attempts += 1
        c0 = fq_random(rng)
        c1 = fq_random(rng)
        gw = rng.next_u32()
        greatest = bool(gw >> 31) if HSPEC & HSPEC_GREATEST_MSB else (gw % 2) != 0
        P = g2_get_point_from_x((c0, c1), greatest)
        if P is not None:
            P = E2.mul(P, H2)
            if P is not None:
                if stats is not None:
                    stats['attempts'] = attempts
                return P


def hash_g2(msg, stats=None):
    """src/lib.rs:691-694."""
    return g2_random(ChaChaRng(sha3_256(msg)), stats)


def hash_g1_g2(g1, msg):
    """src/lib.rs:697-707."""
    msg = bytes(msg)
    m = sha3_256(msg) if len(msg) > 64 else msg
    return hash_g2(m + g1_compressed(g1))


def xor_with_hash(g1, data):
    """src/lib.rs:710-715: byte i ^= (u8) of the i-th next_u32()."""
    rng = ChaChaRng(sha3_256(g1_compressed(g1)))
    if HSPEC & HSPEC_KEYSTREAM_BYTES:
        out, w = bytearray(), 0
        for i, b in enumerate(data):
            if i % 4 == 0:
                w = rng.next_u32()
            out.append(((w >> (8 * (i % 4))) & 0xff) ^ b)
        return bytes(out)
    return bytes((rng.next_u32() & 0xff) ^ b for b in data)


# ----------------------------------------------------------------------------------------
# threshold algebra
# ----------------------------------------------------------------------------------------
class NotEnoughShares(Exception):
    """Error::NotEnoughShares (src/error.rs:9-10)."""


class DuplicateEntry(Exception):
    """Error::DuplicateEntry (src/error.rs:12-13)."""


def into_fr_plus_1(i):
    """src/lib.rs:769-773 with IntoFr for u64/usize (src/into_fr.rs:16-26)."""
    return (int(i) + 1) % R


def lagrange_coeffs(t, xs):
    """The Fr part of src/lib.rs:739-763 for the first t+1 sample abscissae xs (already +1)."""
    n = len(xs)
    assert n == t + 1
    x_prod = [1]
    tmp = 1
    for x in xs[:t]:
        tmp = tmp * x % R
        x_prod.append(tmp)
    tmp = 1
    for i in range(t - 1, -1, -1):
        tmp = tmp * xs[i + 1] % R
        x_prod[i] = x_prod[i] * tmp % R
    out = []
    for l0, x in zip(x_prod, xs):
        denom = 1
        for x0 in xs:
            if x0 != x:
                denom = denom * ((x0 - x) % R) % R
        if denom == 0:
            raise DuplicateEntry()
        out.append(l0 * pow(denom, R - 2, R) % R)
    return out


def interpolate(curve, t, items):
    """src/lib.rs:719-767.  items: iterable of (index, affine point); takes the first t+1."""
    samples = []
    for (i, s) in items:
        if len(samples) == t + 1:
            break
        samples.append((into_fr_plus_1(i), s))
    if len(samples) <= t:
        raise NotEnoughShares()
    if t == 0:
        return samples[0][1]
    lam = lagrange_coeffs(t, [x for x, _ in samples])
    result = None
    for l0, (_, s) in zip(lam, samples):
        result = curve.add(result, curve.mul(s, l0))
    return result


def poly_evaluate(coeffs, x):
    """Poly::evaluate (src/poly.rs:358-369), Horner in Fr."""
    res = 0
    for c in reversed(coeffs):
        res = (res * x + c) % R
    return res


def commitment_evaluate(commit, x):
    """Commitment::evaluate (src/poly.rs:497-508), Horner in G1."""
    if not commit:
        return None
    res = commit[-1]
    for c in reversed(commit[:-1]):
        res = E1.add(E1.mul(res, x % R), c)
    return res


# ---- reference API restated (single-item, as the Rust methods) ---------------------------
def secret_key_share(poly, i):
    """SecretKeySet::secret_key_share (src/lib.rs:670-673)."""
    return poly_evaluate(poly, into_fr_plus_1(i))


def public_key(sk):
    """SecretKey::public_key (src/lib.rs:367-369)."""
    return E1.mul(G1_GEN, sk % R)


def commitment(poly):
    """Poly::commitment (src/poly.rs:372-377)."""
    return [E1.mul(G1_GEN, c % R) for c in poly]


def public_key_share(commit, i):
    """PublicKeySet::public_key_share (src/lib.rs:570-573)."""
    return commitment_evaluate(commit, into_fr_plus_1(i))


def sign_g2(sk, h):
    """SecretKey::sign_g2 (src/lib.rs:372-374)."""
    return E2.mul(h, sk % R)


def sign(sk, msg):
    """SecretKey::sign (src/lib.rs:379-381)."""
    return sign_g2(sk, hash_g2(msg))


def verify_g2(pk, sig, h):
    """PublicKey::verify_g2 (src/lib.rs:108-110)."""
    return pairing(pk, h) == pairing(G1_GEN, sig)


def verify(pk, sig, msg):
    """PublicKey::verify (src/lib.rs:115-117)."""
    return verify_g2(pk, sig, hash_g2(msg))


def combine_signatures(t, shares):
    """PublicKeySet::combine_signatures (src/lib.rs:608-615); shares: iterable (idx, G2 affine)."""
    return interpolate(E2, t, shares)


def encrypt_with_r(pk, r, msg):
    """PublicKey::encrypt_with_rng (src/lib.rs:128-137) with the Fr draw r supplied."""
    u = E1.mul(G1_GEN, r)
    v = xor_with_hash(E1.mul(pk, r), msg)
    w = E2.mul(hash_g1_g2(u, v), r)
    return (u, v, w)


def ciphertext_verify(ct):
    """Ciphertext::verify (src/lib.rs:508-512)."""
    u, v, w = ct
    return pairing(G1_GEN, w) == pairing(u, hash_g1_g2(u, v))


def decrypt_share_no_verify(sk, ct):
    """SecretKeyShare::decrypt_share_no_verify (src/lib.rs:460-462)."""
    return E1.mul(ct[0], sk % R)


def decrypt_share(sk, ct):
    """SecretKeyShare::decrypt_share (src/lib.rs:452-457); None if ct invalid."""
    if not ciphertext_verify(ct):
        return None
    return decrypt_share_no_verify(sk, ct)


def verify_decryption_share(pk_share, share, ct):
    """PublicKeyShare::verify_decryption_share (src/lib.rs:182-186)."""
    u, v, w = ct
    return pairing(share, hash_g1_g2(u, v)) == pairing(pk_share, w)


def decrypt(sk, ct):
    """SecretKey::decrypt (src/lib.rs:384-391)."""
    if not ciphertext_verify(ct):
        return None
    return xor_with_hash(E1.mul(ct[0], sk % R), ct[1])


def threshold_decrypt(t, shares, ct):
    """PublicKeySet::decrypt (src/lib.rs:618-626)."""
    g = interpolate(E1, t, shares)
    return xor_with_hash(g, ct[1])


def signature_parity(sig):
    """Signature::parity (src/lib.rs:237-243)."""
    x = 0
    for b in g2_uncompressed(sig):
        x ^= b
    return bin(x).count('1') % 2 != 0


# ---- DKG algebra (src/poly.rs), SURVEY.md 8f rank 4 ------------------------------------------------
def coeff_pos(i, j):
    """coeff_pos (src/poly.rs:746-750): position of coefficient (i, j) of a SYMMETRIC bivariate polynomial."""
    if j < i:
        i, j = j, i
    return i + j * (j + 1) // 2


def powers(x, degree):
    """powers (src/poly.rs:734-743): x^0 .. x^degree in Fr."""
    out, p = [], 1
    for _ in range(degree + 1):
        out.append(p)
        p = p * x % R
    return out


def bivar_poly_row(degree, coeff, x):
    """BivarPoly::row (src/poly.rs:606-622): the univariate polynomial f(x, .) as Fr coefficients."""
    xp = powers(x % R, degree)
    return [sum(coeff[coeff_pos(i, j)] * xp[j] for j in range(degree + 1)) % R for i in range(degree + 1)]


def bivar_poly_evaluate(degree, coeff, x, y):
    """BivarPoly::evaluate (src/poly.rs:587-603)."""
    xp, yp = powers(x % R, degree), powers(y % R, degree)
    return sum(coeff[coeff_pos(i, j)] * xp[i] * yp[j] for i in range(degree + 1) for j in range(degree + 1)) % R


def bivar_commitment(coeff):
    """BivarPoly::commitment (src/poly.rs:625-632): every coefficient times the G1 generator."""
    return [E1.mul(G1_GEN, c % R) for c in coeff]


def bivar_commitment_row(degree, commit, x):
    """BivarCommitment::row (src/poly.rs:713-727): row[i] = sum_j commit[pos(i, j)] * x^j."""
    xp = powers(x % R, degree)
    row = []
    for i in range(degree + 1):
        acc = None
        for j in range(degree + 1):
            acc = E1.add(acc, E1.mul(commit[coeff_pos(i, j)], xp[j]))
        row.append(acc)
    return row


def bivar_commitment_evaluate(degree, commit, x, y):
    """BivarCommitment::evaluate (src/poly.rs:694-710)."""
    xp, yp = powers(x % R, degree), powers(y % R, degree)
    acc = None
    for i in range(degree + 1):
        for j in range(degree + 1):
            acc = E1.add(acc, E1.mul(commit[coeff_pos(i, j)], xp[i] * yp[j] % R))
    return acc


def poly_interpolate(samples):
    """Poly::interpolate / compute_interpolation (src/poly.rs:341-350, 388-417): the unique polynomial of
    degree len(samples) - 1 through the (x, y) pairs (x taken as given, NOT + 1), built sample by sample
    exactly as the reference does.  Raises ZeroDivisionError where the reference panics ("sample points must
    be distinct")."""
    if not samples:
        return []
    samples = [(x % R, y % R) for x, y in samples]
    poly = [samples[0][1]]
    base = [(-samples[0][0]) % R, 1]
    for x, y in samples[1:]:
        diff = (y - poly_evaluate(poly, x)) % R
        base_val = poly_evaluate(base, x)
        if base_val == 0:
            raise ZeroDivisionError("sample points must be distinct")
        diff = diff * pow(base_val, R - 2, R) % R
        scaled = [c * diff % R for c in base]
        poly = [((poly[k] if k < len(poly) else 0) + scaled[k]) % R for k in range(len(scaled))]
        nb = [0] * (len(base) + 1)            # base *= (X - x)
        for k, c in enumerate(base):
            nb[k] = (nb[k] - c * x) % R
            nb[k + 1] = (nb[k + 1] + c) % R
        base = nb
    return poly
