"""Host-side mirror of the reference's `poly` module (src/poly.rs) for the DKG algebra: same type and
method names, group work on the MI355X behind the C ABI.

  Poly.commitment / BivarPoly.commitment   tc_g1_commitment_batch      (fixed-base G1, LDS window table)
  Commitment.evaluate                      tc_public_key_share_batch   (Horner in G1; src/poly.rs:497-508)
  BivarCommitment.row / evaluate           tc_bivar_commitment_row_batch (+ Horner in y)
  Poly.interpolate                         tc_fr_interpolate_batch

Secret polynomials (Poly, BivarPoly) are Fr coefficient lists; their Fr-only arithmetic (evaluate, row,
add/mul) stays on the host exactly as in the reference, where it is key generation outside the hot path
(SecretKeySet does the same in api.py).  Nothing here imports the oracle; there is no CPU fallback for the
group operations.
"""
import numpy as np

from .api import _R, _stack, _raise_status, _require_members, default_engine


def into_fr(x):
    """IntoFr (src/into_fr.rs): integers map to Fr by value (negative i64 wrap), NOT plus one."""
    return int(x) % _R


def _fr_rows(vals):
    return _stack([into_fr(v).to_bytes(32, "little") for v in vals], 32)


def coeff_pos(i, j):
    """coeff_pos (src/poly.rs:746-750)."""
    if j < i:
        i, j = j, i
    return i + j * (j + 1) // 2


class Commitment:
    """struct Commitment { coeff: Vec<G1> } (src/poly.rs:432-437): 96-byte uncompressed points."""

    def __init__(self, coeff, _trusted=False):
        self.coeff = [bytes(c) for c in coeff]
        if self.coeff and not _trusted:
            _require_members(None, False, _stack(self.coeff, 96))

    def __eq__(self, other):
        return isinstance(other, Commitment) and self._trimmed() == other._trimmed()

    def _trimmed(self):
        inf = bytes([0x40]) + bytes(95)
        c = list(self.coeff)
        while c and c[-1] == inf:
            c.pop()
        return c

    def degree(self):
        """Commitment::degree (src/poly.rs:492-494)."""
        return max(len(self.coeff) - 1, 0)

    def evaluate(self, i, engine=None):
        """Commitment::evaluate (src/poly.rs:497-508) at the u64 abscissa i (taken as given)."""
        return self.evaluate_batch([i], engine)[0]

    def evaluate_batch(self, xs, engine=None):
        """Commitment::evaluate for several abscissae.  `i: T: IntoFr` in the reference: any integer is taken by value
        modulo r (negative values wrap, src/into_fr.rs).  1 <= x < 2^64 runs the Horner kernel (tc_public_key_share_batch
        with idx = x - 1); x = 0 is coefficient 0; every other field element (x = 2^64 -- the share index 2^64 - 1 --
        and beyond) is the linear combination sum_k x^k commit[k] (tc_g1_lincomb_batch)."""
        e = engine or default_engine()
        if not self.coeff:
            return [bytes([0x40]) + bytes(95)] * len(xs)
        xs = [into_fr(x) for x in xs]
        out = [None] * len(xs)
        small = [k for k, x in enumerate(xs) if 1 <= x <= 2 ** 64 - 1]
        if small:
            res, st = e.public_key_shares(_stack(self.coeff, 96), np.array([xs[k] - 1 for k in small], dtype=np.uint64))
            for s in st:
                _raise_status(s)
            for k, r in zip(small, res):
                out[k] = bytes(r)
        wide = [k for k, x in enumerate(xs) if x > 2 ** 64 - 1]
        if wide:
            n = len(self.coeff)
            pts = np.ascontiguousarray(np.broadcast_to(_stack(self.coeff, 96)[None], (len(wide), n, 96)))
            sc = np.empty((len(wide), n, 32), dtype=np.uint8)
            for row, k in enumerate(wide):
                p = 1
                for d in range(n):
                    sc[row, d] = np.frombuffer(p.to_bytes(32, "little"), dtype=np.uint8)
                    p = p * xs[k] % _R
            res, st = e.lincomb_g1(sc, pts)
            for s in st:
                _raise_status(s)
            for k, r in zip(wide, res):
                out[k] = bytes(r)
        for k, x in enumerate(xs):
            if x == 0:
                out[k] = self.coeff[0]
        return out


class Poly:
    """struct Poly { coeff: Vec<Fr> } (src/poly.rs:44-49)."""

    def __init__(self, coeff):
        self.coeff = [into_fr(c) for c in coeff]
        self._remove_zeros()

    def _remove_zeros(self):
        while self.coeff and self.coeff[-1] == 0:
            self.coeff.pop()

    def __eq__(self, other):
        return isinstance(other, Poly) and self.coeff == other.coeff

    def degree(self):
        return max(len(self.coeff) - 1, 0)

    def evaluate(self, i):
        """Poly::evaluate (src/poly.rs:358-369)."""
        x, res = into_fr(i), 0
        for c in reversed(self.coeff):
            res = (res * x + c) % _R
        return res

    def __add__(self, other):
        n = max(len(self.coeff), len(other.coeff))
        g = lambda p, k: p.coeff[k] if k < len(p.coeff) else 0
        return Poly([(g(self, k) + g(other, k)) % _R for k in range(n)])

    def commitment(self, engine=None):
        """Poly::commitment (src/poly.rs:372-377)."""
        return Poly.commitment_batch([self], engine)[0]

    @staticmethod
    def commitment_batch(polys, engine=None):
        """every coefficient of every polynomial in ONE fixed-base launch"""
        e = engine or default_engine()
        flat = [c for p in polys for c in p.coeff]
        if not flat:
            return [Commitment([], _trusted=True) for _ in polys]
        out, st = e.g1_commitment(_fr_rows(flat))
        for s in st:
            _raise_status(s)
        res, pos = [], 0
        for p in polys:
            res.append(Commitment([bytes(out[pos + k]) for k in range(len(p.coeff))], _trusted=True))
            pos += len(p.coeff)
        return res

    @staticmethod
    def interpolate(samples, engine=None):
        """Poly::interpolate (src/poly.rs:341-350): samples = dict or sequence of (x, y) pairs, x and y IntoFr."""
        return Poly.interpolate_batch([samples], engine)[0]

    @staticmethod
    def interpolate_batch(jobs, engine=None):
        e = engine or default_engine()
        ordered = [sorted(j.items()) if isinstance(j, dict) else list(j) for j in jobs]
        n = len(ordered[0])
        if any(len(o) != n for o in ordered):
            raise ValueError("all jobs of one batch must hold the same number of samples")
        if n == 0:
            return [Poly([]) for _ in jobs]
        xs = np.stack([_fr_rows([x for x, _ in o]) for o in ordered])
        ys = np.stack([_fr_rows([y for _, y in o]) for o in ordered])
        out, st = e.fr_interpolate(xs, ys)
        for s in st:
            if int(s) == 2:
                raise ValueError("sample points must be distinct")   # the reference panics (src/poly.rs:404)
            _raise_status(s)
        return [Poly([int.from_bytes(bytes(out[j, k]), "little") for k in range(n)]) for j in range(len(jobs))]


class BivarPoly:
    """struct BivarPoly { degree, coeff: Vec<Fr> } (src/poly.rs:530-536): symmetric, coefficients in coeff_pos
    order."""

    def __init__(self, degree, coeff):
        self.degree_ = int(degree)
        self.coeff = [into_fr(c) for c in coeff]
        if len(self.coeff) != (self.degree_ + 1) * (self.degree_ + 2) // 2:
            raise ValueError("a symmetric polynomial of degree d has (d+1)(d+2)/2 coefficients")

    def degree(self):
        return self.degree_

    def _powers(self, x):
        out, p, x = [], 1, into_fr(x)
        for _ in range(self.degree_ + 1):
            out.append(p)
            p = p * x % _R
        return out

    def evaluate(self, x, y):
        """BivarPoly::evaluate (src/poly.rs:587-603)."""
        xp, yp = self._powers(x), self._powers(y)
        d = self.degree_
        return sum(self.coeff[coeff_pos(i, j)] * xp[i] * yp[j] for i in range(d + 1) for j in range(d + 1)) % _R

    def row(self, x):
        """BivarPoly::row (src/poly.rs:606-622)."""
        xp, d = self._powers(x), self.degree_
        return Poly([sum(self.coeff[coeff_pos(i, j)] * xp[j] for j in range(d + 1)) % _R for i in range(d + 1)])

    def commitment(self, engine=None):
        """BivarPoly::commitment (src/poly.rs:625-632)."""
        e = engine or default_engine()
        out, st = e.g1_commitment(_fr_rows(self.coeff))
        for s in st:
            _raise_status(s)
        return BivarCommitment(self.degree_, [bytes(o) for o in out], _trusted=True)


class BivarCommitment:
    """struct BivarCommitment { degree, coeff: Vec<G1> } (src/poly.rs:664-670)."""

    def __init__(self, degree, coeff, _trusted=False):
        self.degree_ = int(degree)
        self.coeff = [bytes(c) for c in coeff]
        if len(self.coeff) != (self.degree_ + 1) * (self.degree_ + 2) // 2:
            raise ValueError("a symmetric commitment of degree d has (d+1)(d+2)/2 coefficients")
        if not _trusted:
            _require_members(None, False, _stack(self.coeff, 96))

    def __eq__(self, other):
        return isinstance(other, BivarCommitment) and (self.degree_, self.coeff) == (other.degree_, other.coeff)

    def degree(self):
        return self.degree_

    def row(self, x, engine=None):
        """BivarCommitment::row (src/poly.rs:713-727)."""
        return self.row_batch([x], engine)[0]

    def row_batch(self, xs, engine=None):
        e = engine or default_engine()
        xs = [int(x) for x in xs]
        if any(x < 0 or x >= 2 ** 64 for x in xs):
            raise ValueError("rows are addressed by u64 abscissae")
        out, st = e.bivar_commitment_rows(_stack(self.coeff, 96), self.degree_, np.array(xs, dtype=np.uint64))
        for s in st.reshape(-1):
            _raise_status(s)
        return [Commitment([bytes(out[m, i]) for i in range(self.degree_ + 1)], _trusted=True) for m in range(len(xs))]

    def evaluate(self, x, y, engine=None):
        """BivarCommitment::evaluate (src/poly.rs:694-710) = row(x).evaluate(y)."""
        return self.row(x, engine).evaluate(y, engine)
