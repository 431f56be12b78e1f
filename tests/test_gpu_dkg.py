"""GPU parity of the DKG algebra: Poly::commitment, BivarPoly::commitment,
BivarCommitment::row / evaluate, Poly::interpolate (src/poly.rs) vs Oracle A, and a replay of the reference's
`distributed_key_generation` test (src/poly.rs:818-900) through the API mirror (threshold_crypto_amd/poly.py)."""
import random

import numpy as np
import pytest

import tc_oracle as o
from threshold_crypto_amd import api
from threshold_crypto_amd.poly import BivarCommitment, BivarPoly, Commitment, Poly

pytestmark = pytest.mark.gpu


def u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


@pytest.fixture(scope="module")
def rnd():
    return random.Random(0xD1C6)


def test_fixed_base_commitment_matches_oracle(engine, rnd):
    """tc_g1_commitment_batch over more scalars than one grid of persistent workgroups holds lanes
    (edge digits of the signed-window recoding included); a sample is checked against the oracle, all of
    them against the variable-base GLV kernel (tc_g1_mul_batch)."""
    ks = [0, 1, 7, 8, 9, 15, 16, 0x88888888, o.R - 1, o.R - 2, (1 << 254) + 5, int("8" * 63, 16) % o.R, int("f" * 63, 16) % o.R]
    ks += [rnd.randrange(o.R) for _ in range(3000 - len(ks))]
    fr = np.stack([u8(o.fr_to_bytes(k)) for k in ks])
    out, st = engine.g1_commitment(fr)
    assert not st.any()
    for i in list(range(13)) + rnd.sample(range(13, len(ks)), 12):
        assert bytes(out[i]) == o.g1_uncompressed(o.E1.mul(o.G1_GEN, ks[i])), hex(ks[i])
    ref, st2 = engine.g1_mul(fr, u8(o.g1_uncompressed(o.G1_GEN))[None])
    assert not st2.any() and (ref[0] == out).all()
    bad = fr[:3].copy()
    bad[1] = u8(o.R.to_bytes(32, "little"))
    out, st = engine.g1_commitment(bad)
    assert st.tolist() == [0, 3, 0] and bytes(out[1]) == o.g1_uncompressed(None)


def test_bivar_rows_and_interpolation_match_oracle(engine, rnd):
    d = 3
    coeff = [rnd.randrange(o.R) for _ in range((d + 1) * (d + 2) // 2)]
    commit = o.bivar_commitment(coeff)
    blob = np.stack([u8(o.g1_uncompressed(c)) for c in commit])
    xs = [0, 1, 2, 7, 200, 2 ** 64 - 1]
    rows, st = engine.bivar_commitment_rows(blob, d, np.array(xs, dtype=np.uint64))
    assert not st.any()
    for m, x in enumerate(xs):
        want = o.bivar_commitment_row(d, commit, x)
        assert [bytes(r) for r in rows[m]] == [o.g1_uncompressed(w) for w in want], x
    # Poly::interpolate: B jobs x n samples, one with a repeated abscissa
    n, B = 5, 70
    xs_, ys_, want = [], [], []
    for j in range(B):
        f = [rnd.randrange(o.R) for _ in range(n)]
        x = rnd.sample(range(0, 300), n)
        xs_.append(x)
        ys_.append([o.poly_evaluate(f, v) for v in x])
        want.append(f)
    xs_[9][3] = xs_[9][0]
    enc = lambda rows_: np.stack([np.stack([u8(o.fr_to_bytes(v)) for v in r]) for r in rows_])
    out, st = engine.fr_interpolate(enc(xs_), enc(ys_))
    assert st.tolist() == [2 if j == 9 else 0 for j in range(B)]
    for j in range(B):
        got = [int.from_bytes(bytes(out[j, k]), "little") for k in range(n)]
        assert got == (want[j] if j != 9 else [0] * n)
        if j != 9:
            assert got == o.poly_interpolate(list(zip(xs_[j], ys_[j])))


def test_ref_distributed_key_generation(engine, rnd):
    """distributed_key_generation (src/poly.rs:818-900) with the reference's sizes: 3 dealers, 5 nodes,
    faulty_num = 2; every group element through the HIP kernels, the secret Fr arithmetic on the host as in
    the reference; each device result is also compared with Oracle A."""
    api.set_default_engine(engine)
    dealer_num, node_num, faulty_num = 3, 5, 2
    ncoef = (faulty_num + 1) * (faulty_num + 2) // 2
    bi_polys = [BivarPoly(faulty_num, [rnd.randrange(o.R) for _ in range(ncoef)]) for _ in range(dealer_num)]
    pub_bi_commits = [bp.commitment() for bp in bi_polys]
    for bp, bc in zip(bi_polys, pub_bi_commits):
        assert bc.coeff == [o.g1_uncompressed(c) for c in o.bivar_commitment(bp.coeff)]
    sec_keys = [0] * node_num
    for bi_poly, bi_commit in zip(bi_polys, pub_bi_commits):
        row_commits = bi_commit.row_batch(list(range(0, node_num + 1)))
        for m in range(1, node_num + 1):
            row_poly = bi_poly.row(m)
            row_commit = row_commits[m]
            assert row_poly.commitment() == row_commit                       # :847
            vals = [row_poly.evaluate(s) for s in range(1, node_num + 1)]
            val_g1 = Poly(vals).commitment().coeff if any(vals) else []
            evals = row_commit.evaluate_batch(list(range(1, node_num + 1)))  # bi_commit.evaluate(m, s) = row(m).evaluate(s)
            for s in range(1, node_num + 1):
                assert evals[s - 1] == o.g1_uncompressed(o.E1.mul(o.G1_GEN, vals[s - 1]))   # :852
                assert bi_poly.evaluate(m, s) == vals[s - 1]                  # :854
            assert [bytes(v) for v in val_g1] == evals[: len(val_g1)]
            # a cheating dealer is detected (:858-861)
            wrong_poly = row_poly + Poly([0, 0, 5])
            assert wrong_poly.commitment() != row_commit
            received = {i: bi_poly.evaluate(m, i) for i in (1, 2, 4)}
            my_row = Poly.interpolate(received)
            assert bi_poly.evaluate(m, 0) == my_row.evaluate(0) and row_poly == my_row   # :876-877
            sec_keys[m - 1] = (sec_keys[m - 1] + my_row.evaluate(0)) % o.R
    sec_key_set = Poly([])
    for bp in bi_polys:
        sec_key_set = sec_key_set + bp.row(0)
    for m in range(1, node_num + 1):
        assert sec_key_set.evaluate(m) == sec_keys[m - 1]                         # :891
    # the sum of the first rows of the public commitments commits to the secret key set (:895-899): compare the
    # device rows with the commitment of the summed polynomial, coefficient sums taken by the oracle
    rows0 = [bc.row(0) for bc in pub_bi_commits]
    want = sec_key_set.commitment()
    for k in range(faulty_num + 1):
        acc = None
        for r in rows0:
            acc = o.E1.add(acc, o.g1_from_uncompressed(r.coeff[k]))
        assert o.g1_uncompressed(acc) == want.coeff[k]
    assert BivarCommitment(faulty_num, pub_bi_commits[0].coeff).evaluate(3, 4) == o.g1_uncompressed(
        o.bivar_commitment_evaluate(faulty_num, o.bivar_commitment(bi_polys[0].coeff), 3, 4))
    assert Commitment(rows0[0].coeff).degree() == faulty_num


def test_commitment_evaluate_any_into_fr_abscissa(engine, rnd):
    """Commitment::evaluate takes any `T: IntoFr` (src/poly.rs:497-508, src/into_fr.rs): 0, the u64 range,
    2^64 (the share index 2^64 - 1), negative i64 values (they wrap modulo r) and full-size field elements -- the
    out-of-u64 ones run as a linear combination of powers instead of re-entering the u64 path."""
    api.set_default_engine(engine)
    poly = [rnd.randrange(o.R) for _ in range(4)]
    commit = o.commitment(poly)
    c = Commitment([o.g1_uncompressed(p) for p in commit])
    xs = [0, 1, 5, 2 ** 64 - 1, 2 ** 64, -1, -7, o.R - 3, rnd.randrange(o.R)]
    got = c.evaluate_batch(xs)
    for x, g in zip(xs, got):
        assert g == o.g1_uncompressed(o.commitment_evaluate(commit, x % o.R)), x
    assert c.evaluate(2 ** 64) == o.g1_uncompressed(o.E1.mul(o.G1_GEN, o.poly_evaluate(poly, 2 ** 64)))
    # BivarCommitment.evaluate reaches the same path through row(x).evaluate(y)
    bp = BivarPoly(1, [rnd.randrange(o.R) for _ in range(3)])
    bc = bp.commitment()
    assert bc.evaluate(2, 2 ** 64) == o.g1_uncompressed(o.E1.mul(o.G1_GEN, bp.evaluate(2, 2 ** 64)))
