"""GPU parity of the round-4 boundary entries, through the C ABI:

  * the WIRE-LEVEL combiners (tc_combine_signatures_wire_batch / tc_decrypt_wire_batch): shares as they travel -- 96-byte
    Signature::to_bytes, 48-byte compressed G1 -- through the checked decode of from_bytes
    (/root/reference/src/lib.rs:140-146, 246-252) on the device, the combined signature back as Signature::to_bytes
    (src/lib.rs:255-259).  Checker: Oracle B's g2_decompress -> combine -> g2_compress, job by job, at BASELINE config 2's
    full batch with planted undecodable and non-member shares;
  * SecretKeyShare::decrypt_share and SecretKey::decrypt as ONE call each (tc_decrypt_share_batch, tc_secret_key_decrypt_batch);
  * `T: IntoFr` beyond u64 (tc_combine_g{1,2}_fr_batch, tc_decrypt_fr_batch): the reference's test_interpolate
    (src/lib.rs:793-808, whose indices are i32) replayed with negative integers, field elements and 2^64-range values.
"""
import os
import random

import numpy as np
import pytest

import c_oracle as c
import tc_oracle as o
from threshold_crypto_amd import api
from threshold_crypto_amd.engine import pack_messages

pytestmark = pytest.mark.gpu
B_FULL = int(os.environ.get("TC_TEST_BASELINE_JOBS", "65536"))


def u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


@pytest.fixture(scope="module")
def rnd():
    return random.Random(20260927)


def non_member_g2(rnd):
    """an on-curve point of E'(Fq2) outside the order-r subgroup (its cofactor is ~2^508: a random point never is inside)"""
    while True:
        P = o.g2_get_point_from_x((rnd.randrange(o.Q), rnd.randrange(o.Q)), True)
        if P is not None and o.E2.mul(P, o.R) is not None:
            return P


def non_member_g1(rnd):
    while True:
        x = rnd.randrange(o.Q)
        y2 = (x * x * x + 4) % o.Q
        y = pow(y2, (o.Q + 1) // 4, o.Q)
        if y * y % o.Q == y2 and o.E1.mul((x, y), o.R) is not None:
            return (x, y)


def test_wire_combine_edge_cases_vs_oracle(engine, rnd):
    """70 jobs (a wave boundary inside), t = 2 of n = 5 shares each: valid jobs; an undecodable share (x >= q; the
    uncompressed flag; a non-square x^3 + b); an on-curve NON-member; the identity as a share; garbage in the samples
    BEYOND the first t + 1 (interpolate() never looks at them); both roots of one x (the sort bit matters).  Status and
    bytes equal Oracle B's decompress -> combine -> compress for every job; failed jobs return the identity's encoding."""
    t, n, B = 2, 5, 70
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    H = [o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R)) for _ in range(B)]
    idx = np.array([sorted(rnd.sample(range(12), n)) for _ in range(B)], dtype=np.uint64)
    sh = np.zeros((B, n, 96), dtype=np.uint8)
    for j in range(B):
        for k in range(n):
            sh[j, k] = u8(o.g2_compressed(o.E2.mul(H[j], o.poly_evaluate(poly, int(idx[j, k]) + 1))))
    sh[3, 1] = 0xFF                                            # x >= q (flags say compressed, not infinity)
    sh[5, 0, 0] &= 0x7F                                        # the compressed flag missing
    bad_x = None
    while bad_x is None:                                       # x with x^3 + b a non-square: no point at all
        cand = (rnd.randrange(o.Q), rnd.randrange(o.Q))
        if o.g2_get_point_from_x(cand, True) is None:
            bad_x = cand
    enc = bytearray(bad_x[1].to_bytes(48, "big") + bad_x[0].to_bytes(48, "big"))
    enc[0] |= 0x80
    sh[7, 2] = u8(bytes(enc))
    sh[9, 1] = u8(o.g2_compressed(non_member_g2(rnd)))        # on the curve, outside G2
    sh[11, 0] = u8(o.g2_compressed(None))                      # the identity IS a member
    sh[13, 3:] = 0xAB                                          # beyond the first t+1 samples: never read
    sh[15, 1, 0] ^= 0x20                                       # the other root: a different (valid) share
    sh[64, 2] = 0                                              # first job of the second wave: all-zero bytes
    out, st = engine.combine_signatures_wire(t, idx, sh)
    for j in range(B):
        rc, want = c.combine_signatures_wire(t, [int(i) for i in idx[j]], [bytes(x) for x in sh[j]])
        assert rc == int(st[j]) and bytes(out[j]) == want, (j, rc, int(st[j]))
    assert [int(st[j]) for j in (3, 5, 7, 9, 64)] == [3] * 5 and not st[[0, 11, 13, 15]].any()
    assert bytes(out[3]) == bytes([0xC0]) + bytes(95)
    assert bytes(out[0]) == o.g2_compressed(o.E2.mul(H[0], poly[0]))
    # too few shares: NotEnoughShares for every job, as src/lib.rs:731-733
    out, st = engine.combine_signatures_wire(t, idx[:, :2].copy(), sh[:, :2].copy())
    assert (st == 1).all()
    # the host-side mirror: bytes in, bytes out
    pk_set = api.PublicKeySet([o.g1_uncompressed(o.E1.mul(o.G1_GEN, cf)) for cf in poly], _trusted=True)
    sigs, st = pk_set.combine_signatures_wire_batch([[(int(idx[j, k]), bytes(sh[j, k])) for k in range(n)] for j in (0, 1, 9)], engine=engine)
    assert [int(s) for s in st] == [0, 0, 3] and sigs[0] == o.g2_compressed(o.E2.mul(H[0], poly[0]))


def test_wire_decrypt_vs_oracle(engine, rnd):
    """PublicKeySet::decrypt with 48-byte decryption shares: plaintexts and statuses equal Oracle B's
    g1_decompress -> threshold_decrypt on 70 ciphertexts, incl. an undecodable and a non-member share."""
    t, n, B = 3, 4, 70
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    pk = o.E1.mul(o.G1_GEN, poly[0])
    idx = np.array([sorted(rnd.sample(range(10), n)) for _ in range(B)], dtype=np.uint64)
    sh = np.zeros((B, n, 48), dtype=np.uint8)
    vs = []
    for j in range(B):
        r = rnd.randrange(1, o.R)
        u = o.E1.mul(o.G1_GEN, r)
        msg = bytes(rnd.randrange(256) for _ in range(1 + j % 40))
        vs.append(o.xor_with_hash(o.E1.mul(pk, r), msg))
        for k in range(n):
            sh[j, k] = u8(o.g1_compressed(o.E1.mul(u, o.poly_evaluate(poly, int(idx[j, k]) + 1))))
    sh[4, 2] = 0xFF
    sh[6, 0] = u8(o.g1_compressed(non_member_g1(rnd)))
    v, off = pack_messages(vs)
    out, st = engine.decrypt_wire(t, idx, sh, v, off)
    for j in range(B):
        rc, want = c.decrypt_wire(t, [int(i) for i in idx[j]], [bytes(x) for x in sh[j]], vs[j])
        assert rc == int(st[j]) and bytes(out[int(off[j]):int(off[j + 1])]) == want, j
    assert int(st[4]) == 3 and int(st[6]) == 3 and st.sum() == 6


def test_wire_combine_full_baseline_batch_vs_oracle(engine):
    """BASELINE config 2 on the wire: 65 536 jobs x 4 compressed shares -> 65 536 compressed signatures, EVERY job against
    Oracle B (from_bytes -> interpolate -> to_bytes on all host threads), with an undecodable share planted in every 4096th
    job (+5) and a non-member in every 4096th (+9).  TC_TEST_BASELINE_JOBS shrinks it for local iterations."""
    from threshold_crypto_amd.workload import ThresholdSigWorkload
    rnd = random.Random(7)
    wl = ThresholdSigWorkload(engine, 3, 10, B_FULL)
    B = B_FULL
    comp, st = engine.g2_compress(np.ascontiguousarray(wl.shares.reshape(B * 4, 192)))
    assert not st.any()
    sh = comp.reshape(B, 4, 96).copy()
    outsider = u8(o.g2_compressed(non_member_g2(rnd)))
    sh[5::4096, 1] = 0xFF
    sh[9::4096, 2] = outsider
    out, st = engine.combine_signatures_wire(3, wl.idx, sh)
    want, rc = c.combine_signatures_wire_batch(3, wl.idx, sh, c.host_threads())
    assert (rc == st.astype(np.int32)).all() and (out == want).all()
    bad = np.zeros(B, bool)
    bad[5::4096] = True
    bad[9::4096] = True
    assert ((st == 3) == bad).all()


# ---- SecretKeyShare::decrypt_share / SecretKey::decrypt as ONE call each ----------------------------------------------------
def test_decrypt_share_and_secret_key_decrypt_in_one_call_vs_oracle(engine, rnd):
    """SecretKeyShare::decrypt_share (src/lib.rs:452-457) and SecretKey::decrypt (src/lib.rs:384-391) for 70 ciphertexts of
    ragged length under one key: Ciphertext::verify, [sk] u (and xor_with_hash) in one call.  Oracle B recomputes every job as
    the composition the reference writes (ciphertext_verify, then g1_mul, then xor_with_hash); a corrupted v, a swapped w, a
    non-member w and an undecodable u must give ok = 0, the identity as the share and zeros as the plaintext -- never [sk] u.
    Host buffers and device-resident buffers agree."""
    import torch
    B = 70
    sk = rnd.randrange(1, o.R)
    pk = u8(o.g1_uncompressed(o.E1.mul(o.G1_GEN, sk)))
    msgs = [bytes(rnd.randrange(256) for _ in range(j % 37)) for j in range(B)]          # job 0 and 37: empty plaintexts
    flat, off = pack_messages(msgs)
    r = np.stack([u8(rnd.randrange(1, o.R).to_bytes(32, "little")) for _ in range(B)])
    u, v, w, st = engine.encrypt(pk, r, flat, off)
    assert not st.any()
    u, v, w = u.copy(), v.copy(), w.copy()
    v[int(off[5])] ^= 1                                            # a flipped ciphertext byte
    w[9] = w[10]                                                   # the neighbour's w
    w[11] = u8(o.g2_uncompressed(non_member_g2(rnd)))              # on the curve, outside G2
    u[13, 50] ^= 4                                                 # not on the curve any more
    bad = {5, 9, 11, 13}
    fr = u8(sk.to_bytes(32, "little"))
    shares, ok = engine.decrypt_share(fr, u, v, off, w)
    plain, ok2 = engine.secret_key_decrypt(fr, u, v, off, w)
    identity = bytes([0x40]) + bytes(95)
    for j in range(B):
        vj = bytes(v[int(off[j]):int(off[j + 1])])
        good = c.ciphertext_verify(bytes(u[j]), vj, bytes(w[j])) if j != 13 else False
        assert good == (j not in bad) and bool(ok[j]) == good == bool(ok2[j]), j
        got = bytes(plain[int(off[j]):int(off[j + 1])])
        if not good:
            assert bytes(shares[j]) == identity and got == bytes(len(vj)), j
            continue
        rc, g = c.g1_mul(bytes(fr), bytes(u[j]))
        assert rc == 0 and bytes(shares[j]) == g, j
        rc, pt = c.xor_with_hash(g, vj)
        assert rc == 0 and got == pt == msgs[j], j
    # device-resident operands: the same bytes
    dev = [torch.from_numpy(x).cuda() for x in (fr, u, v, off, w)]
    shares_d, ok_d = engine.decrypt_share(*dev)
    plain_d, ok_d2 = engine.secret_key_decrypt(*dev)
    engine.sync()
    assert (shares_d.cpu().numpy() == shares).all() and (ok_d.cpu().numpy() == ok).all()
    assert (plain_d.cpu().numpy() == plain).all() and (ok_d2.cpu().numpy() == ok2).all()
    # the host-side mirror: Option<DecryptionShare> / Option<Vec<u8>>
    cts = [api.Ciphertext(bytes(u[j]), bytes(v[int(off[j]):int(off[j + 1])]), bytes(w[j]), _trusted=True) for j in (0, 1, 5, 9)]
    sks = api.SecretKeyShare(sk)
    got = sks.decrypt_share_batch(cts, engine=engine)
    assert got[2] is None and got[3] is None and got[1].to_bytes() == o.g1_compressed(o.E1.mul(o.g1_from_uncompressed(bytes(u[1])), sk))
    assert api.SecretKey(sk).decrypt_batch(cts, engine=engine) == [msgs[0], msgs[1], None, None]


def test_decrypt_share_never_multiplies_the_secret_by_a_point_outside_g1(engine, rnd):
    """u' = u1 + T with T of small order in E(Fq) passes the pairing check of Ciphertext::verify (e(T, H) = 1), and
    [sk] u' would leak sk modulo the order of T.  The reference cannot hold such a Ciphertext (checked decode,
    src/lib.rs:140-146); tc_decrypt_share_batch / tc_secret_key_decrypt_batch therefore test u and w for membership even with the
    context's input checks OFF.  Also: a non-canonical secret key (>= r) is an error for every job, not an identity share."""
    sk = rnd.randrange(1, o.R)
    r_enc = rnd.randrange(1, o.R)
    v = bytes(rnd.randrange(256) for _ in range(24))
    T = o.E1.mul(non_member_g1(rnd), o.R)           # the cofactor component of a random curve point: order divides h1
    assert T is not None and o.E1.mul(T, o.R) is not None
    u_bad = o.E1.add(o.E1.mul(o.G1_GEN, r_enc), T)
    w_bad = o.E2.mul(o.hash_g1_g2(u_bad, v), r_enc)
    u_good = o.E1.mul(o.G1_GEN, r_enc)
    w_good = o.E2.mul(o.hash_g1_g2(u_good, v), r_enc)
    u = np.stack([u8(o.g1_uncompressed(u_bad)), u8(o.g1_uncompressed(u_good))])
    w = np.stack([u8(o.g2_uncompressed(w_bad)), u8(o.g2_uncompressed(w_good))])
    flat, off = pack_messages([v, v])
    fr = u8(sk.to_bytes(32, "little"))
    identity = bytes([0x40]) + bytes(95)
    was = engine.input_checks()
    try:
        engine.set_input_checks(False)
        assert engine.ciphertext_verify(u, flat, off, w).tolist() == [1, 1]    # the pairing equation alone accepts u'
        shares, ok = engine.decrypt_share(fr, u, flat, off, w)
        plain, ok2 = engine.secret_key_decrypt(fr, u, flat, off, w)
        assert ok.tolist() == [0, 1] == ok2.tolist()
        assert bytes(shares[0]) == identity and bytes(plain[:24]) == bytes(24)
        assert bytes(shares[1]) == o.g1_uncompressed(o.E1.mul(u_good, sk))
        big = u8(o.R.to_bytes(32, "little"))                                     # sk = r: not a canonical Fr
        shares, ok = engine.decrypt_share(big, u, flat, off, w)
        assert ok.tolist() == [0, 0] and bytes(shares[1]) == identity
        engine.set_input_checks(True)
        assert engine.ciphertext_verify(u, flat, off, w).tolist() == [0, 1]
    finally:
        engine.set_input_checks(was)


# ---- `T: IntoFr` --------------------------------------------------------------------------------------------------------
def fr_rows(ids):
    return np.stack([np.stack([u8((i % o.R).to_bytes(32, "little")) for i in row]) for row in ids])


def test_interpolate_with_into_fr_indices(engine, rnd):
    """test_interpolate (src/lib.rs:793-808) for every degree 0..4, in BOTH groups, with the index types u64 cannot carry:
    negative i32 / i64 values (-(|x|) mod r, src/into_fr.rs:28-56), field elements (src/into_fr.rs:10-14), 2^64 and above.
    Each job interpolates (x - 1, comm.evaluate(x)) and must return comm.evaluate(0); Oracle A's interpolate -- which
    takes any integer through into_fr_plus_1 -- recomputes every job.  A batch of plain small indices sent through the Fr
    entry must equal the u64 entry's result (it runs the u64 kernels), and a non-canonical abscissa fails its job."""
    for deg in range(5):
        B = 6
        poly = [rnd.randrange(o.R) for _ in range(deg + 1)]
        rows = [[-(k + 2) for k in range(deg + 1)],                                     # negative integers
                [rnd.randrange(o.R) for _ in range(deg + 1)],                           # field elements
                [2 ** 64 + k for k in range(deg + 1)],                                  # beyond u64
                [(-1) ** k * (3 * k + 1) for k in range(deg + 1)],                      # mixed signs
                [o.R - 1 - 2 * k for k in range(deg + 1)],                              # r - 1: its interpolation point is 0 ...
                [5 * k + 2 for k in range(deg + 1)]]                                    # ... and ordinary indices
        idx_fr = fr_rows(rows)
        for g2 in (False, True):
            E, gen, enc = (o.E2, o.G2_GEN, o.g2_uncompressed) if g2 else (o.E1, o.G1_GEN, o.g1_uncompressed)
            pts = [[E.mul(gen, o.poly_evaluate(poly, (i + 1) % o.R)) for i in row] for row in rows]
            sh = np.stack([np.stack([u8(enc(p)) for p in row]) for row in pts])
            out, st = (engine.combine_g2_fr if g2 else engine.combine_g1_fr)(deg, idx_fr, sh)
            assert not st.any()
            for j in range(B):
                assert bytes(out[j]) == enc(E.mul(gen, poly[0])) == enc(o.interpolate(E, deg, list(zip(rows[j], pts[j])))), (deg, g2, j)
            # the same small indices through the u64 entry
            small = np.array([rows[5]] * 2, dtype=np.uint64)
            o64, st64 = (engine.combine_g2 if g2 else engine.combine_g1)(deg, small, np.stack([sh[5]] * 2))
            ofr, stfr = (engine.combine_g2_fr if g2 else engine.combine_g1_fr)(deg, fr_rows([rows[5]] * 2), np.stack([sh[5]] * 2))
            assert not st64.any() and not stfr.any() and (o64 == ofr).all() and bytes(o64[0]) == bytes(out[5])
        # a repeated abscissa is filtered BY VALUE from the denominator (src/lib.rs:757-762): whatever comes out equals the
        # reference's construction, here Oracle A's
        if deg >= 2:
            dup = [[o.R - 5, 7, o.R - 5] + [100 + k for k in range(deg - 2)]]
            pts = [[o.E1.mul(o.G1_GEN, o.poly_evaluate(poly, (i + 1) % o.R)) for i in dup[0]]]
            outd, std = engine.combine_g1_fr(deg, fr_rows(dup), np.stack([np.stack([u8(o.g1_uncompressed(p)) for p in pts[0]])]))
            assert not std.any() and bytes(outd[0]) == o.g1_uncompressed(o.interpolate(o.E1, deg, list(zip(dup[0], pts[0]))))
        # a non-canonical abscissa (>= r) fails its own job only
        bad = idx_fr.copy()
        bad[1, 0] = 0xFF
        out2, st2 = engine.combine_g1_fr(deg, bad, np.stack([np.stack([u8(o.g1_uncompressed(o.E1.mul(o.G1_GEN, o.poly_evaluate(poly, (i + 1) % o.R))))
                                                                          for i in row]) for row in rows]))
        assert int(st2[1]) == 3 and not st2[[0, 2, 3, 4, 5]].any()


def test_api_mirror_takes_into_fr_keys(engine, rnd):
    """PublicKeySet::combine_signatures / decrypt are generic over `T: IntoFr` (src/lib.rs:608-622): the host mirror takes
    share maps keyed by negative integers and by Fr values, iterated in the key type's order like the reference's BTreeMap,
    and returns the master key's signature / the plaintext; duplicate u64-range keys keep working."""
    api.set_default_engine(engine)
    t = 2
    sk_set = api.SecretKeySet([rnd.randrange(o.R) for _ in range(t + 1)])
    pk_set = sk_set.public_keys()
    msg = b"IntoFr keys"
    want = o.g2_uncompressed(o.sign(sk_set.poly[0], msg))
    for keys in ([-5, -1, 7], [api.Fr(rnd.randrange(o.R)) for _ in range(3)], [2 ** 64, 3, -(2 ** 31)]):
        shares = {k: sk_set.secret_key_share(k).sign(msg) for k in keys}
        sig = pk_set.combine_signatures(shares)
        assert sig.raw == want and pk_set.public_key().verify(sig, msg)
    ct = pk_set.public_key().encrypt_with_r(rnd.randrange(1, o.R), b"secret")
    for keys in ([-3, 4, 9], [api.Fr(12345), api.Fr(o.R - 7), api.Fr(6)]):
        dshares = {k: sk_set.secret_key_share(k).decrypt_share_no_verify(ct) for k in keys}
        assert pk_set.decrypt(dshares, ct) == b"secret"


def test_g1_two_wave_kernels_equal_the_one_wave_kernels(engine, rnd):
    """The single G1 multiplication and the G1 combine fast path run their TWO-waves-per-SIMD builds (k_g1_mul_arena,
    k_combine_fast_g1_arena: 256 registers, the base-4 GLV ladder's table in the lane's arena entries) at every batch size.  A
    70 000-job call -- more waves than SIMDs, so lanes share their SIMD and the arena slots turn over -- must return exactly what
    two calls of 35 000 jobs return: DecryptionShare generation (src/lib.rs:460-462) and PublicKeySet::decrypt's combination
    (src/lib.rs:618-626) with the D = 1, 2^a and generic denominators all present, an identity operand, an undecodable one --
    and Oracle B recomputes a sample."""
    B, t = 70000, 3
    e = engine
    e.set_input_checks(False)
    try:
        r = np.random.default_rng(11)
        fr = r.integers(0, 256, size=(B, 32), dtype=np.uint8)
        fr[:, 31] &= 0x3F
        base, st = e.g1_mul(fr[:1].copy(), np.tile(u8(o.g1_uncompressed(o.G1_GEN))[None], (B, 1)))     # one point ...
        pts = np.ascontiguousarray(base[:, 0])
        scal = fr[1:5].copy()
        sh, st = e.g1_mul(scal, pts)                                                                   # (B, 4, 96): four multiples each
        assert not st.any()
        pts[7] = u8(o.g1_uncompressed(None))
        pts[9, 3] ^= 0x40
        one, st1 = e.g1_mul(fr[5:6].copy(), pts)
        halves = [e.g1_mul(fr[5:6].copy(), np.ascontiguousarray(pts[lo:lo + 35000])) for lo in (0, 35000)]
        assert (one == np.concatenate([h[0] for h in halves])).all() and (st1 == np.concatenate([h[1] for h in halves])).all()
        assert int(st1[9, 0]) == 3 and bytes(one[7, 0]) == o.g1_uncompressed(None)
        for j in (0, 1, 35000, B - 1):
            assert bytes(one[j, 0]) == c.g1_mul(bytes(fr[5]), bytes(pts[j]))[1]
        subsets = [[0, 1, 2, 3], [0, 1, 2, 4], [1, 3, 5, 7], [0, 2, 5, 9], [2, 3, 4, 8]]                # D = 12, 24? ... mixed classes
        idx = np.array([subsets[j % len(subsets)] for j in range(B)], dtype=np.uint64)
        sh[11, 1] = u8(o.g1_uncompressed(None))
        sh[13, 2, 5] ^= 0x01
        comb, stc = e.combine_g1(t, idx, sh)
        parts = [e.combine_g1(t, np.ascontiguousarray(idx[lo:lo + 35000]), np.ascontiguousarray(sh[lo:lo + 35000])) for lo in (0, 35000)]
        assert (comb == np.concatenate([p[0] for p in parts])).all() and (stc == np.concatenate([p[1] for p in parts])).all()
        assert int(stc[13]) == 3 and stc.sum() == 3
        for j in (0, 1, 2, 3, 4, 11, 35001, B - 1):
            rc, want = c.combine_g1(t, [int(i) for i in idx[j]], [bytes(x) for x in sh[j]])
            assert rc == 0 and want == bytes(comb[j]), j
    finally:
        e.set_input_checks(True)


def test_g1_combination_grouped_by_class_equals_ungrouped(engine):
    """From 524 288 jobs on the G1 fast path groups its jobs by the class of their common denominator (k_combine_classify /
    k_combine_scatter, a wave of D = 1 jobs skips the [1 / D] ladder: csrc/tc_launch.h kG1GroupMinJobs).  PublicKeySet::decrypt's
    combination (src/lib.rs:618-626) over 524 288 jobs with the bench's random 4-of-10 subsets -- every class present -- must
    return exactly what eight ungrouped calls of 65 536 jobs return; Oracle B recomputes 64 jobs spread over the batch."""
    from threshold_crypto_amd.workload import ThresholdSigWorkload
    e = engine
    e.set_input_checks(False)
    try:
        wl = ThresholdSigWorkload(e, 3, 10, 65536)
        B = 524288
        r = np.random.default_rng(23)
        fr = r.integers(0, 256, size=(4, 32), dtype=np.uint8)
        fr[:, 31] &= 0x3F
        pts, st = e.g1_mul(fr, np.tile(wl.master_pk[None], (B, 1)))      # (B, 4, 96): four "decryption shares" per job
        assert not st.any()
        idx = np.ascontiguousarray(np.tile(wl.idx, (B // 65536, 1)))
        big, stb = e.combine_g1(3, idx, pts)
        assert not stb.any()
        for lo in range(0, B, 65536):
            part, stp = e.combine_g1(3, np.ascontiguousarray(idx[lo:lo + 65536]), np.ascontiguousarray(pts[lo:lo + 65536]))
            assert not stp.any() and (part == big[lo:lo + 65536]).all(), lo
        pick = np.unique(np.linspace(0, B - 1, 64).astype(np.int64))
        want, rc = c.combine_g1_batch(3, idx[pick], pts[pick], c.host_threads())
        assert not rc.any() and (want == big[pick]).all()
    finally:
        e.set_input_checks(True)


def test_two_jobs_per_lane_pair_forms_equal_the_one_job_forms(engine, rnd):
    """r05 (tc_duo.h): from 65 536 jobs on the checked G2 decode (from_bytes, /root/reference/src/lib.rs:246-252), hash_g2
    (:691-694) and hash_g1_g2 (:697-707) give a lane pair TWO jobs and run each job's Fq-only phases on one lane.  Forced on
    and off (TC_DUO_MIN, read when a context is created: one context per form) over the SAME odd-sized batches -- members, the identity, points outside G2, x
    without a point, x >= q, flag errors in either slot of a pair; ragged messages on both sides of the SHA3 rate and of the
    64-byte switch; undecodable G1 operands -- both forms must return identical bytes and statuses, and a sample of them is
    recomputed by Oracle A here.  (The full-size tests run the two-job form against Oracle B on every job.)"""
    n = 1023
    enc, want_st = [], []
    members = [o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R)) for _ in range(6)]
    outside = [non_member_g2(rnd) for _ in range(3)]
    nopoint = []
    while len(nopoint) < 3:
        x = (rnd.randrange(o.Q), rnd.randrange(o.Q))
        if o.g2_get_point_from_x(x, False) is None:
            e = bytearray(x[1].to_bytes(48, "big") + x[0].to_bytes(48, "big")); e[0] |= 0x80 | (0x20 * (len(nopoint) & 1))
            nopoint.append(bytes(e))
    for i in range(n):
        k = rnd.randrange(12)
        if k < 6:
            P = members[rnd.randrange(6)]
            if rnd.randrange(2):
                P = (P[0], ((-P[1][0]) % o.Q, (-P[1][1]) % o.Q))  # the other root of the same x
            enc.append(o.g2_compressed(P)); want_st.append(0)
        elif k == 6:
            enc.append(o.g2_compressed(None)); want_st.append(0)
        elif k == 7:
            enc.append(o.g2_compressed(outside[rnd.randrange(3)])); want_st.append(3)
        elif k == 8:
            enc.append(nopoint[rnd.randrange(3)]); want_st.append(3)
        elif k == 9:
            e = bytearray(o.g2_compressed(members[0])); e[0] &= 0x7f; enc.append(bytes(e)); want_st.append(3)
        elif k == 10:
            e = bytearray(o.Q.to_bytes(48, "big") + (5).to_bytes(48, "big")); e[0] |= 0x80; enc.append(bytes(e)); want_st.append(3)
        else:
            e = bytearray(o.g2_compressed(None)); e[rnd.randrange(1, 96)] |= 1 << rnd.randrange(8); enc.append(bytes(e)); want_st.append(3)
    msgs = [bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 1, 31, 32, 63, 64, 65, 88, 135, 136, 137, 200]))) for _ in range(n)]
    g1 = [o.g1_uncompressed(o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R))) for _ in range(8)] + [o.g1_uncompressed(None)]
    g1rows = []
    for i in range(n):
        p = bytearray(g1[rnd.randrange(len(g1))])
        if rnd.randrange(9) == 0:
            p[95] ^= 1  # off the curve
        g1rows.append(bytes(p))
    enc_a = np.frombuffer(b"".join(enc), np.uint8).reshape(n, 96).copy()
    g1_a = np.frombuffer(b"".join(g1rows), np.uint8).reshape(n, 96).copy()
    blob, off = pack_messages(msgs)
    res = {}
    from conftest import engine_with_env
    for form, minimum in (("two", 1), ("one", 10 ** 12)):
        with engine_with_env(TC_DUO_MIN=minimum) as eng:    # read once, when the context is created (tc_launch.h Tuning)
            assert eng.tuning()["duo_min_hash"] == eng.tuning()["duo_min_decode"] == minimum
            pts, st = eng.g2_decompress(enc_a)
            h = eng.hash_g2(blob, off)
            hg, hst = eng.hash_g1_g2(g1_a, blob, off)
            res[form] = (pts, st, h, hg, hst)
    assert engine.tuning() == {"duo_min_decode": 32769, "duo_min_hash": 131072, "pairing_form": 0, "pairing_budget": 0, "checks_beside": 1,
                               "msm_budget": 0, "private_reserve": (1 << 64) - 1}   # the defaults
    for x, y in zip(res["two"], res["one"]):
        assert (x == y).all(), np.flatnonzero((x != y).reshape(n, -1).any(axis=1))[:16]
    pts, st, h, hg, hst = res["two"]
    assert st.tolist() == want_st
    for i in list(range(0, n, 37)) + [n - 1]:
        if want_st[i] == 0:
            assert bytes(pts[i]) == o.g2_uncompressed(o.g2_from_compressed(enc[i]))
        else:
            assert bytes(pts[i]) == o.g2_uncompressed(None)
        assert bytes(h[i]) == o.g2_uncompressed(o.hash_g2(msgs[i]))
    for i in list(range(0, n, 101)) + [n - 1]:
        try:
            P = o.g1_from_uncompressed(g1rows[i])
            ok = True
        except Exception:
            ok = False
        assert int(hst[i]) == (0 if ok else 3)
        if ok:
            assert bytes(hg[i]) == o.g2_uncompressed(o.hash_g1_g2(P, msgs[i]))
        else:
            assert bytes(hg[i]) == o.g2_uncompressed(None)


def test_hashes_above_the_two_message_threshold_equal_the_one_message_form(engine, rnd):
    """131 073 messages (odd: the last lane pair holds one) -- the size from which the library hashes two messages per lane pair by
    itself (csrc/tc_launch.h kDuoMinHash) -- through hash_g2 with the library's own choice and with the one-message form forced:
    identical bytes for every message, Oracle B on a sample.  Checked decodes the same way at 32 769 points (kDuoMinDecode)."""
    n = 131073
    msgs = [b"tc/duo/%d" % i + bytes(i % 5) for i in range(n)]
    blob, off = pack_messages(msgs)
    from conftest import engine_with_env
    pts = [o.g2_compressed(o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R))) for _ in range(16)] + [o.g2_compressed(None)]
    enc = np.frombuffer(b"".join(pts[rnd.randrange(len(pts))] for _ in range(32769)), np.uint8).reshape(32769, 96).copy()
    enc[777, 5] ^= 0x40                                    # one undecodable encoding (with overwhelming probability)
    with engine_with_env(TC_DUO_MIN=None) as own_eng:      # the library's own thresholds
        assert own_eng.tuning()["duo_min_hash"] == 131072 and own_eng.tuning()["duo_min_decode"] == 32769
        own = own_eng.hash_g2(blob, off)
        own_p, own_s = own_eng.g2_decompress(enc)
    with engine_with_env(TC_DUO_MIN=10 ** 12) as one_eng:  # the one-job form forced
        one = one_eng.hash_g2(blob, off)
        one_p, one_s = one_eng.g2_decompress(enc)
    assert (own == one).all()
    for j in [0, 1, 65535, 65536, 131071, 131072] + [rnd.randrange(n) for _ in range(10)]:
        assert bytes(own[j]) == c.hash_g2(msgs[j])
    assert (own_p == one_p).all() and (own_s == one_s).all() and int(own_s.sum()) in (0, 3)


def test_membership_tests_beside_the_main_kernels_equal_the_one_stream_order(engine, rnd):
    """r06: in checked-input mode (the context's default; the reference tests every value once, in from_bytes:
    /root/reference/src/lib.rs:140-146, 246-252) the membership tests of a call run on the context's SECOND stream beside the
    call's main kernels and are joined before their verdicts are applied (tc_api.hip Call::apply_checks).  The order of execution
    must not be observable: with non-members, off-curve points and the identity planted in every operand position, every entry
    returns the same bytes, statuses and ok-vectors as a context created with TC_CHECKS_BESIDE=0 (tests first, one stream) --
    host buffers and device buffers, a batch with a wave boundary inside and one large enough for several waves per SIMD."""
    import torch
    from conftest import engine_with_env
    from threshold_crypto_amd.workload import ThresholdSigWorkload, ThresholdEncWorkload
    bad2 = [u8(o.g2_uncompressed(non_member_g2(rnd))) for _ in range(3)]
    bad1 = [u8(o.g1_uncompressed(non_member_g1(rnd))) for _ in range(3)]
    off_curve2 = u8(o.g2_uncompressed(o.E2.mul(o.G2_GEN, 7))); off_curve2[100] ^= 1
    t, N = 3, 10
    for B in (70, 9000):
        wl = ThresholdSigWorkload(engine, t, N, B)
        we = ThresholdEncWorkload(engine, t, N, B)
        sig, st = engine.combine_g2(t, wl.idx, wl.shares)
        assert not st.any()
        shares, sigs, hashes = wl.shares.copy(), sig.copy(), wl.hashes.copy()
        u, w, dsh = we.u.copy(), we.w.copy(), we.shares.copy()
        pks = np.tile(wl.master_pk[None], (B, 1))
        for n_planted, j in enumerate(sorted(rnd.sample(range(B), 24))):
            k = n_planted % 8
            if k == 0:
                shares[j, rnd.randrange(t + 1)] = bad2[n_planted % 3]
            elif k == 1:
                shares[j, rnd.randrange(t + 1)] = off_curve2
            elif k == 2:
                sigs[j] = bad2[n_planted % 3]
            elif k == 3:
                hashes[j] = bad2[n_planted % 3]
            elif k == 4:
                pks[j] = bad1[n_planted % 3]
            elif k == 5:
                u[j] = bad1[n_planted % 3]
            elif k == 6:
                w[j] = bad2[n_planted % 3]
            else:
                dsh[j, rnd.randrange(t + 1)] = bad1[n_planted % 3]
        fr = np.stack([u8(o.fr_to_bytes(rnd.randrange(1, o.R))) for _ in range(2)])
        rs = np.stack([u8(o.fr_to_bytes(rnd.randrange(1, o.R))) for _ in range(B)])
        blob, moff = pack_messages(wl.msgs)

        def run(eng, dev):
            # device-I/O calls return before their kernels have run (the caller's stream orders them): every operand tensor stays
            # alive in `keep` until the context has been synchronised -- torch would hand a dead tensor's memory to the next one
            keep = []

            def to(a):
                if not dev:
                    return a
                keep.append(torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to(dev))
                return keep[-1]
            host = (lambda x: x.cpu().numpy()) if dev else (lambda x: x)
            out = []
            out += list(eng.combine_g2(t, to(wl.idx), to(shares)))
            out += list(eng.g2_mul(to(fr), to(hashes)))
            out += [eng.verify_g2(to(pks), to(sigs), to(hashes))]
            out += [eng.verify_sig(to(pks), to(sigs), to(blob), to(moff))]
            out += [eng.pairing_check(to(pks), to(hashes), to(pks), to(hashes))]
            out += [eng.ciphertext_verify(to(u), to(we.v), to(we.off), to(w))]
            out += list(eng.decrypt(t, to(we.idx), to(dsh), to(we.v), to(we.off)))
            out += list(eng.decrypt_share(to(fr[0]), to(u), to(we.v), to(we.off), to(w)))
            out += list(eng.encrypt(to(pks), to(rs), to(blob), to(moff)))
            out += list(eng.hash_g1_g2(to(u), to(blob), to(moff)))
            eng.sync()
            return [host(x) for x in out]

        with engine_with_env(TC_CHECKS_BESIDE=0) as plain:
            assert plain.tuning()["checks_beside"] == 0 and engine.tuning()["checks_beside"] == 1
            want = run(plain, None)
            want_dev = run(plain, torch.device("cuda", 0)) if B == 70 else None
        for dev in (None, torch.device("cuda", 0)):
            got = run(engine, dev)
            assert len(got) == len(want)
            for i, (g, x) in enumerate(zip(got, want)):
                assert g.shape == x.shape and (g == x).all(), (B, dev, i)
        if want_dev is not None:
            for g, x in zip(want_dev, want):
                assert (g == x).all()
        # ... and the planted operands did fail their jobs (the tests ran): statuses / ok of the checked entries are not all-OK
        assert want[1].any() and not want[4].all() and not want[6].all() and want[9].any() and not want[11].all() and want[15].any()


def test_ragged_and_megabyte_plaintexts_encrypt_verify_decrypt_vs_oracle(engine, rnd):
    """Maximum sizes on the plaintext side: PublicKey::encrypt_with_rng (src/lib.rs:128-137), Ciphertext::verify (:508-512) and
    SecretKey::decrypt (:384-391) over ONE batch whose plaintexts are 0, 1, 15, 16, 17, 63, 64, 65 (both sides of hash_g1_g2's
    64-byte switch), 1 023, 4 096, 65 537 and 1 048 576 bytes long -- xor_with_hash draws one ChaCha word per BYTE, so the last
    job walks 65 536 keystream blocks in one lane while its neighbours finished long ago.  Every ciphertext equals Oracle B's
    composition (u = [r] g1, v = xor_with_hash([r] pk, m), w = [r] hash_g1_g2(u, v)), verifies in both, and decrypts to m."""
    lens = [0, 1, 15, 16, 17, 63, 64, 65, 1023, 4096, 65537, 1 << 20]
    B = len(lens)
    sk = rnd.randrange(1, o.R)
    pk = u8(o.g1_uncompressed(o.E1.mul(o.G1_GEN, sk)))
    rs = np.random.RandomState(20261001)
    msgs = [rs.randint(0, 256, size=n, dtype=np.uint8).tobytes() for n in lens]
    flat, off = pack_messages(msgs)
    r = np.stack([u8(rnd.randrange(1, o.R).to_bytes(32, "little")) for _ in range(B)])
    u, v, w, st = engine.encrypt(pk, r, flat, off)
    assert not st.any()
    g1_gen = o.g1_uncompressed(o.G1_GEN)
    for j in range(B):
        vj = bytes(v[int(off[j]):int(off[j + 1])])
        rc, uj = c.g1_mul(bytes(r[j]), g1_gen)
        assert rc == 0 and bytes(u[j]) == uj, j
        rc, g = c.g1_mul(bytes(r[j]), bytes(pk))
        rc2, want_v = c.xor_with_hash(g, msgs[j])
        assert rc == 0 and rc2 == 0 and vj == want_v, "v of the %d-byte plaintext" % lens[j]
        rc, h = c.hash_g1_g2(uj, vj)
        rc2, wj = c.g2_mul(bytes(r[j]), h)
        assert rc == 0 and rc2 == 0 and bytes(w[j]) == wj, j
        assert c.ciphertext_verify(uj, vj, wj), j
    assert engine.ciphertext_verify(u, v, off, w).all()
    fr = u8(sk.to_bytes(32, "little"))
    plain, ok = engine.secret_key_decrypt(fr, u, v, off, w)
    assert ok.all() and bytes(plain[: int(off[-1])]) == b"".join(msgs)
    # one flipped byte in the middle of the megabyte: that job alone fails, and returns zeros
    v2 = v.copy()
    v2[int(off[B - 1]) + (1 << 19)] ^= 0x80
    plain2, ok2 = engine.secret_key_decrypt(fr, u, v2, off, w)
    assert ok2.tolist() == [1] * (B - 1) + [0]
    assert bytes(plain2[: int(off[B - 1])]) == b"".join(msgs[:-1]) and not plain2[int(off[B - 1]):int(off[B])].any()


def test_in_place_device_calls_keep_the_tests_in_front(engine, rnd):
    """A deferred membership test reads its operand on the second stream WHILE the main kernels write their results: when the
    caller's output buffer IS the operand (device-resident, S = 1 multiplication in place; verify_g2's ok bytes written over the
    head of the signature buffer) the call must behave like a one-stream context -- tests first (tc_api.hip Call::out /
    check_points: an operand that overlaps an output is not deferred).  Same bytes and statuses as the out-of-place call,
    non-members and an off-curve point planted, at a batch with several waves per SIMD and at a small one.  (A regression test
    of the in-place semantics, not a reproducer: the stale read needs the low-priority tests to start AFTER a main wave wrote its
    result, i.e. a saturated device, and the library before the change passed this test too.)"""
    import torch
    from threshold_crypto_amd.engine import _ptr
    assert engine.input_checks()
    for B in (33, 40000):
        ks = [rnd.randrange(1, o.R) for _ in range(8)]
        base = np.stack([u8(o.g2_uncompressed(o.E2.mul(o.G2_GEN, k))) for k in ks])
        pts = np.ascontiguousarray(base[np.arange(B) % 8])
        bad = sorted(rnd.sample(range(B), 5))
        for n, j in enumerate(bad):
            pts[j] = u8(o.g2_uncompressed(non_member_g2(rnd)))
            if n == 4:
                pts[j, 100] ^= 1                                   # not even on the curve
        fr = u8(o.fr_to_bytes(rnd.randrange(1, o.R)))[None]
        want, want_st = engine.g2_mul(fr, pts)                     # host buffers, out of place
        assert sorted(np.flatnonzero(want_st[:, 0]).tolist()) == bad
        d_fr, d_pts = torch.from_numpy(fr).cuda(), torch.from_numpy(pts).cuda()
        d_st = torch.empty((B, 1), dtype=torch.uint8, device="cuda")
        engine._mode(d_fr, d_pts, d_st)
        torch.cuda.synchronize()
        engine._call("tc_g2_mul_batch", _ptr(d_fr), _ptr(d_pts), 1, B, _ptr(d_pts), _ptr(d_st))      # out == pts
        engine.sync()
        assert (d_st.cpu().numpy() == want_st).all()
        assert (d_pts.cpu().numpy() == want[:, 0]).all()
    # verify_g2 with its ok bytes written INTO the signature buffer it reads
    B = 20000
    from threshold_crypto_amd.workload import ThresholdSigWorkload
    wl = ThresholdSigWorkload(engine, 3, 10, B)
    sig, st = engine.combine_g2(3, wl.idx, wl.shares)
    sig[7] = u8(o.g2_uncompressed(non_member_g2(rnd)))
    sig[9] = sig[10]
    want = engine.verify_g2(wl.master_pk, sig, wl.hashes)
    assert want.tolist()[:12] == [1] * 7 + [0, 1, 0, 1, 1]
    d_pk, d_sig, d_h = (torch.from_numpy(x).cuda() for x in (wl.master_pk, sig, wl.hashes))
    engine._mode(d_pk, d_sig, d_h)
    torch.cuda.synchronize()
    engine._call("tc_verify_g2_batch", _ptr(d_pk), 0, _ptr(d_sig), _ptr(d_h), B, _ptr(d_sig))          # ok == sig
    engine.sync()
    assert (d_sig.reshape(-1)[:B].cpu().numpy() == want).all()
