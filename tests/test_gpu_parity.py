"""GPU parity: every C-ABI batch entry point vs the oracle (bit-exact), on seeded inputs at
sizes the Python oracle finishes in seconds.  Run with `pytest -m gpu` on an MI355X."""
import random

import numpy as np
import pytest

import tc_oracle as o
from threshold_crypto_amd.engine import pack_messages

pytestmark = pytest.mark.gpu

SEED = 0x7C5EED


def u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def g2s(points):
    return np.stack([u8(o.g2_uncompressed(p)) for p in points])


def g1s(points):
    return np.stack([u8(o.g1_uncompressed(p)) for p in points])


def frs(scalars):
    return np.stack([u8(o.fr_to_bytes(s)) for s in scalars])


@pytest.fixture(scope="module")
def rnd():
    return random.Random(SEED)


def test_version(engine):
    assert "gfx950" in engine.version()


def test_g2_mul_matches_oracle(engine, rnd):
    """SecretKeyShare::sign_g2 (src/lib.rs:442-444): S signers x B hash points."""
    S, B = 3, 70  # B > 64 so a wave boundary is crossed
    sks = [rnd.randrange(o.R) for _ in range(S)]
    sks[1] = 1
    pts = [o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R)) for _ in range(B)]
    pts[5] = None  # infinity in, infinity out
    out, st = engine.g2_mul(frs(sks), g2s(pts))
    assert st.shape == (B, S) and not st.any()
    for j in rnd.sample(range(B), 12) + [5]:
        for s in range(S):
            assert bytes(out[j, s]) == o.g2_uncompressed(o.E2.mul(pts[j], sks[s])), (j, s)


def test_g2_mul_edge_scalars_of_the_sign_aligned_ladder(engine, rnd):
    """tc_gls.h sac_recode4: even scalars (run as r - k), first digit 1, digits |x| - 1, short digits."""
    X = o.BLS_X
    ks = [0, 1, 2, 3, 4, o.R - 1, o.R - 2, o.R - 3, X - 1, X, X + 1, X ** 2, X ** 3, 1 + X ** 3, 2 * X + 2,
          1 + (X - 1) * X + (X - 1) * X ** 2 + (X - 2) * X ** 3, (X - 1) + (X - 1) * X, (X - 1) * X ** 2,
          rnd.randrange(o.R) & ~1, rnd.randrange(o.R) | 1]
    pts = [o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R)), o.G2_GEN]
    out, st = engine.g2_mul(frs(ks), g2s(pts))
    assert not st.any()
    for j, P in enumerate(pts):
        for s, k in enumerate(ks):
            assert bytes(out[j, s]) == o.g2_uncompressed(o.E2.mul(P, k)), (j, hex(k))


def test_g1_mul_matches_oracle(engine, rnd):
    """decrypt_share_no_verify (src/lib.rs:460-462)."""
    S, B = 2, 65
    sks = [rnd.randrange(o.R), 0]
    pts = [o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R)) for _ in range(B)]
    out, st = engine.g1_mul(frs(sks), g1s(pts))
    assert not st.any()
    for j in rnd.sample(range(B), 10):
        assert bytes(out[j, 0]) == o.g1_uncompressed(o.E1.mul(pts[j], sks[0]))
        assert bytes(out[j, 1]) == o.g1_uncompressed(None)


def test_mul_rejects_bad_encodings(engine):
    pt = bytearray(o.g2_uncompressed(o.G2_GEN))
    pt[191] ^= 1  # off the curve
    big = (o.R).to_bytes(32, "little")  # non-canonical scalar
    out, st = engine.g2_mul(np.stack([u8(o.fr_to_bytes(7)), u8(big)]), np.stack([u8(pt), u8(o.g2_uncompressed(o.G2_GEN))]))
    assert st[0, 0] == 3 and st[0, 1] == 3 and st[1, 1] == 3 and st[1, 0] == 0
    assert bytes(out[1, 0]) == o.g2_uncompressed(o.E2.mul(o.G2_GEN, 7))


@pytest.mark.parametrize("t", [0, 1, 3, 5])
def test_combine_g2_matches_oracle(engine, rnd, t):
    """PublicKeySet::combine_signatures (src/lib.rs:608-615) on per-job signer subsets."""
    B, N = 66, 10
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    sk = [o.secret_key_share(poly, i) for i in range(N)]
    idx = np.zeros((B, t + 1), dtype=np.uint64)
    shares = np.zeros((B, t + 1, 192), dtype=np.uint8)
    expect = []
    cache = {}
    for j in range(B):
        h = o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R)) if j < 6 else cache["h"]
        cache["h"] = h
        ids = sorted(rnd.sample(range(N), t + 1))
        idx[j] = ids
        pts = []
        for k, i in enumerate(ids):
            key = (id(h), i)
            if key not in cache:
                cache[key] = o.E2.mul(h, sk[i])
            pts.append(cache[key])
            shares[j, k] = u8(o.g2_uncompressed(cache[key]))
        expect.append((h, ids, pts))
    out, st = engine.combine_g2(t, idx, shares)
    assert not st.any()
    for j in list(range(8)) + rnd.sample(range(8, B), 6):
        h, ids, pts = expect[j]
        want = o.combine_signatures(t, list(zip(ids, pts)))
        assert want == o.E2.mul(h, poly[0])
        assert bytes(out[j]) == o.g2_uncompressed(want), j


def test_combine_takes_first_t_plus_1_and_flags_too_few(engine, rnd):
    """interpolate: take(t+1) (src/lib.rs:728) and NotEnoughShares (:731-733)."""
    t, N = 2, 7
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    h = o.E2.mul(o.G2_GEN, 99)
    ids = [0, 2, 3, 5, 6]
    pts = [o.E2.mul(h, o.secret_key_share(poly, i)) for i in ids]
    idx = np.array([ids], dtype=np.uint64)
    shares = g2s(pts)[None]
    out, st = engine.combine_g2(t, idx, shares)  # 5 supplied, first 3 used
    assert st[0] == 0 and bytes(out[0]) == o.g2_uncompressed(o.E2.mul(h, poly[0]))
    out, st = engine.combine_g2(t, idx[:, :2].copy(), shares[:, :2].copy())  # 2 <= t
    assert st[0] == 1


def test_combine_duplicate_index_quirk(engine, rnd):
    """Equal indices are filtered by VALUE from the denominator (src/lib.rs:758): no error,
    and the (wrong) point the reference would return is reproduced bit-exactly."""
    t = 2
    h = o.E2.mul(o.G2_GEN, 5)
    pts = [o.E2.mul(h, k) for k in (3, 4, 9)]
    ids = [1, 1, 4]
    out, st = engine.combine_g2(t, np.array([ids], dtype=np.uint64), g2s(pts)[None])
    assert st[0] == 0
    assert bytes(out[0]) == o.g2_uncompressed(o.interpolate(o.E2, t, list(zip(ids, pts))))


def test_combine_g1_and_decrypt(engine, rnd):
    """PublicKeySet::decrypt (src/lib.rs:618-626)."""
    t, N, B = 3, 10, 5
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    pk = o.public_key(poly[0])
    idx = np.zeros((B, t + 1), dtype=np.uint64)
    shares = np.zeros((B, t + 1, 96), dtype=np.uint8)
    vs, plains = [], []
    for j in range(B):
        msg = bytes(rnd.randrange(256) for _ in range(5 + 13 * j))
        ct = o.encrypt_with_r(pk, rnd.randrange(1, o.R), msg)
        ids = sorted(rnd.sample(range(N), t + 1))
        idx[j] = ids
        for k, i in enumerate(ids):
            shares[j, k] = u8(o.g1_uncompressed(o.decrypt_share_no_verify(o.secret_key_share(poly, i), ct)))
        vs.append(ct[1])
        plains.append(msg)
    v, off = pack_messages(vs)
    out, st = engine.decrypt(t, idx, shares, v, off)
    assert not st.any()
    assert bytes(out[: int(off[-1])]) == b"".join(plains)
    g, st = engine.combine_g1(t, idx, shares)
    assert not st.any()
    assert bytes(g[0]) == o.g1_uncompressed(o.interpolate(o.E1, t, [(int(idx[0, k]), o.g1_from_uncompressed(bytes(shares[0, k]), check=False)) for k in range(t + 1)]))


def test_hash_g2_matches_oracle(engine, rnd):
    """hash_g2 (src/lib.rs:691-694); lengths straddle the 136-byte SHA3 rate."""
    msgs = [b"", b"a", b"Test message", bytes(range(135)), bytes(range(136)), bytes(range(137)), bytes(300)]
    msgs += [bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 80))) for _ in range(60)]
    flat, off = pack_messages(msgs)
    out = engine.hash_g2(flat, off)
    for j in list(range(7)) + rnd.sample(range(7, len(msgs)), 5):
        assert bytes(out[j]) == o.g2_uncompressed(o.hash_g2(msgs[j])), j


def test_hash_g1_g2_and_xor(engine, rnd):
    """hash_g1_g2 (src/lib.rs:697-707) both sides of the 64-byte switch; xor_with_hash (:710-715)."""
    pts = [o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R)) for _ in range(4)]
    msgs = [bytes(range(10)), bytes(range(64)), bytes(range(65)), bytes(200)]
    flat, off = pack_messages(msgs)
    out, st = engine.hash_g1_g2(g1s(pts), flat, off)
    assert not st.any()
    for j in range(4):
        assert bytes(out[j]) == o.g2_uncompressed(o.hash_g1_g2(pts[j], msgs[j])), j
    x, st = engine.xor_with_hash(g1s(pts), flat, off)
    assert not st.any()
    assert bytes(x[: int(off[-1])]) == b"".join(o.xor_with_hash(p, m) for p, m in zip(pts, msgs))


def test_pairing_check_matches_oracle(engine, rnd):
    """e(a,b) == e(c,d) (src/lib.rs:109,185,511): true, false and infinity cases."""
    B = 70
    a_, b_, c_, d_, want = [], [], [], [], []
    for j in range(B):
        x, y = rnd.randrange(1, o.R), rnd.randrange(1, o.R)
        a_.append(o.E1.mul(o.G1_GEN, x))
        b_.append(o.E2.mul(o.G2_GEN, y))
        c_.append(o.G1_GEN)
        good = (j % 3 != 1)
        d_.append(o.E2.mul(o.G2_GEN, (x * y + (0 if good else 1)) % o.R))
        want.append(1 if good else 0)
    # infinity operands: e(inf, b) == e(c, inf) -> 1 == 1
    a_[2], d_[2], want[2] = None, None, 1
    a_[3], want[3] = None, 0
    ok = engine.pairing_check(g1s(a_), g2s(b_), g1s(c_), g2s(d_))
    assert ok.tolist() == want
    # oracle agrees on a sample (two full pairings, as the reference)
    for j in (0, 1, 2, 3):
        assert o.pairing_check(a_[j], b_[j], c_[j], d_[j]) == bool(want[j])
    # broadcast of the constant operand (stride 0)
    ok2 = engine.pairing_check(g1s(a_), g2s(b_), u8(o.g1_uncompressed(o.G1_GEN)), g2s(d_))
    assert ok2.tolist() == want


def test_sign_combine_verify_pipeline(engine, rnd):
    """The doc-test of combine_signatures (src/lib.rs:583-607) and test_threshold_sig (:822-873)
    at the C ABI: sign shares -> verify shares -> combine two disjoint subsets -> same signature
    -> verifies under the master key; wrong message fails."""
    t, N = 3, 10
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    commit = o.commitment(poly)
    sks = [o.secret_key_share(poly, i) for i in range(N)]
    msgs = [b"Totally real news", b"Real news", b"Happy birthday!"]
    flat, off = pack_messages(msgs)
    shares, st = engine.sign(frs(sks), flat, off)  # (B, N, 192)
    assert not st.any()
    assert bytes(shares[0, 4]) == o.g2_uncompressed(o.sign(sks[4], msgs[0]))
    # share verification under public_key_share(i)
    pks = g1s([o.public_key_share(commit, i) for i in range(N)])
    for j in range(len(msgs)):
        m_flat, m_off = pack_messages([msgs[j]] * N)
        ok = engine.verify_sig(pks, np.ascontiguousarray(shares[j]), m_flat, m_off)
        assert ok.all()
    sets = [[5, 8, 7, 9], [0, 1, 2, 3]]
    sigs = []
    for ids in sets:
        ids = sorted(ids)
        idx = np.array([ids] * len(msgs), dtype=np.uint64)
        sh = np.ascontiguousarray(shares[:, ids, :])
        sig, st = engine.combine_g2(t, idx, sh)
        assert not st.any()
        sigs.append(sig)
    assert (sigs[0] == sigs[1]).all()
    assert bytes(sigs[0][1]) == o.g2_uncompressed(o.sign(poly[0], msgs[1]))
    pk = u8(o.g1_uncompressed(commit[0]))
    assert engine.verify_sig(pk, sigs[0], flat, off).all()
    wrong_flat, wrong_off = pack_messages([msgs[1], msgs[2], msgs[0]])
    assert not engine.verify_sig(pk, sigs[0], wrong_flat, wrong_off).any()


def test_ciphertext_verify_and_decryption_shares(engine, rnd):
    """Ciphertext::verify (src/lib.rs:508-512), verify_decryption_share (:182-186),
    test_threshold_enc (:907-939): tampered v fails."""
    t, N = 2, 5
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    commit = o.commitment(poly)
    cts = [o.encrypt_with_r(commit[0], rnd.randrange(1, o.R), m) for m in (b"Totally real news", bytes(100))]
    bad = (cts[0][0], b"X" + cts[0][1][1:], cts[0][2])
    allc = cts + [bad]
    v, off = pack_messages([c[1] for c in allc])
    ok = engine.ciphertext_verify(g1s([c[0] for c in allc]), v, off, g2s([c[2] for c in allc]))
    assert ok.tolist() == [1, 1, 0]
    assert [o.ciphertext_verify(c) for c in allc] == [True, True, False]
    # decryption shares of ct 0 by the N nodes, verified against their public key shares
    ct = cts[0]
    dsh = [o.decrypt_share_no_verify(o.secret_key_share(poly, i), ct) for i in range(N)]
    dsh_bad = list(dsh)
    dsh_bad[1] = dsh[2]
    pks = g1s([o.public_key_share(commit, i) for i in range(N)])
    v1, off1 = pack_messages([ct[1]] * N)
    u = g1s([ct[0]] * N)
    w = g2s([ct[2]] * N)
    assert engine.verify_decryption_share(pks, g1s(dsh), u, v1, off1, w).all()
    assert engine.verify_decryption_share(pks, g1s(dsh_bad), u, v1, off1, w).tolist() == [1, 0, 1, 1, 1]
    # an undecodable U fails both checks even when the remaining operands are identities (the pair that
    # would carry the hash must not be skipped)
    u_bad = u.copy()
    u_bad[3, 95] ^= 1
    inf1, inf2 = g1s([None] * N), g2s([None] * N)
    assert engine.verify_decryption_share(inf1, inf1, u_bad, v1, off1, inf2).tolist() == [1, 1, 1, 0, 1]
    assert engine.verify_decryption_share(pks, g1s(dsh), u_bad, v1, off1, w).tolist() == [1, 1, 1, 0, 1]
    assert engine.ciphertext_verify(u_bad, v1, off1, w).tolist() == [1, 1, 1, 0, 1]


def test_compress(engine, rnd):
    """to_bytes (src/lib.rs:149-153, 255-259)."""
    p1 = [o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R)) for _ in range(5)] + [None]
    p2 = [o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R)) for _ in range(5)] + [None]
    c1, st1 = engine.g1_compress(g1s(p1))
    c2, st2 = engine.g2_compress(g2s(p2))
    assert not st1.any() and not st2.any()
    assert [bytes(x) for x in c1] == [o.g1_compressed(p) for p in p1]
    assert [bytes(x) for x in c2] == [o.g2_compressed(p) for p in p2]


def test_device_resident_io(engine, rnd):
    """Device-pointer mode (torch CUDA tensors own the memory): same bytes as host mode."""
    import torch
    S, B = 2, 64
    sks = frs([rnd.randrange(o.R) for _ in range(S)])
    pts = g2s([o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R)) for _ in range(4)] * 16)
    host, _ = engine.g2_mul(sks, pts)
    dev, st = engine.g2_mul(torch.from_numpy(sks).cuda(), torch.from_numpy(pts).cuda())
    torch.cuda.synchronize()
    engine.sync()
    assert (dev.cpu().numpy() == host).all() and not st.cpu().numpy().any()


def test_checked_decompress(engine, rnd):
    """from_bytes (src/lib.rs:140-146, 246-252) in batch: valid, infinity, off-curve, out-of-subgroup."""
    import c_oracle
    g1 = [o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R)) for _ in range(66)] + [None]
    g2 = [o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R)) for _ in range(66)] + [None]
    c1 = np.stack([u8(o.g1_compressed(p)) for p in g1])
    c2 = np.stack([u8(o.g2_compressed(p)) for p in g2])
    while True:
        P0 = o.g2_get_point_from_x((rnd.randrange(o.Q), rnd.randrange(o.Q)), False)
        if P0 is not None:
            break
    c2[3] = u8(o.g2_compressed(P0))        # on the twist, outside G2
    c2[4, 0] &= 0x7F                        # compression flag missing
    c1[5, 47] ^= 1                          # off the curve or outside G1 (checked against Oracle B)
    out1, st1 = engine.g1_decompress(c1)
    out2, st2 = engine.g2_decompress(c2)
    want1 = [0] * 67
    want1[5] = c_oracle.g1_decompress(bytes(c1[5]))[0]
    assert st1.tolist() == want1 and st1[5] == 3
    assert st2.tolist() == [0, 0, 0, 3, 3] + [0] * 62
    for j in (0, 1, 2, 6, 40, 66):
        assert bytes(out1[j]) == o.g1_uncompressed(g1[j]) and bytes(out2[j]) == o.g2_uncompressed(g2[j])


def test_config5_shape_t67_n200(engine, rnd):
    """BASELINE config 5 shape (t=67, N=200) at a small batch: 68-point combination through the
    chunked general path, against Oracle B; sign -> combine -> verify round trip."""
    import c_oracle
    t, N, B = 67, 200, 6
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    sks = [o.secret_key_share(poly, i) for i in range(N)]
    msgs = [b"tc/msg" + j.to_bytes(8, "little") for j in range(B)]
    flat, off = pack_messages(msgs)
    hashes = engine.hash_g2(flat, off)
    idx = np.stack([np.array(sorted(rnd.sample(range(N), t + 1)), dtype=np.uint64) for _ in range(B)])
    shares = np.zeros((B, t + 1, 192), dtype=np.uint8)
    for j in range(B):
        sel = frs([sks[int(i)] for i in idx[j]])
        out, st = engine.g2_mul(sel, hashes[j:j + 1].copy())
        assert not st.any()
        shares[j] = out[0]
    sig, st = engine.combine_g2(t, idx, shares)
    assert not st.any()
    for j in (0, B - 1):
        rc, want = c_oracle.combine_g2(t, [int(i) for i in idx[j]], [bytes(shares[j, k]) for k in range(t + 1)])
        assert rc == 0 and bytes(sig[j]) == want
    pk = u8(o.g1_uncompressed(o.public_key(poly[0])))
    assert engine.verify_g2(pk, sig, hashes).all()
    assert engine.verify_sig(pk, sig, flat, off).all()


def test_empty_batches_and_long_messages(engine, rnd):
    """B = 0 is a no-op for every entry point; messages far beyond one SHA3 block hash correctly."""
    z8 = np.zeros((0, 192), dtype=np.uint8)
    out, st = engine.combine_g2(3, np.zeros((0, 4), dtype=np.uint64), np.zeros((0, 4, 192), dtype=np.uint8))
    assert out.shape == (0, 192) and st.shape == (0,)
    out, st = engine.g2_mul(frs([5]), z8)
    assert out.shape == (0, 1, 192)
    assert engine.pairing_check(np.zeros((0, 96), np.uint8), z8, np.zeros((0, 96), np.uint8), z8, B=0).shape == (0,)
    flat, off = pack_messages([])
    assert engine.hash_g2(flat, off).shape == (0, 192)
    # the same with device-resident operands: an empty torch tensor has NO address (data_ptr() == 0) -- still a no-op, not an argument error
    import torch

    def dev(shape, dt=torch.uint8):
        return torch.empty(shape, dtype=dt, device="cuda")
    zoff = torch.zeros(1, dtype=torch.int64, device="cuda")
    assert engine.combine_g2(3, dev((0, 4), torch.int64), dev((0, 4, 192)))[0].shape == (0, 192)
    assert engine.combine_g1(3, dev((0, 4), torch.int64), dev((0, 4, 96)))[0].shape == (0, 96)
    assert engine.g2_mul(dev((1, 32)), dev((0, 192)))[0].shape == (0, 1, 192)
    assert engine.verify_g2(dev((96,)), dev((0, 192)), dev((0, 192))).shape == (0,)
    assert engine.hash_g2(dev((0,)), zoff).shape == (0, 192)
    assert engine.g2_compress(dev((0, 192)))[0].shape == (0, 96) and engine.g1_decompress(dev((0, 48)))[0].shape == (0, 96)
    assert engine.decrypt(3, dev((0, 4), torch.int64), dev((0, 4, 96)), dev((0,)), zoff)[0].shape == (0,)
    assert engine.ciphertext_verify(dev((0, 96)), dev((0,)), zoff, dev((0, 192))).shape == (0,)
    assert engine.xor_with_hash(dev((0, 96)), dev((0,)), zoff)[0].shape == (0,)
    engine.sync()
    # B > 0 jobs whose messages are ALL empty: no message / plaintext / keystream bytes at all, so the device-resident blob, `v`
    # and plaintext buffers have no address -- encrypt(b""), Ciphertext::verify, SecretKey::decrypt and the threshold decryption
    # still work and agree with the host-buffer run and the oracle
    sk = rnd.randrange(1, o.R)
    pk = g1s([o.E1.mul(o.G1_GEN, sk)])[0]
    r = np.stack([u8(o.fr_to_bytes(rnd.randrange(1, o.R))) for _ in range(3)])
    flat0, off0 = pack_messages([b"", b"", b""])
    u_h, v_h, w_h, st_h = engine.encrypt(pk, r, flat0, off0)
    assert not st_h.any() and v_h.shape[0] in (0, 1) and engine.ciphertext_verify(u_h, v_h, off0, w_h).all()
    want_ct = o.encrypt_with_r(o.E1.mul(o.G1_GEN, sk), int.from_bytes(bytes(r[0]), "little"), b"")
    assert bytes(u_h[0]) == o.g1_uncompressed(want_ct[0]) and bytes(w_h[0]) == o.g2_uncompressed(want_ct[2])
    d_off0 = torch.from_numpy(off0.astype(np.int64)).cuda()
    u_d, v_d, w_d, st_d = engine.encrypt(torch.from_numpy(pk).cuda(), torch.from_numpy(r).cuda(), dev((0,)), d_off0)
    engine.sync()
    assert (u_d.cpu().numpy() == u_h).all() and (w_d.cpu().numpy() == w_h).all() and not st_d.cpu().numpy().any()
    fr_sk = torch.from_numpy(u8(o.fr_to_bytes(sk))).cuda()
    plain_d, ok_d = engine.secret_key_decrypt(fr_sk, u_d, dev((0,)), d_off0, w_d)
    x_d, stx = engine.xor_with_hash(u_d, dev((0,)), d_off0)
    engine.sync()
    assert ok_d.cpu().numpy().all() and plain_d.shape == (0,) and not stx.cpu().numpy().any()
    msgs = [bytes(rnd.randrange(256) for _ in range(n)) for n in (4096, 10000, 136 * 7, 136 * 7 + 1)]
    flat, off = pack_messages(msgs)
    out = engine.hash_g2(flat, off)
    for j, m in enumerate(msgs):
        assert bytes(out[j]) == o.g2_uncompressed(o.hash_g2(m)), len(m)
    # hash_g1_g2 with a long v (> 64 bytes is pre-hashed, src/lib.rs:700-704)
    P = o.E1.mul(o.G1_GEN, 31337)
    h, st = engine.hash_g1_g2(g1s([P] * 2), *pack_messages([msgs[0], msgs[1]]))
    assert not st.any() and bytes(h[1]) == o.g2_uncompressed(o.hash_g1_g2(P, msgs[1]))


def test_not_enough_shares_in_device_mode(engine, rnd):
    """n_per_job <= t in device-resident mode: status NotEnoughShares for every job, no kernel launched."""
    import torch
    idx = torch.zeros((5, 2), dtype=torch.int64, device="cuda")
    sh = torch.zeros((5, 2, 192), dtype=torch.uint8, device="cuda")
    out, st = engine.combine_g2(3, idx, sh)
    engine.sync()
    assert st.cpu().tolist() == [1] * 5


def test_heterogeneous_waves_every_job_checked(engine, rnd):
    """Waves whose lanes take different paths (message lengths across the SHA3 rate and the 64-byte
    switch, rejection-sampling attempts, valid / invalid / infinity operands, fast and general
    combine paths side by side) -- EVERY job compared with the C oracle.  Guards the lane-pair
    kernels against divergence-dependent corruption (profiles/r01_d_pair_notes.md)."""
    import c_oracle
    c_oracle.load()
    # hash_g2: 130 messages, lengths 0..289
    msgs = [bytes(rnd.randrange(256) for _ in range((j * 37) % 290)) for j in range(130)]
    flat, off = pack_messages(msgs)
    out = engine.hash_g2(flat, off)
    for j, m in enumerate(msgs):
        assert bytes(out[j]) == c_oracle.hash_g2(m), j
    # hash_g1_g2: valid, off-curve and infinity G1 operands, lengths 0..130
    B = 70
    g1 = np.zeros((B, 96), np.uint8)
    msgs = []
    for j in range(B):
        P = o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R))
        enc = bytearray(o.g1_uncompressed(None if j % 11 == 5 else P))
        if j % 7 == 3:
            enc[95] ^= 1  # leaves the curve
        g1[j] = u8(enc)
        msgs.append(bytes(rnd.randrange(256) for _ in range((j * 13) % 131)))
    flat, off = pack_messages(msgs)
    out, st = engine.hash_g1_g2(g1, flat, off)
    for j in range(B):
        rc, want = c_oracle.hash_g1_g2(bytes(g1[j]), msgs[j])
        assert (st[j] != 0) == (rc != 0), j
        if rc == 0:
            assert bytes(out[j]) == want, j
    x, stx = engine.xor_with_hash(g1, flat, off)
    for j in range(B):
        rc, want = c_oracle.xor_with_hash(bytes(g1[j]), msgs[j])
        assert (stx[j] != 0) == (rc != 0), j
        if rc == 0:
            assert bytes(x[int(off[j]): int(off[j + 1])]) == want, j
    # combine_g2, t = 3: fast path, large indices, duplicate indices and an invalid share in one wave
    t, N = 3, 12
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    H = o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R))
    idx = np.zeros((B, t + 1), np.uint64)
    shares = np.zeros((B, t + 1, 192), np.uint8)
    cache = {}
    for j in range(B):
        ids = sorted(rnd.sample(range(N), t + 1))
        if j % 5 == 0:
            ids = [i + (1 << 20) for i in ids]
        elif j % 5 == 1:
            ids = [ids[0], ids[0], ids[2], ids[3]]
        idx[j] = ids
        for k, i in enumerate(ids):
            if i not in cache:
                cache[i] = u8(o.g2_uncompressed(o.E2.mul(H, o.poly_evaluate(poly, (i + 1) % o.R))))
            shares[j, k] = cache[i]
        if j % 9 == 4:
            shares[j, 2, 191] ^= 1
    got, st = engine.combine_g2(t, idx, shares)
    for j in range(B):
        rc, want = c_oracle.combine_g2(t, [int(i) for i in idx[j]], [bytes(shares[j, k]) for k in range(t + 1)])
        assert (st[j] != 0) == (rc != 0), (j, st[j], rc)
        if rc == 0:
            assert bytes(got[j]) == want, j
    # the same mix in G1 (threshold decryption's combiner)
    U = o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R))
    sh1 = np.zeros((B, t + 1, 96), np.uint8)
    cache = {}
    for j in range(B):
        for k, i in enumerate(int(i) for i in idx[j]):
            if i not in cache:
                cache[i] = u8(o.g1_uncompressed(o.E1.mul(U, o.poly_evaluate(poly, (i + 1) % o.R))))
            sh1[j, k] = cache[i]
    got1, st1 = engine.combine_g1(t, idx, sh1)
    for j in range(B):
        rc, want = c_oracle.combine_g1(t, [int(i) for i in idx[j]], [bytes(sh1[j, k]) for k in range(t + 1)])
        assert (st1[j] != 0) == (rc != 0), j
        if rc == 0:
            assert bytes(got1[j]) == want, j
    # pairing checks: true / false / infinity / invalid encodings interleaved
    a_ = np.zeros((B, 96), np.uint8); b_ = np.zeros((B, 192), np.uint8)
    c_ = np.zeros((B, 96), np.uint8); d_ = np.zeros((B, 192), np.uint8)
    for j in range(B):
        xs, ys = rnd.randrange(1, o.R), rnd.randrange(1, o.R)
        a_[j] = u8(o.g1_uncompressed(None if j % 13 == 6 else o.E1.mul(o.G1_GEN, xs)))
        b_[j] = u8(o.g2_uncompressed(o.E2.mul(o.G2_GEN, ys)))
        c_[j] = u8(o.g1_uncompressed(o.G1_GEN))
        d_[j] = u8(o.g2_uncompressed(None if j % 17 == 8 else o.E2.mul(o.G2_GEN, (xs * ys + (j % 3 == 1)) % o.R)))
        if j % 19 == 9:
            d_[j, 100] ^= 4
    ok = engine.pairing_check(a_, b_, c_, d_)
    for j in range(B):
        assert int(ok[j]) == int(c_oracle.pairing_check(bytes(a_[j]), bytes(b_[j]), bytes(c_[j]), bytes(d_[j])) == 1), j


def test_combine_grouped_by_denominator_class_matches_ungrouped_and_oracle(engine, rnd):
    """Batches of >= 4096 G2 jobs are regrouped by the class of their Lagrange denominator (k_combine.hip):
    same outputs as the same jobs run in small, ungrouped batches; sampled jobs against the C oracle; large
    indices (general path), a flagged job and an invalid share ride along."""
    import c_oracle
    c_oracle.load()
    t, N, B = 3, 10, 4096 + 77
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    sks = [o.secret_key_share(poly, i) for i in range(N)]
    ks = [rnd.randrange(1, o.R) for _ in range(B)]
    G2U = np.frombuffer(o.g2_uncompressed(o.G2_GEN), np.uint8)
    pts, _ = engine.g2_mul(frs(ks), G2U[None].copy())          # (1, B, 192)?  -> one point, B scalars
    pts = np.ascontiguousarray(pts.reshape(B, 192))
    shares_all, st = engine.g2_mul(frs(sks), pts)               # (B, N, 192)
    assert not st.any()
    idx = np.zeros((B, t + 1), np.uint64)
    shares = np.zeros((B, t + 1, 192), np.uint8)
    for j in range(B):
        ids = sorted(rnd.sample(range(N), t + 1))
        idx[j] = ids
        shares[j] = shares_all[j, ids]
    idx[5] += np.uint64(1 << 40)            # large indices: general path (wrong interpolation points, still deterministic)
    idx[9, 1] = idx[9, 0]                   # duplicate index quirk (src/lib.rs:758)
    shares[13, 2, 191] ^= 1                 # leaves the curve
    got, gst = engine.combine_g2(t, idx, shares)
    ref = np.zeros_like(got); rst = np.zeros_like(gst)
    for lo in range(0, B, 1024):
        ref[lo:lo + 1024], rst[lo:lo + 1024] = engine.combine_g2(t, idx[lo:lo + 1024].copy(), shares[lo:lo + 1024].copy())
    assert (gst == rst).all() and gst[13] == 3 and (got == ref).all()
    for j in [0, 5, 9, 13, B - 1] + rnd.sample(range(B), 40):
        rc, want = c_oracle.combine_g2(t, [int(x) for x in idx[j]], [bytes(shares[j, k]) for k in range(t + 1)])
        assert (gst[j] != 0) == (rc != 0), j
        if rc == 0:
            assert bytes(got[j]) == want, j


def test_lincomb_with_infinity_points_zero_scalars_and_ragged_chunks(engine, rnd):
    """sum_i s_i P_i (Commitment::evaluate's shape, src/poly.rs:497-508) for n = 1..6 points, with the
    identity among the points, zero scalars and repeated points: the affine window tables of the
    Straus ladder (one shared inversion) must cope with entries at infinity."""
    for grp, E, gen, enc, fn, nb in (("g2", o.E2, o.G2_GEN, o.g2_uncompressed, engine.lincomb_g2, 192),
                                     ("g1", o.E1, o.G1_GEN, o.g1_uncompressed, engine.lincomb_g1, 96)):
        for n in (1, 2, 3, 4, 5, 6):
            B = 9
            pts, scs, want = np.zeros((B, n, nb), np.uint8), np.zeros((B, n, 32), np.uint8), []
            for j in range(B):
                P = [E.mul(gen, rnd.randrange(1, o.R)) for _ in range(n)]
                s = [rnd.randrange(o.R) for _ in range(n)]
                if j % 3 == 1:
                    P[rnd.randrange(n)] = None
                if j % 3 == 2:
                    s[rnd.randrange(n)] = 0
                if j == 4 and n >= 2:
                    P[1] = P[0]
                if j == 5 and n >= 2:
                    P[1] = E.neg(P[0]) if hasattr(E, "neg") else E.mul(P[0], o.R - 1)
                    s[1] = s[0]
                if j == 6:
                    P = [None] * n
                acc = None
                for Pi, si in zip(P, s):
                    acc = E.add(acc, E.mul(Pi, si))
                want.append(enc(acc))
                for i in range(n):
                    pts[j, i] = u8(enc(P[i]))
                    scs[j, i] = u8(o.fr_to_bytes(s[i]))
            out, st = fn(scs, pts)
            assert not st.any()
            for j in range(B):
                assert bytes(out[j]) == want[j], (grp, n, j)
    # scalar multiplication of the identity and by 0 / 1 / r - 1 through the GLS / GLV tables
    Q = o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R))
    out, st = engine.g2_mul(frs([0, 1, o.R - 1, 5]), g2s([Q, None]))
    assert not st.any()
    for si, s in enumerate([0, 1, o.R - 1, 5]):
        assert bytes(out[0, si]) == o.g2_uncompressed(o.E2.mul(Q, s)) and bytes(out[1, si]) == o.g2_uncompressed(None)


def test_randomised_soak_short():
    """tests/soak.py for a few seconds: random shapes, index patterns, message lengths and
    operand validity, every job against the C oracle (the long run is `python tests/soak.py 600`)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "soak.py"), "10", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SOAK-OK" in r.stdout, (r.stdout[-500:], r.stderr[-500:])
