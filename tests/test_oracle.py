"""Pins the oracles: public anchors, Oracle A (pure Python) vs Oracle B (plain C) differential,
and both against the committed golden vectors (tests/golden/vectors.json, "self-oracle": the
reference holds no BLS12-381 KATs -- SURVEY.md 8c).  Also replays the reference's own property
tests (src/lib.rs:793-1008) against the oracle."""
import hashlib
import json
import os
import random

import pytest

import c_oracle as c
import tc_oracle as o

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))
bx = bytes.fromhex


@pytest.fixture(scope="module")
def rnd():
    return random.Random(20260926)


def test_public_anchors():
    x = -o.BLS_X
    assert o.R == x ** 4 - x ** 2 + 1 and o.Q == (x - 1) ** 2 * o.R // 3 + x
    assert o.H2 == (x ** 8 - 4 * x ** 7 + 5 * x ** 6 - 4 * x ** 4 + 6 * x ** 3 - 4 * x ** 2 - 4 * x + 13) // 9
    assert o.H2.bit_length() == 507 and bin(o.H2).count("1") == 247
    assert o.E1.on_curve(o.G1_GEN) and o.E2.on_curve(o.G2_GEN)
    assert o.E1.mul(o.G1_GEN, o.R) is None and o.E2.mul(o.G2_GEN, o.R) is None
    assert o.g1_compressed(o.G1_GEN).hex() == ("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
                                               "6c55e83ff97a1aeffb3af00adb22c6bb")
    assert o.g2_compressed(o.G2_GEN).hex().startswith("93e02b6052719f607dacd3a088274f65596bd0d09920b61a")
    # ChaCha20 (djb layout) zero key / zero nonce keystream 76b8e0ad a0f13d90 405d6ae5 5386bd28
    assert o.chacha20_block((0,) * 8, 0)[:4] == [0xade0b876, 0x903df1a0, 0xe56a5d40, 0x28bd8653]
    assert o.sha3_256(b"") == hashlib.sha3_256(b"").digest()


def test_pairing_is_bilinear_and_matches_textbook_miller_loop():
    a, b = 0x1234567890abcdef1234, 0xfedcba0987654321
    P, Qp = o.E1.mul(o.G1_GEN, a), o.E2.mul(o.G2_GEN, b)
    e1 = o.pairing(o.G1_GEN, o.G2_GEN)
    assert e1 != o.F12_ONE and o.f12_pow(e1, o.R) == o.F12_ONE
    assert o.pairing(P, Qp) == o.f12_pow(e1, a * b % o.R)
    # fast (line-coefficient) Miller loop vs affine arithmetic on E(Fq12), after the final exponentiation
    assert o.final_exponentiation(o.miller_loop([(P, Qp)])) == o.final_exponentiation(o.miller_loop_textbook(P, Qp))
    # the pairing-0.16 hard-part chain computes the cube of the defining power
    f = o.miller_loop([(P, Qp)])
    d = o.final_exponentiation(f)
    assert o.final_exponentiation_chain(f) == o.f12_mul(o.f12_mul(d, d), d)


def test_golden_anchors_and_mul():
    g = GOLD["generators"]
    assert o.g1_uncompressed(o.G1_GEN).hex() == g["g1_uncompressed"] and o.g2_uncompressed(o.G2_GEN).hex() == g["g2_uncompressed"]
    for m in GOLD["mul"]:
        k = o.fr_from_bytes(bx(m["fr"]))
        p1 = o.g1_from_uncompressed(bx(m["g1"]))
        p2 = o.g2_from_uncompressed(bx(m["g2"]))
        assert o.g1_uncompressed(o.E1.mul(p1, k)).hex() == m["g1_out"]
        assert o.g2_uncompressed(o.E2.mul(p2, k)).hex() == m["g2_out"]
        assert c.g1_mul(bx(m["fr"]), bx(m["g1"])) == (0, bx(m["g1_out"]))
        assert c.g2_mul(bx(m["fr"]), bx(m["g2"])) == (0, bx(m["g2_out"]))
        assert c.g1_compress(bx(m["g1_out"])) == (0, bx(m["g1_out_compressed"]))
        assert c.g2_compress(bx(m["g2_out"])) == (0, bx(m["g2_out_compressed"]))
        # compressed decode round trip (from_bytes, src/lib.rs:140-146, 246-252)
        assert o.g1_from_compressed(bx(m["g1_out_compressed"])) == o.E1.mul(p1, k)
        assert o.g2_from_compressed(bx(m["g2_out_compressed"])) == o.E2.mul(p2, k)


def test_golden_combine_both_oracles():
    for cb in GOLD["combine"]:
        t, ids = cb["t"], cb["idx"]
        s2 = [o.g2_from_uncompressed(bx(s), check=False) for s in cb["shares_g2"]]
        assert o.g2_uncompressed(o.interpolate(o.E2, t, list(zip(ids, s2)))).hex() == cb["combined_g2"]
        assert c.combine_g2(t, ids, [bx(s) for s in cb["shares_g2"]]) == (0, bx(cb["combined_g2"]))
        assert c.combine_g1(t, ids, [bx(s) for s in cb["shares_g1"]]) == (0, bx(cb["combined_g1"]))
        if t:
            rc, lam = c.lagrange(t, ids[: t + 1])
            assert rc == 0 and [o.fr_to_bytes(l).hex() for l in lam] == cb["lagrange"]


def test_golden_pairing_hash_sign_enc_both_oracles():
    for p in GOLD["pairing_check"]:
        assert c.pairing_check(bx(p["a"]), bx(p["b"]), bx(p["c"]), bx(p["d"])) == int(p["equal"])
    rc, gt = c.pairing_gt(o.g1_uncompressed(o.G1_GEN), o.g2_uncompressed(o.G2_GEN))
    assert rc == 0 and gt.hex() == GOLD["pairing_gt_generators"]
    for h in GOLD["hash_g2"]:
        assert c.hash_g2(bx(h["msg"])).hex() == h["out"]
        assert o.g2_uncompressed(o.hash_g2(bx(h["msg"]))).hex() == h["out"]
    ch = GOLD["chacha20"]
    rng = o.ChaChaRng(bx(ch["seed"]))
    assert [rng.next_u32() for _ in range(40)] == ch["words"]
    for s in GOLD["sign"]:
        assert c.sign(bx(s["sk"]), bx(s["msg"])) == (0, bx(s["sig"]))
        assert c.verify(bx(s["pk"]), bx(s["sig"]), bx(s["msg"])) == 1
        assert c.verify(bx(s["pk"]), bx(s["sig"]), bx(s["msg"]) + b"x") == 0
    for h in GOLD["hash_g1_g2"]:
        assert c.hash_g1_g2(bx(h["g1"]), bx(h["msg"])) == (0, bx(h["out"]))
    for x in GOLD["xor_with_hash"]:
        assert c.xor_with_hash(bx(x["g1"]), bx(x["data"])) == (0, bx(x["out"]))
    te = GOLD["threshold_enc"]
    assert c.ciphertext_verify(bx(te["u"]), bx(te["v"]), bx(te["w"])) == 1
    assert c.ciphertext_verify(bx(te["u"]), b"X" + bx(te["v"])[1:], bx(te["w"])) == 0
    for pk, sh in zip(te["pk_shares"], te["dec_shares"]):
        assert c.verify_decryption_share(bx(pk), bx(sh), bx(te["u"]), bx(te["v"]), bx(te["w"])) == 1
    assert c.threshold_decrypt(te["t"], te["idx"], [bx(s) for s in te["dec_shares"]], bx(te["v"])) == (0, bx(te["plaintext"]))


def test_oracle_a_vs_b_random(rnd):
    for _ in range(3):
        k = rnd.randrange(o.R)
        p2 = o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R))
        assert c.g2_mul(o.fr_to_bytes(k), o.g2_uncompressed(p2)) == (0, o.g2_uncompressed(o.E2.mul(p2, k)))
        m = bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 200)))
        assert c.hash_g2(m) == o.g2_uncompressed(o.hash_g2(m))
        assert c.sha3_256(m) == hashlib.sha3_256(m).digest()
    # infinity and invalid encodings
    assert c.g2_mul(o.fr_to_bytes(3), o.g2_uncompressed(None)) == (0, o.g2_uncompressed(None))
    bad = bytearray(o.g2_uncompressed(o.G2_GEN))
    bad[100] ^= 1
    assert c.g2_mul(o.fr_to_bytes(3), bytes(bad))[0] == 3
    assert c.g2_mul(o.R.to_bytes(32, "little"), o.g2_uncompressed(o.G2_GEN))[0] == 3


# ---- the reference's own property tests, replayed on the oracle ------------------------------
def test_ref_test_interpolate(rnd):
    """test_interpolate (src/lib.rs:793-808): random increasing x's, deg 0..4."""
    for deg in range(5):
        comm = [o.E1.mul(o.G1_GEN, rnd.randrange(o.R)) for _ in range(deg + 1)]
        x, items = 0, []
        for _ in range(deg + 1):
            x += rnd.randrange(1, 5)
            items.append((x - 1, o.commitment_evaluate(comm, x)))
        assert o.interpolate(o.E1, deg, items) == o.commitment_evaluate(comm, 0)
        rc, out = c.combine_g1(deg, [i for i, _ in items], [o.g1_uncompressed(p) for _, p in items])
        assert rc == 0 and out == o.g1_uncompressed(comm[0])


def test_ref_test_threshold_sig(rnd):
    """test_threshold_sig (src/lib.rs:822-873) with t = 3."""
    t = 3
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    commit = o.commitment(poly)
    msg = b"Totally real news"
    shares = {i: o.sign(o.secret_key_share(poly, i), msg) for i in (5, 8, 7, 10)}
    for i, s in shares.items():
        pk_i = o.public_key_share(commit, i)
        assert c.verify(o.g1_uncompressed(pk_i), o.g2_uncompressed(s), msg) == 1
    sig = o.combine_signatures(t, sorted(shares.items()))
    assert o.verify(commit[0], sig, msg)
    shares2 = {i: o.sign(o.secret_key_share(poly, i), msg) for i in (42, 43, 44, 45)}
    assert o.combine_signatures(t, sorted(shares2.items())) == sig
    with pytest.raises(o.NotEnoughShares):
        o.combine_signatures(t, sorted(shares.items())[:3])
    assert c.combine_g2(t, [5, 7, 8], [o.g2_uncompressed(shares[i]) for i in (5, 7, 8)])[0] == 1


def test_ref_test_simple_sig_and_enc(rnd):
    """test_simple_sig (src/lib.rs:810-820), test_simple_enc (:875-897), test_xor_with_hash (:972-982)."""
    sk0, sk1 = rnd.randrange(o.R), rnd.randrange(o.R)
    pk0, pk1 = o.public_key(sk0), o.public_key(sk1)
    sig0 = o.sign(sk0, b"Real news")
    assert o.verify(pk0, sig0, b"Real news") and not o.verify(pk1, sig0, b"Real news") and not o.verify(pk0, sig0, b"Fake news")
    ct = o.encrypt_with_r(pk0, rnd.randrange(1, o.R), b"Muffins in the canteen today!")
    assert o.ciphertext_verify(ct) and o.decrypt(sk0, ct) == b"Muffins in the canteen today!"
    assert o.decrypt(sk1, ct) != b"Muffins in the canteen today!"
    fake = (ct[0], b"\x00" + ct[1][1:] if ct[1][0] else b"\x01" + ct[1][1:], ct[2])
    assert not o.ciphertext_verify(fake) and o.decrypt(sk0, fake) is None
    g0 = o.E1.mul(o.G1_GEN, 11)
    g1 = o.E1.mul(o.G1_GEN, 12)
    xor = lambda a, b: bytes(x ^ y for x, y in zip(a, b))
    z = bytes(20)
    assert xor(o.xor_with_hash(g0, z), o.xor_with_hash(g0, b"\x55" * 20)) == b"\x55" * 20
    assert o.xor_with_hash(g0, z) != o.xor_with_hash(g1, z) and len(o.xor_with_hash(g0, bytes(5))) == 5


def test_checked_decompress_both_oracles(rnd):
    """from_bytes (src/lib.rs:140-146, 246-252): valid points round-trip; off-curve x, points outside
    the order-r subgroup and bad flags are FromBytesError::Invalid."""
    for _ in range(3):
        P = o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R))
        Q2 = o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R))
        assert c.g1_decompress(o.g1_compressed(P)) == (0, o.g1_uncompressed(P))
        assert c.g2_decompress(o.g2_compressed(Q2)) == (0, o.g2_uncompressed(Q2))
    assert c.g1_decompress(o.g1_compressed(None)) == (0, o.g1_uncompressed(None))
    assert c.g2_decompress(o.g2_compressed(None)) == (0, o.g2_uncompressed(None))
    while True:  # a point of E'(Fq2) outside G2
        P0 = o.g2_get_point_from_x((rnd.randrange(o.Q), rnd.randrange(o.Q)), False)
        if P0 is not None:
            break
    assert o.E2.mul(P0, o.R) is not None
    assert c.g2_decompress(o.g2_compressed(P0))[0] == 3
    with pytest.raises(o.DecodeError):
        o.g2_from_compressed(o.g2_compressed(P0))
    bad = bytearray(o.g1_compressed(o.G1_GEN))
    bad[0] &= 0x7f
    assert c.g1_decompress(bytes(bad))[0] == 3


def test_threaded_oracle_drivers_equal_the_single_job_functions(rnd):
    """The threaded drivers the large-threshold GPU tests use (or_sign_combine_batch, or_combine_g1_batch,
    or_sign_shares_batch) against Oracle A: sign t + 1 shares and combine = the master key's signature; a signer index outside
    the key table fails its message only."""
    import numpy as np
    u8 = lambda b: np.frombuffer(bytes(b), dtype=np.uint8)
    t, N, B = 3, 7, 6
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    sk = np.stack([u8(o.secret_key_share(poly, i).to_bytes(32, "little")) for i in range(N)])
    H = [o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R)) for _ in range(B)]
    hs = np.stack([u8(o.g2_uncompressed(h)) for h in H])
    idx = np.array([sorted(rnd.sample(range(N), t + 1)) for _ in range(B)], dtype=np.uint64)
    out, rc = c.sign_combine_batch(t, sk, idx, hs, 3)
    assert not rc.any() and all(bytes(out[j]) == o.g2_uncompressed(o.E2.mul(H[j], poly[0])) for j in range(B))
    sh, rc = c.sign_shares_batch(sk, idx, hs, 2)
    assert not rc.any()
    for j in range(B):
        for k in range(t + 1):
            assert bytes(sh[j, k]) == o.g2_uncompressed(o.E2.mul(H[j], o.secret_key_share(poly, int(idx[j, k]))))
    bad = idx.copy()
    bad[1, 2] = N + 3
    _, rc = c.sign_shares_batch(sk, bad, hs, 2)
    assert list(rc) == [0, 3, 0, 0, 0, 0]
    h1 = o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R))
    s1 = np.stack([np.stack([u8(o.g1_uncompressed(o.E1.mul(h1, o.secret_key_share(poly, int(i))))) for i in row]) for row in idx])
    out1, rc = c.combine_g1_batch(t, idx, s1, 4)
    assert not rc.any() and all(bytes(out1[j]) == o.g1_uncompressed(o.E1.mul(h1, poly[0])) for j in range(B))


def test_reference_fixtures_if_present():
    """Pins BOTH oracles to vectors printed by the real threshold_crypto 0.4.0 crate (tools/ref_fixtures)
    when tests/golden/ref_v0.4.0/vectors.hex exists.  It does not in this repository (no Rust toolchain in
    the build image: SURVEY.md 8c), so the H-spec consumers stay "compat-unverified" and this test skips --
    the day the file is committed it becomes the known-answer test of hash_g2 / sign / encrypt."""
    import ref_fixtures as rf
    assert rf.source() == ("self-oracle" if not rf.present() else rf.source())
    if not rf.present():
        pytest.skip("no reference fixtures (tests/golden/ref_v0.4.0/vectors.hex): parity against the Rust crate unpinned")
    v = rf.load()
    hits, verdict = rf.diagnose(v)
    assert hits and hits[0] == 0, verdict      # a mismatch names the H-spec alternative that DOES reproduce the vectors
    for h in v.get("hash_g2", []):
        want = h["out"]
        assert c.g2_compress(c.hash_g2(h["msg"])) == (0, want)
        assert o.g2_compressed(o.hash_g2(h["msg"])) == want
    for s in v.get("sign", []):
        rc, sig = c.sign(rf.fr_le(s["sk_be"]), s["msg"])
        assert rc == 0 and c.g2_compress(sig) == (0, s["sig"])
    for k in v.get("key", []):
        rng = o.ChaChaRng(k["seed"])
        assert o.fr_random(rng) == int.from_bytes(k["sk_be"], "big")
        assert o.g1_compressed(o.public_key(int.from_bytes(k["sk_be"], "big"))) == k["pk"]
    for e in v.get("encrypt", []):
        u, vv, w = rf.split_ciphertext(e["ciphertext_bincode"])
        sk = int.from_bytes(e["sk_be"], "big")
        ct = (o.g1_from_compressed(u), vv, o.g2_from_compressed(w))
        assert o.ciphertext_verify(ct) and o.decrypt(sk, ct) == e["msg"]


def _fabricated_vectors(setting, tmp_path):
    """the records tools/ref_fixtures prints, made by Oracle A under one H-spec alternative (what a reference run would look
    like if THAT were the crate's behaviour)"""
    import ref_fixtures as rf
    o.set_hspec(setting)
    try:
        lines = []
        for m in (b"", b"a", b"Test message", b"tc/msg" + (0).to_bytes(8, "little")):
            lines.append("hash_g2 msg=%s out=%s" % (m.hex(), o.g2_compressed(o.hash_g2(m)).hex()))
        seed = bytes((31 + i) & 0xff for i in range(32))
        rng = o.ChaChaRng(seed)
        sk = o.fr_random(rng)
        pk = o.public_key(sk)
        lines.append("key seed=%s sk_be=%s pk=%s" % (seed.hex(), sk.to_bytes(32, "big").hex(), o.g1_compressed(pk).hex()))
        msg = b"Muffins in the canteen today! Don't tell Bob."
        u, v, w = o.encrypt_with_r(pk, o.fr_random(rng), msg)
        blob = o.g1_compressed(u) + len(v).to_bytes(8, "little") + v + o.g2_compressed(w)
        lines.append("encrypt seed=%s sk_be=%s msg=%s ciphertext_bincode=%s" % (seed.hex(), sk.to_bytes(32, "big").hex(), msg.hex(), blob.hex()))
    finally:
        o.set_hspec(0)
    path = tmp_path / "vectors.hex"
    path.write_text("\n".join(lines) + "\n")
    return str(path)


@pytest.mark.parametrize("setting", [0, 1, 4, 8, 16, 2 | 16])
def test_hspec_mismatch_is_diagnosed(setting, tmp_path, monkeypatch):
    """the day tests/golden/ref_v0.4.0/vectors.hex arrives and a hash_g2 / xor_with_hash vector
    disagrees, the failure must NAME the documented alternative that reproduces it (word order of next_u64, compare vs
    mask order in Fq::random, the source of `greatest`, the keystream's byte source, Montgomery vs canonical draw) -- each a
    one-constant switch in both oracles and in csrc/tc_hash.h.  A vectors.hex fabricated by Oracle A under one
    alternative is diagnosed as exactly that alternative, and Oracle B switched the same way reproduces it too."""
    import ref_fixtures as rf
    monkeypatch.setattr(rf, "PATH", _fabricated_vectors(setting, tmp_path))
    assert rf.present()
    hits, verdict = rf.diagnose()
    assert hits and hits[0] == setting, (setting, hits, verdict)
    if setting:
        assert "TC_HSPEC=%d" % setting in verdict and 0 not in hits
    else:
        assert "parity pinned" in verdict
    v = rf.load()
    c.set_hspec(setting)
    try:
        for h in v["hash_g2"]:
            assert c.g2_compress(c.hash_g2(h["msg"])) == (0, h["out"])
    finally:
        c.set_hspec(0)
