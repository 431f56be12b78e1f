#!/usr/bin/env python3
"""Generates tests/golden/vectors.json from Oracle A (oracle/tc_oracle.py).

Fixtures are DATA (inputs and expected outputs, hex).  "source": "self-oracle": the reference's
own tests hold no BLS12-381 known-answer vectors and the Rust crate cannot run
here, so these vectors pin Oracle B and the HIP kernels to Oracle A, and pin all three against
regressions.  The H-spec-dependent entries (hash_g2, hash_g1_g2, xor_with_hash, sign, ciphertext)
are marked "compat-unverified"; everything else is mathematically determined.
Run:  python tools/gen_golden.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tc_oracle as o  # noqa: E402

rnd = random.Random(0x7C5EED)
hx = lambda b: bytes(b).hex()


def main():
    v = {"source": "self-oracle (oracle/tc_oracle.py, pure-Python big-int)", "seed": "0x7C5EED"}
    # --- anchors ---------------------------------------------------------------------------
    v["generators"] = {"g1_compressed": hx(o.g1_compressed(o.G1_GEN)), "g2_compressed": hx(o.g2_compressed(o.G2_GEN)),
                       "g1_uncompressed": hx(o.g1_uncompressed(o.G1_GEN)),
                       "g2_uncompressed": hx(o.g2_uncompressed(o.G2_GEN))}
    # --- scalar multiplication (determined) -------------------------------------------------
    muls = []
    for k in [0, 1, 2, o.R - 1, rnd.randrange(o.R), rnd.randrange(o.R)]:
        p1 = o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R))
        p2 = o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R))
        muls.append({"fr": hx(o.fr_to_bytes(k)), "g1": hx(o.g1_uncompressed(p1)), "g1_out": hx(o.g1_uncompressed(o.E1.mul(p1, k))),
                     "g2": hx(o.g2_uncompressed(p2)), "g2_out": hx(o.g2_uncompressed(o.E2.mul(p2, k))),
                     "g1_out_compressed": hx(o.g1_compressed(o.E1.mul(p1, k))),
                     "g2_out_compressed": hx(o.g2_compressed(o.E2.mul(p2, k)))})
    v["mul"] = muls
    # --- Lagrange + combine (determined) ----------------------------------------------------
    combos = []
    for t, ids in [(0, [4]), (1, [0, 1]), (3, [5, 7, 8, 10]), (3, [42, 43, 44, 45]), (2, [1, 1, 4]), (5, [0, 2, 3, 6, 9, 11]),
                   (2, [0, 2, 3, 5, 6])]:
        poly = [rnd.randrange(o.R) for _ in range(t + 1)]
        h = o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R))
        u = o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R))
        s2 = [o.E2.mul(h, o.secret_key_share(poly, i)) for i in ids]
        s1 = [o.E1.mul(u, o.secret_key_share(poly, i)) for i in ids]
        c2 = o.interpolate(o.E2, t, list(zip(ids, s2)))
        c1 = o.interpolate(o.E1, t, list(zip(ids, s1)))
        xs = [o.into_fr_plus_1(i) for i in ids[: t + 1]]
        combos.append({"t": t, "idx": ids, "lagrange": [hx(o.fr_to_bytes(l)) for l in o.lagrange_coeffs(t, xs)] if t else [],
                       "shares_g2": [hx(o.g2_uncompressed(s)) for s in s2], "combined_g2": hx(o.g2_uncompressed(c2)),
                       "shares_g1": [hx(o.g1_uncompressed(s)) for s in s1], "combined_g1": hx(o.g1_uncompressed(c1)),
                       "equals_master": len(set(ids[: t + 1])) == t + 1 and c2 == o.E2.mul(h, poly[0])})
    v["combine"] = combos
    # --- pairing booleans (determined) ------------------------------------------------------
    pairs = []
    for good in (True, False, True):
        a, b = rnd.randrange(1, o.R), rnd.randrange(1, o.R)
        A, Bq = o.E1.mul(o.G1_GEN, a), o.E2.mul(o.G2_GEN, b)
        D = o.E2.mul(o.G2_GEN, (a * b + (0 if good else 7)) % o.R)
        assert o.pairing_check(A, Bq, o.G1_GEN, D) == good
        pairs.append({"a": hx(o.g1_uncompressed(A)), "b": hx(o.g2_uncompressed(Bq)), "c": hx(o.g1_uncompressed(o.G1_GEN)),
                      "d": hx(o.g2_uncompressed(D)), "equal": good})
    pairs.append({"a": hx(o.g1_uncompressed(None)), "b": hx(o.g2_uncompressed(o.G2_GEN)), "c": hx(o.g1_uncompressed(o.G1_GEN)),
                  "d": hx(o.g2_uncompressed(None)), "equal": True})
    v["pairing_check"] = pairs
    gt = o.pairing(o.G1_GEN, o.G2_GEN)
    v["pairing_gt_generators"] = "".join(x.to_bytes(48, "big").hex() for f6 in gt for f2 in f6 for x in f2)
    # --- H-spec dependent (compat-unverified) ------------------------------------------------
    msgs = [b"", b"a", b"Test message", bytes(range(256)), bytes(136), b"tc/msg" + (0).to_bytes(8, "little")]
    v["hash_g2"] = [{"msg": hx(m), "out": hx(o.g2_uncompressed(o.hash_g2(m))), "status": "compat-unverified"} for m in msgs]
    seed = bytes(range(32))
    rng = o.ChaChaRng(seed)
    v["chacha20"] = {"seed": hx(seed), "words": [rng.next_u32() for _ in range(40)],
                     "zero_key_first_words": o.chacha20_block((0,) * 8, 0)[:4]}
    sk = rnd.randrange(o.R)
    pk = o.public_key(sk)
    v["sign"] = [{"sk": hx(o.fr_to_bytes(sk)), "pk": hx(o.g1_uncompressed(pk)), "msg": hx(m),
                  "sig": hx(o.g2_uncompressed(o.sign(sk, m))), "status": "compat-unverified"} for m in msgs[:3]]
    p = o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R))
    v["hash_g1_g2"] = [{"g1": hx(o.g1_uncompressed(p)), "msg": hx(m), "out": hx(o.g2_uncompressed(o.hash_g1_g2(p, m))),
                        "status": "compat-unverified"} for m in (bytes(range(10)), bytes(range(64)), bytes(range(65)))]
    v["xor_with_hash"] = [{"g1": hx(o.g1_uncompressed(p)), "data": hx(bytes(range(100))),
                           "out": hx(o.xor_with_hash(p, bytes(range(100)))), "status": "compat-unverified"}]
    # threshold encryption round trip
    t, ids = 2, [0, 2, 4]
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    commit = o.commitment(poly)
    ct = o.encrypt_with_r(commit[0], rnd.randrange(1, o.R), b"Totally real news")
    dsh = [o.decrypt_share_no_verify(o.secret_key_share(poly, i), ct) for i in ids]
    v["threshold_enc"] = {"t": t, "idx": ids, "commit": [hx(o.g1_uncompressed(c)) for c in commit],
                          "pk_shares": [hx(o.g1_uncompressed(o.public_key_share(commit, i))) for i in ids],
                          "u": hx(o.g1_uncompressed(ct[0])), "v": hx(ct[1]), "w": hx(o.g2_uncompressed(ct[2])),
                          "dec_shares": [hx(o.g1_uncompressed(d)) for d in dsh], "plaintext": hx(b"Totally real news"),
                          "valid": o.ciphertext_verify(ct), "status": "compat-unverified"}
    assert o.threshold_decrypt(t, list(zip(ids, dsh)), ct) == b"Totally real news"
    out = os.path.join(ROOT, "tests", "golden", "vectors.json")
    with open(out, "w") as f:
        json.dump(v, f, indent=1)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
