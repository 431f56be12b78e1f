"""Randomised differential soak test on the GPU: random batch shapes, signer subsets, message lengths,
valid / invalid / identity operands -- EVERY job compared with the C oracle (oracle/c, test
infrastructure).  Usage on an MI355X:  python tests/soak.py [seconds] [seed]
Exits non-zero on the first mismatch and prints the reproducer (seed, round, entry point, job)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import c_oracle  # noqa: E402
import tc_oracle as o  # noqa: E402
from threshold_crypto_amd.engine import Engine, pack_messages  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(SEED)
c_oracle.load()
# two contexts: the library's own form thresholds, and the TWO-jobs-per-lane-pair forms forced at every size (TC_DUO_MIN is read
# once, when a context is created: csrc/tc_launch.h Tuning)
e_default = Engine(0)
os.environ["TC_DUO_MIN"] = "1"
e_duo = Engine(0)
os.environ.pop("TC_DUO_MIN")
assert e_duo.tuning()["duo_min_hash"] == 1 and e_default.tuning()["duo_min_hash"] == 131072
e = e_default


class DeviceResident:
    """The same engine with DEVICE-resident operands (r06): every numpy argument of a call goes up as a torch tensor first, the
    call runs in device-I/O mode (nothing crosses PCIe inside it, results are written where torch allocated them), and the results
    come back as numpy arrays -- so that every entry the soak drives is also compared with the oracle in the residency the bench
    and a GPU-resident caller use.  Scalars, None and Python ints pass through."""

    def __init__(self, eng):
        self._eng = eng

    def __getattr__(self, name):
        import torch
        fn = getattr(self._eng, name)
        if not callable(fn) or name in ("tuning", "sync", "trim", "close", "version", "set_input_checks", "input_checks"):
            return fn

        def up(a):
            if isinstance(a, np.ndarray):
                return torch.from_numpy(np.ascontiguousarray(a).view(np.int64) if a.dtype == np.uint64 else np.ascontiguousarray(a)).cuda()
            return a

        def down(r):
            if type(r).__module__.startswith("torch"):
                return r.cpu().numpy().view(np.uint64) if r.dtype == torch.int64 else r.cpu().numpy()
            if isinstance(r, tuple):
                return tuple(down(x) for x in r)
            return r

        def call(*args, **kw):
            if not any(isinstance(a, np.ndarray) for a in list(args) + list(kw.values())):
                return fn(*args, **kw)
            d_args = [up(a) for a in args]
            d_kw = {k: up(v) for k, v in kw.items()}
            torch.cuda.synchronize()        # (the copies above ran on torch's stream, the library runs on the context's)
            r = fn(*d_args, **d_kw)
            self._eng.sync()
            return down(r)
        return call


dev_rounds = 0


def u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def rbytes(n):
    return bytes(rnd.randrange(256) for _ in range(n))


def fr(k):
    return o.fr_to_bytes(k)


# pools of valid points made with the C oracle (fast), so that batches are cheap to assemble
G1U, G2U = o.g1_uncompressed(o.G1_GEN), o.g2_uncompressed(o.G2_GEN)
pool1 = [c_oracle.g1_mul(fr(rnd.randrange(1, o.R)), G1U)[1] for _ in range(24)] + [o.g1_uncompressed(None)]
pool2 = [c_oracle.g2_mul(fr(rnd.randrange(1, o.R)), G2U)[1] for _ in range(24)] + [o.g2_uncompressed(None)]


def maybe_corrupt(enc):
    if rnd.random() < 0.06:
        b = bytearray(enc)
        b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        return bytes(b)
    return enc


def fail(what, rnd_no, j, extra=""):
    print("MISMATCH seed=%d round=%d %s job=%d %s" % (SEED, rnd_no, what, j, extra), flush=True)
    sys.exit(1)


checked = 0
rounds = 0
t_end = time.time() + SECONDS
duo_rounds = 0
while time.time() < t_end:
    rounds += 1
    B = rnd.choice([1, 2, 31, 32, 33, 63, 64, 65, 97, 130])
    # r05: every other round runs the TWO-jobs-per-lane-pair forms of the checked G2 decode and the G2 hashes (csrc/tc_duo.h), which
    # the library picks from 32 769 / 131 072 jobs on by itself: those rounds go through the context created with TC_DUO_MIN=1
    if rnd.randrange(2):
        e = e_duo
        duo_rounds += 1
    else:
        e = e_default
    # r06: every third round with device-resident operands
    if rnd.randrange(3) == 0:
        e = DeviceResident(e)
        dev_rounds += 1
    # ---- scalar multiplication (S signers x B points) -------------------------------------------
    S = rnd.choice([1, 2, 3])
    ks = [rnd.choice([0, 1, o.R - 1, rnd.randrange(o.R), rnd.randrange(1 << 64)]) for _ in range(S)]
    pts2 = [maybe_corrupt(rnd.choice(pool2)) for _ in range(B)]
    out, st = e.g2_mul(np.stack([u8(fr(k)) for k in ks]), np.stack([u8(p) for p in pts2]))
    for j in range(B):
        for s in range(S):
            rc, want = c_oracle.g2_mul(fr(ks[s]), pts2[j])
            if (st[j, s] != 0) != (rc != 0) or (rc == 0 and bytes(out[j, s]) != want):
                fail("g2_mul", rounds, j)
    pts1 = [maybe_corrupt(rnd.choice(pool1)) for _ in range(B)]
    out, st = e.g1_mul(np.stack([u8(fr(k)) for k in ks]), np.stack([u8(p) for p in pts1]))
    for j in range(B):
        for s in range(S):
            rc, want = c_oracle.g1_mul(fr(ks[s]), pts1[j])
            if (st[j, s] != 0) != (rc != 0) or (rc == 0 and bytes(out[j, s]) != want):
                fail("g1_mul", rounds, j)
    checked += 2 * B * S
    # ---- combine (G2 and G1), mixed index patterns --------------------------------------------------
    t = rnd.choice([1, 2, 3, 3, 3, 5, 7, 8, 12])   # 7+ : the two-stage large-threshold path in G2
    n = t + 1 + rnd.choice([0, 0, 1])
    N = rnd.choice([t + 2, 10, 40, 200])
    idx = np.zeros((B, n), np.uint64)
    sh2 = np.zeros((B, n, 192), np.uint8)
    sh1 = np.zeros((B, n, 96), np.uint8)
    for j in range(B):
        ids = sorted(rnd.sample(range(max(N, n)), n))
        mode = rnd.random()
        if mode < 0.1:
            ids = [i + (1 << rnd.choice([16, 20, 40, 63])) for i in ids]
        elif mode < 0.2:
            ids[1] = ids[0]
        idx[j] = ids
        for k_ in range(n):
            sh2[j, k_] = u8(maybe_corrupt(rnd.choice(pool2)))
            sh1[j, k_] = u8(maybe_corrupt(rnd.choice(pool1)))
        if mode >= 0.1 and mode < 0.2:
            sh2[j, 1] = sh2[j, 0]
            sh1[j, 1] = sh1[j, 0]
    got2, st2 = e.combine_g2(t, idx, sh2)
    got1, st1 = e.combine_g1(t, idx, sh1)
    for j in range(B):
        ids = [int(i) for i in idx[j]]
        rc, want = c_oracle.combine_g2(t, ids, [bytes(sh2[j, k_]) for k_ in range(n)])
        if (st2[j] != 0) != (rc != 0) or (rc == 0 and bytes(got2[j]) != want):
            fail("combine_g2", rounds, j, "t=%d ids=%s st=%d rc=%d" % (t, ids, st2[j], rc))
        rc, want = c_oracle.combine_g1(t, ids, [bytes(sh1[j, k_]) for k_ in range(n)])
        if (st1[j] != 0) != (rc != 0) or (rc == 0 and bytes(got1[j]) != want):
            fail("combine_g1", rounds, j, "t=%d ids=%s st=%d rc=%d" % (t, ids, st1[j], rc))
    checked += 2 * B
    # ---- hashing ---------------------------------------------------------------------------------
    msgs = [rbytes(rnd.choice([0, 1, 14, 63, 64, 65, 135, 136, 137, rnd.randrange(300)])) for _ in range(B)]
    flat, off = pack_messages(msgs)
    h = e.hash_g2(flat, off)
    for j in range(B):
        if bytes(h[j]) != c_oracle.hash_g2(msgs[j]):
            fail("hash_g2", rounds, j, "len=%d" % len(msgs[j]))
    g1 = np.stack([u8(maybe_corrupt(rnd.choice(pool1))) for _ in range(B)])
    hg, st = e.hash_g1_g2(g1, flat, off)
    x, stx = e.xor_with_hash(g1, flat, off)
    for j in range(B):
        rc, want = c_oracle.hash_g1_g2(bytes(g1[j]), msgs[j])
        if (st[j] != 0) != (rc != 0) or (rc == 0 and bytes(hg[j]) != want):
            fail("hash_g1_g2", rounds, j)
        rc, want = c_oracle.xor_with_hash(bytes(g1[j]), msgs[j])
        if (stx[j] != 0) != (rc != 0) or (rc == 0 and bytes(x[int(off[j]): int(off[j + 1])]) != want):
            fail("xor_with_hash", rounds, j)
    checked += 3 * B
    # ---- pairing checks: true / false / identity / invalid ------------------------------------------
    Bp = min(B, 48)
    a_ = np.zeros((Bp, 96), np.uint8); b_ = np.zeros((Bp, 192), np.uint8)
    c_ = np.zeros((Bp, 96), np.uint8); d_ = np.zeros((Bp, 192), np.uint8)
    for j in range(Bp):
        xs, ys = rnd.randrange(1, o.R), rnd.randrange(1, o.R)
        a_[j] = u8(maybe_corrupt(c_oracle.g1_mul(fr(xs), G1U)[1] if rnd.random() > 0.05 else o.g1_uncompressed(None)))
        b_[j] = u8(maybe_corrupt(c_oracle.g2_mul(fr(ys), G2U)[1]))
        c_[j] = u8(G1U)
        good = rnd.random() < 0.6
        d_[j] = u8(maybe_corrupt(c_oracle.g2_mul(fr((xs * ys + (0 if good else 1)) % o.R), G2U)[1] if rnd.random() > 0.05 else o.g2_uncompressed(None)))
    ok = e.pairing_check(a_, b_, c_, d_)
    for j in range(Bp):
        if int(ok[j]) != int(c_oracle.pairing_check(bytes(a_[j]), bytes(b_[j]), bytes(c_[j]), bytes(d_[j])) == 1):
            fail("pairing_check", rounds, j)
    checked += Bp
    # ---- composed entry points (the hash's constant folded into a scalar / the generator side) -----------
    Bc = min(B, 40)
    cm = msgs[:Bc]
    cflat, coff = pack_messages(cm)
    Sc = rnd.choice([1, 2, 5])
    sks = [rnd.choice([0, 1, o.R - 1, rnd.randrange(o.R)]) for _ in range(Sc)]
    sg, sst = e.sign(np.stack([u8(fr(k)) for k in sks]), cflat, coff)
    for j in range(Bc):
        hj = c_oracle.hash_g2(cm[j])
        for s_ in range(Sc):
            rc, want = c_oracle.g2_mul(fr(sks[s_]), hj)
            if sst[j, s_] != 0 or rc != 0 or bytes(sg[j, s_]) != want:
                fail("sign", rounds, j, "s=%d" % s_)
    sk0 = rnd.randrange(1, o.R)
    pk0 = c_oracle.g1_mul(fr(sk0), G1U)[1]
    pks = np.zeros((Bc, 96), np.uint8); sigs = np.zeros((Bc, 192), np.uint8)
    for j in range(Bc):
        good = rnd.random() < 0.6
        pks[j] = u8(maybe_corrupt(pk0 if rnd.random() > 0.05 else o.g1_uncompressed(None)))
        sj = c_oracle.g2_mul(fr((sk0 + (0 if good else 1)) % o.R), c_oracle.hash_g2(cm[j]))[1]
        sigs[j] = u8(maybe_corrupt(sj if rnd.random() > 0.05 else o.g2_uncompressed(None)))
    okv = e.verify_sig(pks, sigs, cflat, coff)
    oks = e.verify_sig(pks[0], sigs, cflat, coff)  # one shared key (stride 0)
    for j in range(Bc):
        hj = c_oracle.hash_g2(cm[j])
        if int(okv[j]) != int(c_oracle.pairing_check(bytes(pks[j]), hj, G1U, bytes(sigs[j])) == 1):
            fail("verify_sig", rounds, j)
        if int(oks[j]) != int(c_oracle.pairing_check(bytes(pks[0]), hj, G1U, bytes(sigs[j])) == 1):
            fail("verify_sig(shared key)", rounds, j)
    us = np.zeros((Bc, 96), np.uint8); ws = np.zeros((Bc, 192), np.uint8)
    shs = np.zeros((Bc, 96), np.uint8); pss = np.zeros((Bc, 96), np.uint8)
    for j in range(Bc):
        rj = rnd.randrange(1, o.R)
        uj = c_oracle.g1_mul(fr(rj), G1U)[1]
        us[j] = u8(maybe_corrupt(uj))
        rc, hj = c_oracle.hash_g1_g2(bytes(us[j]), cm[j])
        good = rnd.random() < 0.6
        wj = c_oracle.g2_mul(fr((rj + (0 if good else 1)) % o.R), hj)[1] if rc == 0 else o.g2_uncompressed(None)
        ws[j] = u8(maybe_corrupt(wj))
        ski = rnd.randrange(1, o.R)
        pss[j] = u8(maybe_corrupt(c_oracle.g1_mul(fr(ski), G1U)[1]))
        shs[j] = u8(maybe_corrupt(c_oracle.g1_mul(fr((ski + (0 if rnd.random() < 0.6 else 1)) % o.R), uj)[1]))
    okc = e.ciphertext_verify(us, cflat, coff, ws)
    okd = e.verify_decryption_share(pss, shs, us, cflat, coff, ws)
    for j in range(Bc):
        rc, hj = c_oracle.hash_g1_g2(bytes(us[j]), cm[j])
        want_c = rc == 0 and c_oracle.pairing_check(G1U, bytes(ws[j]), bytes(us[j]), hj) == 1
        if int(okc[j]) != int(want_c):
            fail("ciphertext_verify", rounds, j)
        want_d = rc == 0 and c_oracle.pairing_check(bytes(shs[j]), hj, bytes(pss[j]), bytes(ws[j])) == 1
        if int(okd[j]) != int(want_d):
            fail("verify_decryption_share", rounds, j)
    checked += Bc * (Sc + 4)
    # ---- round-2 entries: on-device share generation, fixed-base commitments, membership, RLC share validation --
    Bg = min(B, 24)
    Ng, ng = rnd.choice([5, 12, 200]), rnd.choice([1, 3, 4, 5, 9])
    skt = [rnd.choice([0, 1, o.R - 1, rnd.randrange(o.R)]) for _ in range(Ng)]
    gidx = np.array([[rnd.randrange(Ng + (1 if rnd.random() < 0.05 else 0)) for _ in range(ng)] for _ in range(Bg)], dtype=np.uint64)
    hp = [maybe_corrupt(rnd.choice(pool2)) for _ in range(Bg)]
    gs, gst = e.sign_shares_g2(np.stack([u8(fr(k)) for k in skt]), gidx, np.stack([u8(p) for p in hp]))
    for j in range(Bg):
        for k_ in range(ng):
            i_ = int(gidx[j, k_])
            rc, want = c_oracle.g2_mul(fr(skt[i_]), hp[j]) if i_ < Ng else (3, b"")
            if (gst[j, k_] != 0) != (rc != 0) or (rc == 0 and bytes(gs[j, k_]) != want):
                fail("sign_shares_g2", rounds, j, "k=%d idx=%d" % (k_, i_))
    cks = [rnd.choice([0, 1, 8, 9, o.R - 1, rnd.randrange(o.R), rnd.randrange(1 << 64)]) for _ in range(Bg)]
    cm_, cst = e.g1_commitment(np.stack([u8(fr(k)) for k in cks]))
    for j in range(Bg):
        rc, want = c_oracle.g1_mul(fr(cks[j]), G1U)
        if cst[j] != 0 or bytes(cm_[j]) != want:
            fail("g1_commitment", rounds, j)
    mem2 = e.g2_subgroup_check(np.stack([u8(p) for p in hp]))
    for j in range(Bg):
        if int(mem2[j]) != int(c_oracle.g2_mul(fr(1), hp[j])[0] == 0):      # pool points are members; corrupted ones are off the curve
            fail("g2_subgroup_check", rounds, j)
    Nr = rnd.choice([2, 3, 10])
    rsk = [rnd.randrange(1, o.R) for _ in range(Nr)]
    rpk = np.stack([u8(c_oracle.g1_mul(fr(k), G1U)[1]) for k in rsk])
    rm = msgs[:Bg]
    rflat, roff = pack_messages(rm)
    rsig = np.zeros((Bg, Nr, 192), np.uint8)
    rexp = np.ones((Bg, Nr), np.uint8)
    for j in range(Bg):
        hj = c_oracle.hash_g2(rm[j])
        for i_ in range(Nr):
            good = rnd.random() < 0.93
            rsig[j, i_] = u8(c_oracle.g2_mul(fr((rsk[i_] + (0 if good else 1)) % o.R), hj)[1])
            rexp[j, i_] = 1 if good else 0
    rok, nfb = e.verify_shares_rlc(rpk, rsig, rflat, roff, seed=rbytes(32))
    if (rok != rexp).any() or nfb != int((rexp.min(axis=1) == 0).sum()):
        fail("verify_shares_rlc", rounds, int(np.flatnonzero((rok != rexp).any(axis=1))[0]) if (rok != rexp).any() else -1, "nfb=%d" % nfb)
    checked += Bg * (ng + 2 + Nr)
    # ---- round-3 entries: same-key signature batches and decryption shares by random linear combination, G1 linear
    # combinations through the two-stage kernels -------------------------------------------------------------------------------
    Bs = min(B, 40)
    sm = msgs[:Bs]
    sflat, soff = pack_messages(sm)
    ssig = np.zeros((Bs, 192), np.uint8)
    shash = np.zeros((Bs, 192), np.uint8)
    sexp = np.ones(Bs, np.uint8)
    for j in range(Bs):
        hj = c_oracle.hash_g2(sm[j])
        shash[j] = u8(hj)
        r_ = rnd.random()
        good = r_ < 0.9
        enc = c_oracle.g2_mul(fr((sk0 + (0 if good else 1)) % o.R), hj)[1]
        if r_ > 0.97:
            enc, good = maybe_corrupt(enc), None
        ssig[j] = u8(enc)
        sexp[j] = int(c_oracle.pairing_check(pk0, hj, G1U, enc) == 1)
    grp = rnd.choice([0, 4, 7, 64])
    sok, nfb = e.verify_g2_rlc(u8(pk0), ssig, shash, group=grp, seed=rbytes(32))
    if (sok != sexp).any():
        fail("verify_g2_rlc", rounds, int(np.flatnonzero(sok != sexp)[0]), "group=%d" % grp)
    sok, nfb = e.verify_sig_rlc(u8(pk0), ssig, sflat, soff, group=grp, seed=rbytes(32))
    if (sok != sexp).any():
        fail("verify_sig_rlc", rounds, int(np.flatnonzero(sok != sexp)[0]), "group=%d" % grp)
    Bd, Nd = min(B, 12), rnd.choice([1, 3, 10])
    dsk = [rnd.randrange(1, o.R) for _ in range(Nd)]
    dpk = np.stack([u8(c_oracle.g1_mul(fr(k), G1U)[1]) for k in dsk])
    dm = msgs[:Bd]
    dflat, doff = pack_messages(dm)
    du = np.zeros((Bd, 96), np.uint8); dw = np.zeros((Bd, 192), np.uint8); dsh = np.zeros((Bd, Nd, 96), np.uint8)
    dexp = np.zeros((Bd, Nd), np.uint8)
    for j in range(Bd):
        rj = rnd.randrange(1, o.R)
        uj = c_oracle.g1_mul(fr(rj), G1U)[1]
        du[j] = u8(uj)
        rc, hj = c_oracle.hash_g1_g2(uj, dm[j])
        dw[j] = u8(c_oracle.g2_mul(fr((rj + (0 if rnd.random() < 0.9 else 1)) % o.R), hj)[1])
        for i_ in range(Nd):
            dsh[j, i_] = u8(c_oracle.g1_mul(fr((dsk[i_] + (0 if rnd.random() < 0.9 else 1)) % o.R), uj)[1])
            dexp[j, i_] = int(c_oracle.pairing_check(bytes(dsh[j, i_]), hj, bytes(dpk[i_]), bytes(dw[j])) == 1)
    dok, nfb = e.verify_decryption_shares_rlc(dpk, dsh, du, dflat, doff, dw, seed=rbytes(32))
    if (dok != dexp).any():
        fail("verify_decryption_shares_rlc", rounds, int(np.flatnonzero((dok != dexp).any(axis=1))[0]))
    nl = rnd.choice([8, 9, 13, 20])
    lsc = [[rnd.choice([0, 1, 2, o.R - 1, rnd.randrange(o.R), rnd.randrange(1 << 64)]) for _ in range(nl)] for _ in range(Bd)]
    lpt = [[maybe_corrupt(rnd.choice(pool1)) for _ in range(nl)] for _ in range(Bd)]
    lout, lst = e.lincomb_g1(np.stack([np.stack([u8(fr(k)) for k in row]) for row in lsc]), np.stack([np.stack([u8(p) for p in row]) for row in lpt]))
    for j in range(Bd):
        acc, bad_ = None, False
        for k_, p_ in zip(lsc[j], lpt[j]):
            rc, term = c_oracle.g1_mul(fr(k_), p_)
            bad_ = bad_ or rc != 0
            if rc == 0:
                acc = o.E1.add(acc, o.g1_from_uncompressed(term, check=False))
        if (lst[j] != 0) != bad_ or (not bad_ and bytes(lout[j]) != o.g1_uncompressed(acc)):
            fail("lincomb_g1", rounds, j, "n=%d" % nl)
    checked += 2 * Bs + Bd * (Nd + 1)
    # ---- round-4 entries: wire-level combiners, IntoFr abscissae, decrypt_share / SecretKey::decrypt in one call -----------------
    Bw = min(B, 40)
    tw = rnd.choice([1, 2, 3, 3, 7])
    nw = tw + 1 + rnd.choice([0, 1])
    widx = np.zeros((Bw, nw), np.uint64)
    w96 = np.zeros((Bw, nw, 96), np.uint8)
    w48 = np.zeros((Bw, nw, 48), np.uint8)
    for j in range(Bw):
        widx[j] = sorted(rnd.sample(range(max(12, nw)), nw))
        for k_ in range(nw):
            w96[j, k_] = u8(maybe_corrupt(c_oracle.g2_compress(rnd.choice(pool2))[1]))
            w48[j, k_] = u8(maybe_corrupt(c_oracle.g1_compress(rnd.choice(pool1))[1]))
    wout, wst = e.combine_signatures_wire(tw, widx, w96)
    wm = msgs[:Bw]
    wflat, woff = pack_messages(wm)
    dout, dst = e.decrypt_wire(tw, widx, w48, wflat, woff)
    for j in range(Bw):
        ids = [int(i) for i in widx[j]]
        rc, want = c_oracle.combine_signatures_wire(tw, ids, [bytes(x) for x in w96[j]])
        if int(wst[j]) != rc or bytes(wout[j]) != want:
            fail("combine_signatures_wire", rounds, j, "t=%d st=%d rc=%d" % (tw, wst[j], rc))
        rc, want = c_oracle.decrypt_wire(tw, ids, [bytes(x) for x in w48[j]], wm[j])
        if int(dst[j]) != rc or bytes(dout[int(woff[j]): int(woff[j + 1])]) != want:
            fail("decrypt_wire", rounds, j, "t=%d st=%d rc=%d" % (tw, dst[j], rc))
    # IntoFr abscissae: the same shares keyed by Fr values / negative integers must give what the u64 entry gives for the
    # u64 image of small keys, and Oracle A's interpolate (any integer through into_fr_plus_1) for wide ones (G1: cheap in Python)
    Bf, tf = min(B, 4), rnd.choice([1, 2, 3])
    keys = [[rnd.choice([rnd.randrange(o.R), -rnd.randrange(1, 1 << 40), (1 << 64) + rnd.randrange(99), rnd.randrange(50)]) for _ in range(tf + 1)]
            for _ in range(Bf)]
    fpts = [[rnd.choice(pool1) for _ in range(tf + 1)] for _ in range(Bf)]
    fidx = np.stack([np.stack([u8((k_ % o.R).to_bytes(32, "little")) for k_ in row]) for row in keys])
    fout, fst = e.combine_g1_fr(tf, fidx, np.stack([np.stack([u8(p) for p in row]) for row in fpts]))
    for j in range(Bf):
        want = o.interpolate(o.E1, tf, [(k_, o.g1_from_uncompressed(p, check=False)) for k_, p in zip(keys[j], fpts[j])])
        if fst[j] != 0 or bytes(fout[j]) != o.g1_uncompressed(want):
            fail("combine_g1_fr", rounds, j, "keys=%s" % keys[j])
    # decrypt_share / SecretKey::decrypt on the ciphertext set above (valid, wrong-w, corrupted operands) under one key
    dsh_, dok_ = e.decrypt_share(u8(fr(sk0)), us, cflat, coff, ws)
    dpl_, dok2_ = e.secret_key_decrypt(u8(fr(sk0)), us, cflat, coff, ws)
    for j in range(Bc):
        if int(dok_[j]) != int(okc[j]) or int(dok2_[j]) != int(okc[j]):
            fail("decrypt_share ok", rounds, j)
        got = bytes(dpl_[int(coff[j]): int(coff[j + 1])])
        if not okc[j]:
            if bytes(dsh_[j]) != o.g1_uncompressed(None) or got != bytes(len(cm[j])):
                fail("decrypt_share of an invalid ciphertext", rounds, j)
            continue
        rc, g_ = c_oracle.g1_mul(fr(sk0), bytes(us[j]))
        if rc != 0 or bytes(dsh_[j]) != g_ or got != c_oracle.xor_with_hash(g_, cm[j])[1]:
            fail("decrypt_share / secret_key_decrypt", rounds, j)
    checked += 2 * Bw + Bf + 2 * Bc
    # ---- compressed round trip ------------------------------------------------------------------------
    c2, stc = e.g2_compress(np.stack([u8(p) for p in pts2]))
    d2, std = e.g2_decompress(c2)
    for j in range(B):
        rc, want = c_oracle.g2_compress(pts2[j])
        if (stc[j] != 0) != (rc != 0) or (rc == 0 and bytes(c2[j]) != want):
            fail("g2_compress", rounds, j)
        if rc == 0:
            rc2, want2 = c_oracle.g2_decompress(bytes(c2[j]))
            if (std[j] != 0) != (rc2 != 0) or (rc2 == 0 and bytes(d2[j]) != want2):
                fail("g2_decompress", rounds, j)
    checked += 2 * B
print("SOAK-OK seed=%d rounds=%d (two-jobs-per-lane-pair forms in %d of them, device-resident operands in %d) jobs_checked=%d seconds=%.0f" % (SEED, rounds, duo_rounds, dev_rounds, checked, SECONDS), flush=True)
