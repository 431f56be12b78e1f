"""The HBM table arena of the G2 ladder kernels (csrc/tc_table.h): slots are borrowed per wave and recycled.

  * a launch with more waves than the arena has slots (8 XCDs x 512) recycles every slot several times --
    all results against a second launch in a different wave order, a sample against Oracle B;
  * two contexts on the same GPU, driven from two host threads at once, each with its own arena;
  * jobs whose ladder meets the point at infinity (identity operand, zero scalar) keep their table entries'
    infinity flags through the arena.
"""
import os
import random
import threading

import numpy as np
import pytest

import c_oracle as c

pytestmark = pytest.mark.gpu


def u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def oracle_mul(fr, pt):
    rc, out = c.g2_mul(bytes(fr), bytes(pt))
    assert rc == 0
    return out


def _fill_hbm(leave):
    """torch tensors that take the device's free memory down to about `leave` bytes: one large block, then 256 MB blocks, then
    whatever is left -- an allocation that fails (the largest free extent shrinks with fragmentation) is retried at half the size."""
    import torch
    torch.cuda.empty_cache()
    hogs = []
    chunk = None
    while True:
        free, _total = torch.cuda.mem_get_info()
        room = free - leave
        if room < (4 << 20):
            break
        if chunk is None:
            chunk = max(room - (2 << 30), 256 << 20)
        try:
            hogs.append(torch.empty(min(chunk, room), dtype=torch.uint8, device="cuda"))
            chunk = min(chunk, 256 << 20)
        except torch.OutOfMemoryError:
            if chunk <= (4 << 20):
                break
            chunk //= 2
    return hogs


@pytest.fixture(scope="module")
def wl(engine):
    from threshold_crypto_amd.workload import ThresholdSigWorkload
    return ThresholdSigWorkload(engine, 3, 10, 4096)


def test_more_waves_than_arena_slots(engine, wl):
    c.load()
    B = 160000  # one scalar per point: 160 000 lane pairs = 5000 waves > 4096 slots
    reps = (B + wl.B - 1) // wl.B
    pts = np.ascontiguousarray(np.tile(wl.hashes, (reps, 1))[:B])
    rng = random.Random(77)
    k = u8(rng.getrandbits(250).to_bytes(32, "little")).reshape(1, 32)
    out, st = engine.g2_mul(k, pts)
    assert not st.any()
    out = out.reshape(B, 192)
    # the same point gives the same bytes wherever in the launch (whichever slot) it was computed
    for r in range(1, reps):
        n = min(wl.B, B - r * wl.B)
        assert (out[r * wl.B:r * wl.B + n] == out[:n]).all()
    for j in rng.sample(range(wl.B), 24):
        assert bytes(out[j]) == oracle_mul(k[0], wl.hashes[j])
    # shared-table kernel (S > 1 scalars per point): 4 scalars x 90 000 points = 2813 waves per launch, twice
    ks = np.stack([u8(rng.getrandbits(250).to_bytes(32, "little")) for _ in range(4)])
    pts2 = np.ascontiguousarray(pts[:90000])
    out2, st2 = engine.g2_mul(ks, pts2)
    assert not st2.any()
    out2 = out2.reshape(90000, 4, 192)
    assert (out2[wl.B:2 * wl.B] == out2[:wl.B]).all()
    for j in rng.sample(range(wl.B), 6):
        for s in range(4):
            assert bytes(out2[j, s]) == oracle_mul(ks[s], wl.hashes[j])


def test_two_contexts_on_one_gpu_from_two_threads(engine, wl):
    from threshold_crypto_amd.engine import Engine
    want, st = engine.combine_g2(3, wl.idx, wl.shares)
    assert not st.any()
    fr2 = np.stack([u8(x._bytes()) for x in wl.shares_sk[:2]])
    want_sh, st = engine.g2_mul(fr2, wl.hashes)
    assert not st.any()
    other = Engine(0)
    errs = []

    def run(e, rounds):
        try:
            for _ in range(rounds):
                got, s = e.combine_g2(3, wl.idx, wl.shares)
                if s.any() or not (got == want).all():
                    errs.append("combine mismatch")
                sh, s2 = e.g2_mul(fr2, wl.hashes)
                if s2.any() or not (sh == want_sh).all():
                    errs.append("share mismatch")
        except Exception as ex:  # noqa: BLE001 -- reported below
            errs.append(repr(ex))

    ts = [threading.Thread(target=run, args=(engine, 6)), threading.Thread(target=run, args=(other, 6))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs[:3]


def test_identity_operands_through_the_arena(engine, wl):
    c.load()
    inf = np.zeros(192, dtype=np.uint8)
    inf[0] = 0x40
    pts = np.ascontiguousarray(wl.hashes[:64]).copy()
    pts[5] = inf
    pts[6] = inf
    k = np.stack([u8((0).to_bytes(32, "little")), u8((1).to_bytes(32, "little")), u8((12345).to_bytes(32, "little"))])
    out, st = engine.g2_mul(k, pts)
    assert not st.any()
    out = out.reshape(64, 3, 192)
    for j in (0, 5, 6, 7, 63):
        for s in range(3):
            assert bytes(out[j, s]) == oracle_mul(k[s], pts[j]), (j, s)
    assert bytes(out[5, 2]) == bytes(inf) and bytes(out[0, 0]) == bytes(inf)


def test_ladder_special_cases_take_the_safe_path(engine, wl):
    """The ladder loops add with the generic formula and redo a ladder on the fully guarded path when a lane may have
    met P = +-Q or the identity (csrc/tc_curve.h jac_add_mixed_generic).  Scalars and point sets built to hit those
    cases -- single multiplications, the chunked (n < 8) and the two-stage (n >= 8) linear combination, the
    small-index combine -- against Oracle A's textbook affine group law."""
    import tc_oracle as o
    R = o.R
    X = 0xd201000000010000
    P = o.g2_from_uncompressed(bytes(wl.hashes[0]))
    Q = o.g2_from_uncompressed(bytes(wl.hashes[1]))
    enc = lambda pt: u8(o.g2_uncompressed(pt))
    frb = lambda k: u8((k % R).to_bytes(32, "little"))
    # single multiplications: tiny scalars, r - small, powers of |x| (one-digit decompositions), 2^64 neighbours
    ks = [0, 1, 2, 3, 4, 7, R - 1, R - 2, R - 3, X, X + 1, X - 1, X * X, X * X + 1, X ** 3, X ** 3 + X, (1 << 64) - 1, 1 << 64,
          (R - 1) // 2, (R + 1) // 2]
    out, st = engine.g2_mul(np.stack([frb(k) for k in ks]), np.stack([enc(P), enc(Q)]))
    assert not st.any()
    for j, pt in enumerate((P, Q)):
        for s, k in enumerate(ks):
            assert bytes(out[j, s]) == o.g2_uncompressed(o.E2.mul(pt, k % R)), (j, k)
    # linear combinations whose partial sums collide or vanish
    negP = o.E2.neg(P)
    P2 = o.E2.dbl(P)
    for n, pts, sc in (
        (4, [P, P, negP, P], [1, 1, 1, 2]),
        (4, [P, negP, Q, Q], [5, 5, 7, R - 7]),
        (8, [P, P, negP, P, None, P, P2, Q], [1, 1, 1, 2, 5, R - 1, 1, 3]),
        (8, [P, negP, P, negP, P, negP, P, negP], [9, 9, 9, 9, 9, 9, 9, 9]),
        (9, [Q, Q, Q, Q, Q, Q, Q, Q, Q], [1, 2, 4, 8, 16, 32, 64, 128, R - 255]),
    ):
        pts_b = np.stack([enc(p) for p in pts])[None]
        sc_b = np.stack([frb(k) for k in sc])[None]
        got, st = engine.lincomb_g2(np.ascontiguousarray(sc_b), np.ascontiguousarray(pts_b))
        assert not st.any()
        want = None
        for p, k in zip(pts, sc):
            want = o.E2.add(want, o.E2.mul(p, k % R))
        assert bytes(got[0]) == o.g2_uncompressed(want), (n, sc)
    # small-index combine of four EQUAL shares (every subset sum of the short ladder is a multiple of one point)
    idx = np.array([[0, 1, 2, 3], [1, 4, 6, 9]], dtype=np.uint64)
    shares = np.stack([np.stack([enc(P)] * 4), np.stack([enc(Q)] * 4)])
    got, st = engine.combine_g2(3, idx, shares)
    assert not st.any()
    # sum_i lambda_i = 1: interpolating the constant polynomial gives the point back
    assert bytes(got[0]) == o.g2_uncompressed(P) and bytes(got[1]) == o.g2_uncompressed(Q)


def test_comb_signing_many_signers_vs_oracle(engine, wl):
    """tc_sign_shares_g2_batch with n >= 24 signers per message runs through the per-message comb (csrc/tc_comb.h):
    every share of 265 messages against Oracle B, plus a bad signer index, an identity point and an undecodable point."""
    c.load()
    rng = random.Random(31)
    N, n, B = 40, 30, 8192       # the comb runs from 24 signers and 8192 messages on (csrc/tc_launch.h)
    sk = np.stack([u8(rng.getrandbits(250).to_bytes(32, "little")) for _ in range(N)])
    sk[0] = u8((1).to_bytes(32, "little"))
    sk[1] = u8((2).to_bytes(32, "little"))
    sets = [np.array(sorted(rng.sample(range(N), n)), dtype=np.uint64) for _ in range(96)]
    idx = np.stack([sets[j % 96] for j in range(B)])
    idx[3, 0], idx[3, 1] = 0, 1                      # tiny scalars: the guarded fallback ladder
    idx[4, 7] = N + 9                                # out of range: fails its own share only
    pts = np.ascontiguousarray(np.tile(wl.hashes, ((B + wl.B - 1) // wl.B, 1))[:B]).copy()
    inf = np.zeros(192, dtype=np.uint8)
    inf[0] = 0x40
    pts[5] = inf
    pts[6, 100] ^= 1                                 # not on the curve any more
    out, st = engine.sign_shares_g2(sk, idx, pts)
    assert st.shape == (B, n) and out.shape == (B, n, 192)
    for j in (0, 1, 2, 3, 4, 5, 95, 96, 97):
        for s in range(n):
            if j == 4 and s == 7:
                assert st[j, s] == 3 and bytes(out[j, s]) == bytes(inf)
                continue
            assert st[j, s] == 0, (j, s)
            assert bytes(out[j, s]) == oracle_mul(sk[int(idx[j, s])], pts[j]), (j, s)
    # every share of 256 messages spread over the batch (wave and tile boundaries included), on all host threads
    pick = np.unique(np.concatenate([np.array([511, 512, 4095, 4096, 4097, B - 2, B - 1]), np.linspace(8, B - 1, 249).astype(np.int64)]))
    want, rc = c.sign_shares_batch(sk, idx[pick], pts[pick], c.host_threads())
    assert not rc.any() and not st[pick].any() and (want == out[pick]).all()
    assert (st[6] == 3).all() and all(bytes(out[6, s]) == bytes(inf) for s in range(n))
    assert not st[7:].any()
    # the comb and the per-chunk ladders agree on a whole batch: n = 23 (ladders) and n = 24 (comb) share 23 signers
    out23, st23 = engine.sign_shares_g2(sk, np.ascontiguousarray(idx[:, :23]), pts)
    out24, st24 = engine.sign_shares_g2(sk, np.ascontiguousarray(idx[:, :24]), pts)
    assert (out24[:, :23] == out23).all() and (st24[:, :23] == st23).all()
    # a small batch of the same jobs takes the ladders with fewer signers per lane pair: same bytes
    outs, sts = engine.sign_shares_g2(sk, np.ascontiguousarray(idx[:200]), np.ascontiguousarray(pts[:200]))
    assert (outs == out[:200]).all() and (sts == st[:200]).all()


def test_failed_device_allocation_is_an_error_and_the_context_survives(engine, wl):
    """SURVEY 8b "no panics on the hot path" at the C ABI: with the HBM full, the first G2 call of a FRESH context cannot allocate
    its 256 MB table arena -- the call must come back TC_ERR_HIP with the allocator's message (the Rust shim turns that into
    Err(GpuError), rust/threshold_crypto_gpu/gpu.rs check()), not abort, not hang, not return stale bytes; and once memory is
    free again the SAME context must work: no sticky HIP error, no half-built arena.  Same for a staging slot that cannot grow."""
    import torch
    from threshold_crypto_amd.engine import Engine, TcError
    from threshold_crypto_amd import _native
    base_want, st = engine.combine_g2(wl.t, wl.idx, wl.shares)
    assert not st.any()
    eng = Engine(0)                      # fresh: no arena, no staging slots yet
    eng.set_input_checks(False)
    hogs = _fill_hbm(32 << 20)
    free, _total = torch.cuda.mem_get_info()
    if free >= (2 << 30):                 # (a precondition of the test, not a property of the library)
        del hogs
        torch.cuda.empty_cache()
        pytest.skip("could not fill the HBM (free %d MB)" % (free >> 20))
    # a batch whose share buffer ALONE is larger than what is left (the allocator does not always get under a few hundred MB
    # when other contexts of the process hold memory): the staging slot, if not the arena, cannot be allocated
    B = max(256, ((free + (96 << 20)) // (4 * 192) + 255) // 256 * 256)
    reps = (B + wl.B - 1) // wl.B
    idx = np.ascontiguousarray(np.tile(wl.idx, (reps, 1))[:B])
    shares = np.ascontiguousarray(np.tile(wl.shares, (reps, 1, 1))[:B])
    want = np.ascontiguousarray(np.tile(base_want, (reps, 1))[:B])
    try:
        with pytest.raises(TcError) as e:
            eng.combine_g2(wl.t, idx, shares)
        # (whichever notices first: a staging / arena hipMalloc that fails, or the guard in front of the first launch)
        assert e.value.code == _native.TC_ERR_HIP and ("hipMalloc" in str(e.value) or "out of memory" in str(e.value)), str(e.value)
        with pytest.raises(TcError) as e2:                                                         # and again: still an error, still no abort
            eng.combine_g2(wl.t, idx, shares)
        assert e2.value.code == _native.TC_ERR_HIP
    finally:
        del hogs
        torch.cuda.empty_cache()
    got, st = eng.combine_g2(wl.t, idx, shares)                                                    # the same context, memory free again
    assert not st.any() and (got == want).all()
    ok = eng.verify_g2(wl.master_pk, got[:256], np.ascontiguousarray(wl.hashes[:256]))
    assert ok.all()
    eng.close()


def test_pairing_check_with_the_hbm_nearly_full_takes_the_form_without_a_line_buffer(engine, wl):
    """ADVICE r05 on the REAL path (the existing test forces it through TC_PAIRING_BUDGET): with under 3 GB of HBM free the prepared
    pairing form's smallest line buffer (1.013 GB) does not fit; pairing_line_budget then answers a tile of 0 and the check runs
    in the one-loop form instead of failing an allocation -- same verdicts as with the memory free."""
    import torch
    from threshold_crypto_amd.engine import Engine
    B = 20000                                           # above the four-lanes-per-check form's 16 384
    reps = (B + wl.B - 1) // wl.B
    idx = np.ascontiguousarray(np.tile(wl.idx, (reps, 1))[:B])
    shares = np.ascontiguousarray(np.tile(wl.shares, (reps, 1, 1))[:B])
    hashes = np.ascontiguousarray(np.tile(wl.hashes, (reps, 1))[:B])
    sig, st = engine.combine_g2(wl.t, idx, shares)
    assert not st.any()
    sig[5] = sig[6]
    sig[B - 1] = sig[0]
    want = engine.verify_g2(wl.master_pk, sig, hashes)
    assert want.sum() == B - 2
    # (TC_PRIVATE_RESERVE=0: this context does not insist on free HBM for the kernels' private segments -- the guard that would
    # turn this very call away, see the test below; 20 000 checks are 625 waves of 117 KB, which the runtime does find)
    os.environ["TC_PRIVATE_RESERVE"] = "0"
    try:
        eng = Engine(0)
    finally:
        os.environ.pop("TC_PRIVATE_RESERVE")
    eng.set_input_checks(False)
    assert eng.verify_g2(wl.master_pk, sig[:64], hashes[:64]).sum() == 63       # (the context's small buffers exist now)
    hogs = _fill_hbm(1500 << 20)    # (a third of what is free is the line budget: < 1.013 GB as long as < 3 GB are free; the kernels' own
                                    # private segments -- 0.3 GB for 20 000 checks -- still fit comfortably)
    try:
        free, _total = torch.cuda.mem_get_info()
        if not (1 << 30) < free < (2800 << 20):
            pytest.skip("could not bring the free HBM into 1-2.8 GB (free %d MB)" % (free >> 20))
        got = eng.verify_g2(wl.master_pk, sig, hashes)
    finally:
        del hogs
        torch.cuda.empty_cache()
    assert (got == want).all()
    eng.close()


def test_two_stage_kernels_run_tile_by_tile_when_the_table_budget_is_short(engine):
    """tc_api.hip msm_g2 / msm_g1: the per-share tables of the large-threshold path live in HBM (2 KB per share in G2); a batch whose
    tables exceed what the call may spend (a third of the free HBM, 1-24 GiB) runs as consecutive tiles through ONE table buffer.
    At 24 GiB no test ever needed a second tile: TC_MSM_BUDGET (read when the context is created) makes 3 000 jobs at t = 21 take
    nine tiles, 20 000 jobs seven -- and 90 000 G1 jobs under the library's own 1 GiB floor.  Every job combines 22 shares of ONE
    point, so every result must be [f(0)] h; a sample is recomputed by Oracle B."""
    import tc_oracle as o
    from conftest import engine_with_env
    c.load()
    rnd = random.Random(2121)
    t, N = 21, 40
    poly = [rnd.randrange(o.R) for _ in range(t + 1)]
    fr = np.stack([u8(o.secret_key_share(poly, i).to_bytes(32, "little")) for i in range(N)])
    h2 = o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R))
    h1 = o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R))
    s2, st = engine.g2_mul(fr, u8(o.g2_uncompressed(h2))[None])
    s1, st1 = engine.g1_mul(fr, u8(o.g1_uncompressed(h1))[None])
    assert not st.any() and not st1.any()
    want2 = u8(o.g2_uncompressed(o.E2.mul(h2, poly[0])))
    want1 = u8(o.g1_uncompressed(o.E1.mul(h1, poly[0])))
    sets = np.stack([np.array(sorted(rnd.sample(range(N), t + 1)), dtype=np.uint64) for _ in range(512)])
    for B, budget in ((3000, 16 << 20), (20000, 128 << 20), (90000, 1 << 30)):
        idx = np.ascontiguousarray(sets[(np.arange(B) * 7) % 512])
        sh2 = np.ascontiguousarray(s2[0][idx.astype(np.int64)])
        sh1 = np.ascontiguousarray(s1[0][idx.astype(np.int64)])
        with engine_with_env(TC_MSM_BUDGET=budget) as eng:
            eng.set_input_checks(False)
            out2, st2 = eng.combine_g2(t, idx, sh2) if B <= 20000 else (None, None)
            out1, st1 = eng.combine_g1(t, idx, sh1)
        assert not st1.any() and (out1 == want1).all(), (B, budget)
        if out2 is not None:
            assert not st2.any() and (out2 == want2).all(), (B, budget)
            for j in (0, B // 2, B - 1):
                rc, w = c.combine_g2(t, [int(i) for i in idx[j]], [bytes(x) for x in sh2[j]])
                assert rc == 0 and bytes(out2[j]) == w, j


def test_a_call_the_runtime_could_not_survive_is_turned_away(engine, wl):
    """The ROCm runtime allocates the kernels' private segments when a dispatch needs them and ABORTS THE PROCESS when it cannot
    (amd::roc::callbackQueue <- AqlQueue::DynamicQueueEventsHandler, reproduced with ~0.4 GB of HBM free after the call's own
    allocations: DESIGN.md 7): no error code reaches anybody.  Call::guard_private therefore compares the free HBM with what a
    call of this size could ask for before the first launch: with ~1.5 GB free a 16 384-job combination comes back TC_ERR_HIP
    with the reason, a 64-job one still runs, and with the memory free again the same context runs the large one."""
    import torch
    from threshold_crypto_amd.engine import Engine, TcError
    from threshold_crypto_amd import _native
    B = 16384
    reps = (B + wl.B - 1) // wl.B
    idx = np.ascontiguousarray(np.tile(wl.idx, (reps, 1))[:B])
    shares = np.ascontiguousarray(np.tile(wl.shares, (reps, 1, 1))[:B])
    want, st = engine.combine_g2(wl.t, idx, shares)
    assert not st.any()
    eng = Engine(0)
    eng.set_input_checks(False)
    out, st = eng.combine_g2(wl.t, idx[:64], shares[:64])
    assert not st.any()
    hogs = _fill_hbm(1500 << 20)
    try:
        free, _total = torch.cuda.mem_get_info()
        if not (1 << 30) < free < (2 << 30):
            pytest.skip("could not bring the free HBM into 1-2 GB (free %d MB)" % (free >> 20))
        with pytest.raises(TcError) as e:
            eng.combine_g2(wl.t, idx, shares)
        assert e.value.code == _native.TC_ERR_HIP and "private segments" in str(e.value), str(e.value)
        small, st = eng.combine_g2(wl.t, idx[:64], shares[:64])           # 64 jobs ask for 3 waves' worth: allowed, and it runs
        assert not st.any() and (small == want[:64]).all()
    finally:
        del hogs
        torch.cuda.empty_cache()
    got, st = eng.combine_g2(wl.t, idx, shares)
    assert not st.any() and (got == want).all()
    eng.close()
