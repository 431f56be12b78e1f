"""The HBM table arena of the G2 ladder kernels (csrc/tc_table.h): slots are borrowed per wave and recycled.

  * a launch with more waves than the arena has slots (8 XCDs x 512) recycles every slot several times --
    all results against a second launch in a different wave order, a sample against Oracle B;
  * two contexts on the same GPU, driven from two host threads at once, each with its own arena;
  * jobs whose ladder meets the point at infinity (identity operand, zero scalar) keep their table entries'
    infinity flags through the arena.
"""
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
