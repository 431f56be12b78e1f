"""Round 6 probe: the DEFAULT configuration of combine_signatures (every share tested for group membership: 4 x 65 536 G2 tests, then
the combination) runs its two phases one after the other on one stream -- 5.9 + 4.9 ms.  The tests do not depend on the combination
and the combination's launch ends with two thirds of the SIMDs idle: does the membership kernel on a SECOND stream fill
that tail?  Existing kernels, two contexts = two HIP streams, device-resident operands:

    seq     g2_subgroup_check(262 144 points) then combine_g2 (checks off), one after the other
    co      combine_g2 on context B first, the membership tests on context A beside it
    co_rev  the membership tests first, the combination beside them
    default what tc_combine_g2_batch does today with the context's default checks on (one call)

    python tools/checks_overlap_probe.py   -> one JSON line (profiles/r06_checks_overlap_probe.txt)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload

dev = torch.device("cuda", 0)
a, b, c = Engine(0), Engine(0), Engine(0)
for e in (a, b):
    e.set_timing(False); e.set_input_checks(False)
c.set_timing(False); c.set_input_checks(True)
t, N, B = 3, 10, int(os.environ.get("PROBE_B", "65536"))
wl = ThresholdSigWorkload(a, t, N, B)
d_idx = torch.from_numpy(wl.idx.view(np.int64)).to(dev)
d_sh = torch.from_numpy(wl.shares).to(dev)
d_pts = d_sh.reshape(B * (t + 1), 192)


def sync():
    a.sync(); b.sync(); c.sync(); torch.cuda.synchronize()


def best(fn, reps=7):
    t_best, out = 1e9, None
    for _ in range(reps + 1):
        sync(); t0 = time.perf_counter(); out = fn(); sync(); t_best = min(t_best, time.perf_counter() - t0)
    return t_best * 1e3, out


def seq():
    ok = a.g2_subgroup_check(d_pts); a.sync()
    return ok, b.combine_g2(t, d_idx, d_sh)


def co():
    r = b.combine_g2(t, d_idx, d_sh)
    return a.g2_subgroup_check(d_pts), r


def co_rev():
    ok = a.g2_subgroup_check(d_pts)
    return ok, b.combine_g2(t, d_idx, d_sh)


chk, _ = best(lambda: a.g2_subgroup_check(d_pts))
cmb, _ = best(lambda: b.combine_g2(t, d_idx, d_sh))
s, (ok, (sig, st)) = best(seq)
o, (ok2, (sig2, st2)) = best(co)
r, (ok3, (sig3, st3)) = best(co_rev)
dflt, (sig4, st4) = best(lambda: c.combine_g2(t, d_idx, d_sh))
assert bool(ok.all().item()) and bool(ok2.all().item()) and bool(ok3.all().item())
assert bool((sig == sig2).all().item()) and bool((sig == sig3).all().item()) and bool((sig == sig4).all().item())
# two contexts in flight (bench.py `streaming`): with the tests OFF (two main streams) and with the tests ON (two main + two second streams)
c2 = Engine(0); c2.set_timing(False); c2.set_input_checks(True)


def in_flight(e1, e2, n=8):
    outs = [(e1 if i % 2 == 0 else e2).combine_g2(t, d_idx, d_sh) for i in range(n)]
    e1.sync(); e2.sync()
    return outs


for e1, e2 in ((a, b), (c, c2)):
    in_flight(e1, e2)
fl_off, _ = best(lambda: in_flight(a, b), reps=3)
fl_on, outs = best(lambda: in_flight(c, c2), reps=3)
assert all(bool((o_sig == sig).all().item()) and not bool(o_st.any().item()) for o_sig, o_st in outs)
print(json.dumps({"jobs": B, "membership_tests_alone_ms": round(chk, 3), "combine_alone_ms": round(cmb, 3), "seq_ms": round(s, 3),
                  "combine_then_tests_beside_ms": round(o, 3), "tests_then_combine_beside_ms": round(r, 3), "default_one_call_ms": round(dflt, 3),
                  "saved_ms": round(s - min(o, r), 3), "saved_frac_of_default": round((s - min(o, r)) / dflt, 4),
                  "two_contexts_in_flight_ms_per_batch": {"tests_off": round(fl_off / 8, 3), "tests_on_default": round(fl_on / 8, 3)},
                  "context_tuning": c.tuning()}), flush=True)
