"""VERDICT r05 item 2, the probe: does the two-messages-per-pair hash (k_hash_g2_x2: one wave per SIMD at 65 536 messages, 18 % fewer
multiply-adds) run BESIDE hash-independent pairing work on a second stream finish sooner than the shipped order (k_hash_g2, then the
pairing kernels, one stream)?  Existing kernels only, two contexts = two HIP streams on one GPU, device-resident operands:

    seq        hash_g2 (one message per pair) then verify_g2, one after the other             -- what tc_verify_sig_batch does today
    co_x2      hash_g2 (two per pair) on context A  ||  verify_g2 on context B                -- the proposal's best case: NO dependency
    co_single  hash_g2 (one per pair) on A          ||  verify_g2 on B                        -- what plain overlap (tail filling) gives

co_* overlap the hash with the WHOLE pairing check of another batch; the real composition could only overlap it with the half of
stage P that does not read the hash point (about 3 of the 21 ms), so `seq - co_x2` is an UPPER bound on what co-scheduling can save.
    python tools/cosched_probe.py   -> one JSON line per batch size (profiles/r06_cosched_probe.txt)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from threshold_crypto_amd.engine import Engine, pack_messages
from threshold_crypto_amd.workload import ThresholdSigWorkload

dev = torch.device("cuda", 0)
# TC_DUO_MIN is read when a context is created: one hashing context per form, one context for the pairing work
os.environ["TC_DUO_MIN"] = str(10 ** 12); a1 = Engine(0)
os.environ["TC_DUO_MIN"] = "1"; a2 = Engine(0)
os.environ.pop("TC_DUO_MIN"); b = Engine(0)
a = a1
for e in (a1, a2, b):
    e.set_timing(False); e.set_input_checks(False)
for B in (65536, 131072):
    wl = ThresholdSigWorkload(a, 3, 10, B)
    sig, st = a.combine_g2(3, wl.idx, wl.shares)
    d_sig, d_hash = torch.from_numpy(sig).to(dev), torch.from_numpy(wl.hashes).to(dev)
    d_pk = torch.from_numpy(wl.master_pk.copy()).to(dev)
    d_msgs, d_off = torch.from_numpy(wl.msg_flat).to(dev), torch.from_numpy(wl.msg_off.view(np.int64)).to(dev)

    def sync():
        a1.sync(); a2.sync(); b.sync(); torch.cuda.synchronize()

    def run(form, overlapped, reps=5):
        a = a2 if form == "x2" else a1
        best, outs = 1e9, None
        for _ in range(reps + 1):
            sync(); t0 = time.perf_counter()
            h = a.hash_g2(d_msgs, d_off)
            if not overlapped:
                a.sync()
            ok = b.verify_g2(d_pk, d_sig, d_hash)
            sync(); dt = time.perf_counter() - t0
            best, outs = min(best, dt), (h, ok)
        assert bool((outs[0] == d_hash).all().item()) and bool(outs[1].all().item())
        return best * 1e3

    def alone(fn, eng, reps=5):
        best = 1e9
        for _ in range(reps + 1):
            sync(); t0 = time.perf_counter(); fn(); eng.sync(); best = min(best, time.perf_counter() - t0)
        return best * 1e3

    h1 = alone(lambda: a1.hash_g2(d_msgs, d_off), a1)
    h2 = alone(lambda: a2.hash_g2(d_msgs, d_off), a2)
    v = alone(lambda: b.verify_g2(d_pk, d_sig, d_hash), b)
    seq1, seq2 = run("single", False), run("x2", False)
    co1, co2 = run("single", True), run("x2", True)
    print(json.dumps({"jobs": B, "hash_single_ms": round(h1, 3), "hash_x2_ms": round(h2, 3), "verify_g2_ms": round(v, 3),
                      "seq_single_ms": round(seq1, 3), "seq_x2_ms": round(seq2, 3), "co_single_ms": round(co1, 3), "co_x2_ms": round(co2, 3),
                      "saved_by_overlap_single_ms": round(seq1 - co1, 3), "saved_by_overlap_x2_ms": round(seq1 - co2, 3),
                      "x2_beside_pairing_vs_single_beside_pairing_ms": round(co1 - co2, 3)}), flush=True)
