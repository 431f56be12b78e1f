"""Round 6: what the second stream of checked-input mode is worth per entry at the BASELINE batch (t = 3, N = 10,
65 536 jobs, device-resident operands, the context's DEFAULT membership tests on): median wall time of one call with
TC_CHECKS_BESIDE=0 (the tests in front of the main kernels, one stream: rounds 2-5) and =1 (shipped), three fresh contexts each.
    python tools/checks_beside_entries.py  -> profiles/r06_checks_beside_entries.txt"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload, ThresholdEncWorkload
dev = torch.device("cuda", 0)
t, N, B = 3, 10, int(os.environ.get("PROBE_B", "65536"))
gen = Engine(0); gen.set_input_checks(False)
wl = ThresholdSigWorkload(gen, t, N, B)
we = ThresholdEncWorkload(gen, t, N, B)
to = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to(dev)
d_idx, d_sh, d_hash = to(wl.idx), to(wl.shares), to(wl.hashes)
sig, st = gen.combine_g2(t, d_idx, d_sh); gen.sync()
pk = to(np.ascontiguousarray(wl.master_pk))
d_msgs, d_off = to(wl.msg_flat), to(wl.msg_off)
du, dv, dw, doff, didx, dsh = to(we.u), to(we.v), to(we.w), to(we.off), to(we.idx), to(we.shares)
entries = {"combine_signatures": lambda e: e.combine_g2(t, d_idx, d_sh), "verify_g2": lambda e: e.verify_g2(pk, sig, d_hash),
           "verify (hash on device)": lambda e: e.verify_sig(pk, sig, d_msgs, d_off),
           "Ciphertext::verify": lambda e: e.ciphertext_verify(du, dv, doff, dw), "PublicKeySet::decrypt": lambda e: e.decrypt(t, didx, dsh, dv, doff)}
res = {k: {} for k in entries}
for mode in ("0", "1"):
    os.environ["TC_CHECKS_BESIDE"] = mode
    for trial in range(3):
        e = Engine(0); e.set_timing(False); e.set_input_checks(True)
        for name, fn in entries.items():
            fn(e); e.sync()
            ts = []
            for _ in range(7):
                e.sync(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(e); e.sync(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            res[name].setdefault("one_stream_ms" if mode == "0" else "beside_ms", []).append(round(sorted(ts)[3] * 1e3, 2))
        e.close()
for name, r in res.items():
    a, b = sorted(r["one_stream_ms"])[1], sorted(r["beside_ms"])[1]
    print(json.dumps({"entry": name, "jobs": B, "one_stream_ms": r["one_stream_ms"], "beside_ms": r["beside_ms"], "saved_ms": round(a - b, 2), "saved_frac": round((a - b) / a, 3)}), flush=True)
