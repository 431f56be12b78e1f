"""Composed entry points (hash folded into the scalar / the G1 operand) against their two-call equivalents."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = int(os.environ.get("PROBE_B", "65536"))
e = Engine(0); e.set_timing(True)
wl = ThresholdSigWorkload(e, 3, 10, B)
res = {"B": B}
sig, st = e.combine_g2(3, wl.idx, wl.shares)
for S in (1, 4):
    fr = np.stack([np.frombuffer(s._bytes(), dtype=np.uint8) for s in wl.shares_sk[:S]])
    for rep in range(2):
        h = e.hash_g2(wl.msg_flat, wl.msg_off); t_h = e.last_kernel_ms()
        a, _ = e.g2_mul(fr, h); t_m = e.last_kernel_ms()
        b, stb = e.sign(fr, wl.msg_flat, wl.msg_off); t_s = e.last_kernel_ms()
    assert (a == b).all() and not stb.any()
    res["S%d_hash+mul_ms" % S] = round(t_h + t_m, 2); res["S%d_sign_batch_ms" % S] = round(t_s, 2)
for rep in range(2):
    h = e.hash_g2(wl.msg_flat, wl.msg_off); t_h = e.last_kernel_ms()
    ok1 = e.verify_g2(wl.master_pk, sig, h); t_v = e.last_kernel_ms()
    ok2 = e.verify_sig(wl.master_pk, sig, wl.msg_flat, wl.msg_off); t_vs = e.last_kernel_ms()
assert ok1.all() and ok2.all()
res["hash+verify_g2_ms"] = round(t_h + t_v, 2); res["verify_sig_ms"] = round(t_vs, 2)
print(json.dumps(res))
