"""Threshold-decryption combiner (G1) at the config-5 threshold: t=67, N=200.  usage: python tools/g1_large_t_probe.py [B]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import key_set
from threshold_crypto_amd.config5 import signer_subsets_np
from threshold_crypto_amd.api import _G1_GEN
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
t, N = 67, 200
e = Engine(0); e.set_timing(True)
sks = key_set(t)
fr = np.stack([np.frombuffer(sks.secret_key_share(i)._bytes(), dtype=np.uint8) for i in range(N)])
u, _ = e.g1_mul(np.arange(1, 33, dtype=np.uint8).reshape(1, 32) % 61, np.frombuffer(_G1_GEN, dtype=np.uint8)[None].copy())
allsh, st = e.g1_mul(fr, np.ascontiguousarray(u[:, 0]))          # (1, N, 96): every node's decryption share of one ciphertext
idx = signer_subsets_np(B, N, t)
shares = np.ascontiguousarray(allsh[0][idx.astype(np.int64)])    # (B, t+1, 96)
res = {"t": t, "N": N, "B": B}
for rep in range(2):   # a context's default: every share tested for group membership first (two 64-bit ladders per G1 point)
    out, st = e.combine_g1(t, idx, shares); res["combine_g1_with_input_checks_ms"] = round(e.last_kernel_ms(), 2)
e.set_input_checks(False)   # the shares are outputs of tc_g1_mul_batch: known members
for rep in range(2):
    out, st = e.combine_g1(t, idx, shares); res["combine_g1_ms"] = round(e.last_kernel_ms(), 2)
assert not st.any() and (out == out[0]).all()
master, _ = e.g1_mul(np.frombuffer(sks.poly[0].to_bytes(32, "little"), dtype=np.uint8)[None].copy(), np.ascontiguousarray(u[:, 0]))
assert (out[0] == master[0, 0]).all()
res["combine_g1_per_s"] = round(B / (res["combine_g1_ms"] * 1e-3))
print(json.dumps(res), flush=True)
