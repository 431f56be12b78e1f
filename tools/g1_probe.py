"""G1 kernels: scalar multiplication (GLV), threshold-decryption combine (fast and general path)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = int(os.environ.get("PROBE_B", "65536"))
e = Engine(0); e.set_timing(True)
if os.environ.get("PROBE_TRUSTED", "0") == "1":
    e.set_input_checks(False)     # operands made by the library itself (as in bench.py)
wl = ThresholdSigWorkload(e, 3, 10, 256)
rng = np.random.default_rng(5)
fr = rng.integers(0, 256, size=(B, 32), dtype=np.uint8); fr[:, 31] &= 0x3f
gen = np.tile(wl.master_pk[None], (1, 1))
res = {"B": B}
pts, st = e.g1_mul(fr[:4].copy(), np.tile(wl.master_pk[None], (B, 1)))          # (B, 4, 96): 4 G1 "shares" per job
assert not st.any()
idx = np.tile(np.array([[0, 2, 5, 7]], dtype=np.uint64), (B, 1))
for rep in range(3):
    out, st = e.g1_mul(fr[:1].copy(), np.ascontiguousarray(pts[:, 0])); res["g1_mul_ms"] = round(e.last_kernel_ms(), 3)
    c, st = e.combine_g1(3, idx, pts); res["combine_g1_fast_ms"] = round(e.last_kernel_ms(), 3)
    c2, st2 = e.combine_g1(3, idx + np.uint64(1 << 20), pts); res["combine_g1_general_ms"] = round(e.last_kernel_ms(), 3)
print(json.dumps(res))
