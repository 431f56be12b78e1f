"""k_g2_mul_gather alone: the shares of B messages by their 68 selected signers out of 200 (BASELINE config 5 shape)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
from threshold_crypto_amd.config5 import signer_subsets_np
B = int(os.environ.get("PROBE_B", "32768"))
e = Engine(0); e.set_timing(True)
wl = ThresholdSigWorkload(e, 3, 10, 1024)
rng = np.random.default_rng(3)
sk = rng.integers(0, 256, size=(200, 32), dtype=np.uint8); sk[:, 31] &= 0x3f
idx = signer_subsets_np(B, 200, 67)
hashes = np.ascontiguousarray(np.tile(wl.hashes, (B // 1024 + 1, 1))[:B])
ms = []
for rep in range(3):
    sh, st = e.sign_shares_g2(sk, idx, hashes); ms.append(round(e.last_kernel_ms(), 2))
assert not st.any()
print(json.dumps({"B": B, "gather_ms": ms, "shares_per_s": round(B * 68 / (min(ms) * 1e-3))}))
