"""Per-multiplication time of tc_g2_mul_batch for S = 1 (one point per lane pair) and S > 1 (shared table)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = int(os.environ.get("PROBE_B", "65536"))
e = Engine(0); e.set_timing(True)
wl = ThresholdSigWorkload(e, 3, 10, B)
res = {}
for S in (4, 10, 4, 10):
    fr = np.stack([np.frombuffer(s._bytes(), dtype=np.uint8) for s in wl.shares_sk[:S]])
    for rep in range(3):
        sh, _ = e.g2_mul(fr, wl.hashes)
    res.setdefault("S%d_ms_per_B" % S, []).append(round(e.last_kernel_ms() / S, 3))
print(json.dumps(res))
