"""k_point_mul<Fq2> alone (S = 1: one G2 point and one scalar per lane pair, no spills, per-lane table look-ups)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = int(os.environ.get("PROBE_B", "65536"))
e = Engine(0); e.set_timing(True)
wl = ThresholdSigWorkload(e, 3, 10, B)
fr = np.stack([np.frombuffer(s._bytes(), dtype=np.uint8) for s in wl.shares_sk[:1]])
ms = []
for rep in range(4):
    sh, _ = e.g2_mul(fr, wl.hashes); ms.append(round(e.last_kernel_ms(), 3))
print(json.dumps({"B": B, "k_point_mul_fq2_ms": ms}))
