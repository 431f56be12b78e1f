"""BASELINE config 5 shape on one GPU: t=67, N=200 threshold signatures (general combine path)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = int(os.environ.get("PROBE_B", "4096"))
e = Engine(0); e.set_timing(True)
wl = ThresholdSigWorkload(e, 67, 200, B)
res = {"t": 67, "N": 200, "B": B}
for rep in range(3):
    sig, st = e.combine_g2(67, wl.idx, wl.shares)
    res["combine_ms_%d" % rep] = round(e.last_kernel_ms(), 2)
assert not st.any()
ok = e.verify_g2(wl.master_pk, sig, wl.hashes)
assert ok.all()
res["combine_per_s"] = round(B / (res["combine_ms_2"] * 1e-3))
res["shares_per_s"] = res["combine_per_s"] * 68
print(json.dumps(res))
