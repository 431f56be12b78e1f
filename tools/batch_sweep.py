"""Throughput of the main kernels against the batch size (tail / occupancy effects)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
e = Engine(0); e.set_timing(True)
if os.environ.get("SWEEP_TRUSTED", "1") == "1":
    e.set_input_checks(False)      # operands made by the library itself, as in bench.py
BMAX = int(os.environ.get("SWEEP_BMAX", "262144"))
wl = ThresholdSigWorkload(e, 3, 10, BMAX)
sig_all, _ = e.combine_g2(3, wl.idx, wl.shares)
for B in [int(x) for x in os.environ.get("SWEEP", "16384,32768,49152,61440,65536,69632,98304,131072,196608,262144").split(",")]:
    if B > BMAX: continue
    best_c = best_v = 1e9
    for rep in range(3):
        sig, st = e.combine_g2(3, wl.idx[:B], wl.shares[:B]); best_c = min(best_c, e.last_kernel_ms())
        ok = e.verify_g2(wl.master_pk, sig_all[:B], wl.hashes[:B]); best_v = min(best_v, e.last_kernel_ms())
    assert ok.all()
    print(json.dumps({"B": B, "combine_ms": round(best_c, 3), "combine_M_per_s": round(B / best_c / 1e3, 3),
                      "verify_ms": round(best_v, 3), "verify_M_per_s": round(B / best_v / 1e3, 3)}), flush=True)
