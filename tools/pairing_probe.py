"""Timing of the pairing check (verify_g2 over a 65 536-job batch): python tools/pairing_probe.py [reps]
TC_PAIRING_FORM = quad | pair | fused overrides the form the library would pick (k_pairing.hip); PROBE_B, PROBE_NOCHECKS."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = int(os.environ.get("PROBE_B", "65536"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
e = Engine(0); e.set_timing(True)
if os.environ.get("PROBE_NOCHECKS"):
    e.set_input_checks(False)   # operands are the library's own outputs: time the check alone
wl = ThresholdSigWorkload(e, 3, 10, B)
sig, st = e.combine_g2(3, wl.idx, wl.shares)
bad = sig.copy(); bad[::16] = sig[(np.arange(0, B, 16) + 1) % B]
want = np.ones(B, np.uint8); want[::16] = 0
ts = []
for rep in range(reps):
    ok = e.verify_g2(wl.master_pk, bad, wl.hashes); ts.append(round(e.last_kernel_ms(), 3))
assert (ok == want).all()
print(json.dumps({"lib": os.path.basename(os.environ.get("TC_AMD_LIB", "default")), "form": os.environ.get("TC_PAIRING_FORM", "auto"), "checks": e.input_checks(), "B": B, "verify_ms": ts}), flush=True)
