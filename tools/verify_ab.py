"""Same-box A/B of the pairing kernel: python tools/verify_ab.py libA.so libB.so  (alternating, 4 rounds each)."""
import os, sys, json, subprocess
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2:
    out = {}
    for rnd in range(3):
        for lib in sys.argv[1:]:
            env = dict(os.environ, TC_AMD_LIB=os.path.join(HERE, "threshold_crypto_amd", lib))
            r = subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True)
            out.setdefault(lib, []).append(json.loads(r.stdout.strip().splitlines()[-1]))
    print(json.dumps(out))
    sys.exit(0)
sys.path.insert(0, HERE)
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = 65536
e = Engine(0); e.set_timing(True)
wl = ThresholdSigWorkload(e, 3, 10, B)
sig, st = e.combine_g2(3, wl.idx, wl.shares)
ts = []
for rep in range(4):
    ok = e.verify_g2(wl.master_pk, sig, wl.hashes); ts.append(round(e.last_kernel_ms(), 2))
assert ok.all()
print(json.dumps(ts))
