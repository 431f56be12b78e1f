"""A/B timing of the main kernels for an experiment build: TC_AMD_LIB=<path> python tools/ab_probe.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = int(os.environ.get("PROBE_B", "65536"))
e = Engine(0); e.set_timing(True)
wl = ThresholdSigWorkload(e, 3, 10, B)
res = {"lib": os.path.basename(os.environ.get("TC_AMD_LIB", "default")), "B": B}
for rep in range(2):
    sig, st = e.combine_g2(3, wl.idx, wl.shares); res["combine_ms_%d" % rep] = round(e.last_kernel_ms(), 2)
    big = wl.idx + np.uint64(1 << 20)   # large indices: general Lagrange path
    sig2, st2 = e.combine_g2(3, big, wl.shares); res["combine_general_ms_%d" % rep] = round(e.last_kernel_ms(), 2)
    ok = e.verify_g2(wl.master_pk, sig, wl.hashes); res["verify_ms_%d" % rep] = round(e.last_kernel_ms(), 2)
    fr = np.stack([np.frombuffer(s._bytes(), dtype=np.uint8) for s in wl.shares_sk[:4]])
    sh, _ = e.g2_mul(fr, wl.hashes); res["sign4xB_ms_%d" % rep] = round(e.last_kernel_ms(), 2)
    h = e.hash_g2(wl.msg_flat, wl.msg_off); res["hash_ms_%d" % rep] = round(e.last_kernel_ms(), 2)
    c2, _ = e.g2_compress(sig); d2, std = e.g2_decompress(c2); res["g2_decompress_ms_%d" % rep] = round(e.last_kernel_ms(), 2)
assert os.environ.get('PROBE_NOASSERT') or ok.all() and not st.any() and (d2 == sig).all() and not std.any() and (h == wl.hashes).all()
print(json.dumps(res), flush=True)
