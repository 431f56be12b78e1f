"""Where does the combine kernel's time go?  Same batch with (a) per-job signer subsets, (b) one subset for all."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = 65536
e = Engine(0); e.set_timing(True)
wl = ThresholdSigWorkload(e, 3, 10, B)
res = {}
for rep in range(2):
    sig, st = e.combine_g2(3, wl.idx, wl.shares); res["combine_random_subsets_ms"] = round(e.last_kernel_ms(), 2)
# all jobs use job 0's subset: need matching shares -> recompute shares for subset of job 0 for every job via g2_mul
ids = wl.idx[0]
fr = np.stack([np.frombuffer(wl.shares_sk[int(i)]._bytes(), dtype=np.uint8) for i in ids])
sh, _ = e.g2_mul(fr, wl.hashes)          # (B, 4, 192)
idx = np.tile(ids[None, :], (B, 1)).astype(np.uint64)
for rep in range(2):
    sig2, st2 = e.combine_g2(3, idx, np.ascontiguousarray(sh)); res["combine_same_subset_ms"] = round(e.last_kernel_ms(), 2)
assert (sig2 == sig).all()
ids2 = np.array([0, 1, 2, 3], np.uint64)
fr = np.stack([np.frombuffer(wl.shares_sk[int(i)]._bytes(), dtype=np.uint8) for i in ids2])
sh, _ = e.g2_mul(fr, wl.hashes)
idx = np.tile(ids2[None, :], (B, 1)).astype(np.uint64)
for rep in range(2):
    sig3, st3 = e.combine_g2(3, idx, np.ascontiguousarray(sh)); res["combine_subset_0123_ms"] = round(e.last_kernel_ms(), 2)
assert (sig3 == sig).all()
print(json.dumps(res))
