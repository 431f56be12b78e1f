"""PublicKeySet::decrypt's combination (t = 3) over RANDOM 4-of-10 signer subsets -- the bench's distribution: a fifth of the jobs
have the common denominator D = 1 -- at 65 536 ... 524 288 jobs: python tools/g1_group_probe.py   (TC_AMD_LIB selects a build)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
e = Engine(0); e.set_timing(True); e.set_input_checks(False)
wl = ThresholdSigWorkload(e, 3, 10, 65536)
rng = np.random.default_rng(5)
fr = rng.integers(0, 256, size=(4, 32), dtype=np.uint8); fr[:, 31] &= 0x3f
res = {"lib": os.path.basename(os.environ.get("TC_AMD_LIB", "default"))}
for B in (65536, 131072, 262144, 524288):
    pts, st = e.g1_mul(fr, np.tile(wl.master_pk[None], (B, 1)))          # (B, 4, 96): four G1 "shares" per job
    idx = np.ascontiguousarray(np.tile(wl.idx, (B // 65536, 1)))
    ts = []
    for rep in range(3):
        out, st = e.combine_g1(3, idx, pts); ts.append(round(e.last_kernel_ms(), 3))
    assert not st.any()
    res[str(B)] = {"ms": min(ts), "M_per_s": round(B / min(ts) / 1e3, 2), "digest": int(out.astype(np.uint64).sum())}
    del pts, idx, out
print(json.dumps(res), flush=True)
