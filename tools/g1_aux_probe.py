"""The secondary G1 kernels at one / two waves per SIMD: Commitment::evaluate (tc_public_key_share_batch), the general-path G1 combination,
tc_g1_lincomb_batch (n = 4), the G1 membership test -- kernel ms per batch size.  TC_AMD_LIB selects an experiment build."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import key_set
e = Engine(0); e.set_timing(True); e.set_input_checks(False)
res = {"lib": os.path.basename(os.environ.get("TC_AMD_LIB", "default"))}
sks = key_set(3)
commit = np.stack([np.frombuffer(c, dtype=np.uint8) for c in sks.public_keys(e).commit])
rng = np.random.default_rng(3)
for B in (65536, 131072, 262144):
    idx = rng.integers(0, 1 << 20, size=B, dtype=np.uint64)
    r = {}
    for rep in range(2):
        pks, st = e.public_key_shares(commit, idx); r["public_key_share_ms"] = round(e.last_kernel_ms(), 3)
        ok = e.g1_subgroup_check(pks); r["g1_subgroup_check_ms"] = round(e.last_kernel_ms(), 3)
        pts = np.ascontiguousarray(np.broadcast_to(pks[:, None, :], (B, 4, 96)))
        big = np.tile(np.array([[1 << 20, (1 << 20) + 2, (1 << 20) + 5, (1 << 20) + 7]], dtype=np.uint64), (B, 1))
        c, st = e.combine_g1(3, big, pts); r["combine_g1_general_ms"] = round(e.last_kernel_ms(), 3)
        sc = rng.integers(0, 256, size=(B, 4, 32), dtype=np.uint8); sc[:, :, 31] &= 0x3f
        l, st = e.lincomb_g1(sc, pts); r["lincomb_g1_n4_ms"] = round(e.last_kernel_ms(), 3)
    assert ok.all()
    r["digest"] = int(c.astype(np.uint64).sum() + l.astype(np.uint64).sum() + pks.astype(np.uint64).sum())
    res[str(B)] = r
print(json.dumps(res), flush=True)
