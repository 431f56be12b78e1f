"""Poly::commitment throughput: fixed-base kernel (LDS window table of g1) vs the variable-base GLV kernel on the
same scalars, device-resident.  usage: python tools/dkg_probe.py [M]  -> one JSON line (profiles/r02_dkg_probe.txt)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.api import _G1_GEN
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
e = Engine(0); e.set_timing(True)
rng = np.random.default_rng(7)
fr = rng.integers(0, 256, size=(M, 32), dtype=np.uint8); fr[:, 31] &= 0x3f      # < 2^254 < r
dev = torch.device("cuda", 0)
d_fr = torch.from_numpy(fr).to(dev)
g = torch.from_numpy(np.frombuffer(_G1_GEN, dtype=np.uint8).copy()).to(dev)[None].contiguous()
res = {"coefficients": M}
for rep in range(2):
    out, st = e.g1_commitment(d_fr); res["fixed_base_ms"] = round(e.last_kernel_ms(), 3)
chunk = 1 << 16                                                  # variable base: S scalars x 1 point
ref, st2 = e.g1_mul(d_fr[:chunk].contiguous(), g); res["variable_base_ms_per_65536"] = round(e.last_kernel_ms(), 3)
assert bool((ref[0] == out[:chunk]).all().item()) and not bool(st.any().item())
res["fixed_base_per_s"] = round(M / (res["fixed_base_ms"] * 1e-3), 1)
res["variable_base_per_s"] = round(chunk / (res["variable_base_ms_per_65536"] * 1e-3), 1)
res["speedup"] = round(res["fixed_base_per_s"] / res["variable_base_per_s"], 2)
print(json.dumps(res), flush=True)
