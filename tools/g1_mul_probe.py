"""tc_g1_mul_batch at several batch sizes, register-table kernel vs the arena form (TC_G1_MUL_FORM=regs|arena; unset = the library's choice).
python tools/g1_mul_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from threshold_crypto_amd.engine import Engine
e = Engine(0); e.set_timing(True); e.set_input_checks(False)
rng = np.random.default_rng(5)
res = {"form": os.environ.get("TC_G1_MUL_FORM", "auto")}
from threshold_crypto_amd.api import _G1_GEN
for B in (65536, 131072, 262144, 524288):
    fr = rng.integers(0, 256, size=(B, 32), dtype=np.uint8); fr[:, 31] &= 0x3f
    pts, st = e.g1_mul(fr[:1].copy(), np.tile(np.frombuffer(_G1_GEN, np.uint8)[None], (1, 1)))
    base, st = e.g1_mul(fr[:B // 64 if B >= 64 else 1][:1].copy(), np.ascontiguousarray(pts[:, 0]))
    # B distinct points: k_j * g1
    P, st = e.g1_mul(fr[:1].copy(), np.tile(np.frombuffer(_G1_GEN, np.uint8)[None], (B, 1)))
    d_pts = torch.from_numpy(np.ascontiguousarray(P[:, 0])).cuda()
    d_fr = torch.from_numpy(fr[1:2].copy()).cuda()
    ts = []
    for rep in range(4):
        out, st = e.g1_mul(d_fr, d_pts); ts.append(e.last_kernel_ms())
    assert not st.cpu().numpy().any()
    res[str(B)] = {"ms": round(min(ts), 3), "M_per_s": round(B / min(ts) / 1e3, 2), "digest": int(out.cpu().numpy().astype(np.uint64).sum())}
print(json.dumps(res), flush=True)
