"""Wire-level combine (tc_combine_signatures_wire_batch): 96-byte shares in, 96-byte signature out, 65 536 jobs resident in HBM.
python tools/wire_probe.py [reps]   (PROBE_B = batch)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = int(os.environ.get("PROBE_B", "65536"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
e = Engine(0); e.set_timing(True)
wl = ThresholdSigWorkload(e, 3, 10, B)
comp, st = e.g2_compress(np.ascontiguousarray(wl.shares.reshape(B * 4, 192)))
d_sh = torch.from_numpy(comp.reshape(B, 4, 96).copy()).cuda()
d_idx = torch.from_numpy(wl.idx.view(np.int64)).cuda()
d_un = torch.from_numpy(wl.shares).cuda()
ts, ts_dec, ts_un = [], [], []
for _ in range(reps):
    out, st = e.combine_signatures_wire(3, d_idx, d_sh); ts.append(round(e.last_kernel_ms(), 3))
    dec, st2 = e.g2_decompress(d_sh.reshape(B * 4, 96)); ts_dec.append(round(e.last_kernel_ms(), 3))
    e.set_input_checks(False)
    sig, st3 = e.combine_g2(3, d_idx, d_un); ts_un.append(round(e.last_kernel_ms(), 3))
    e.set_input_checks(True)
assert not st.cpu().numpy().any()
cs, _ = e.g2_compress(sig)
assert bool((cs == out).all())
print(json.dumps({"B": B, "wire_combine_ms": ts, "wire_combine_per_s": round(B / (min(ts) * 1e-3)), "decompress_4B_points_ms": ts_dec,
                  "combine_uncompressed_unchecked_ms": ts_un}), flush=True)
