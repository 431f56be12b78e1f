"""Two contexts on one GPU, consecutive batches alternating between them without host synchronisation: the tail of one
combine launch overlaps the head of the next (device-resident operands; csrc/tc_api.hip device-I/O mode)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = int(os.environ.get("PROBE_B", "65536")); K = int(os.environ.get("PROBE_STEPS", "40"))
dev = torch.device("cuda:0")
e1, e2, e3, e4 = Engine(0), Engine(0), Engine(0), Engine(0)
wl = ThresholdSigWorkload(e1, 3, 10, B)
d_idx = torch.from_numpy(wl.idx.view(np.int64)).to(dev); d_sh = torch.from_numpy(wl.shares).to(dev)
res = {"B": B, "steps": K}
for name, engines in (("one_context", [e1]), ("two_contexts", [e1, e2]), ("three_contexts", [e1, e2, e3]), ("four_contexts", [e1, e2, e3, e4])):
    for e in engines:
        e.set_timing(False)
        e.combine_g2(3, d_idx, d_sh)
    for e in engines: e.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    for i in range(K):
        outs.append(engines[i % len(engines)].combine_g2(3, d_idx, d_sh))
    for e in engines: e.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(int(st.to(torch.int32).sum().item()) == 0 for _, st in outs[-2:])
    assert bool((outs[-1][0] == outs[-2][0]).all().item())
    res[name + "_ms_per_step"] = round(dt / K * 1e3, 3); res[name + "_M_per_s"] = round(B * K / dt / 1e6, 3)
print(json.dumps(res))
