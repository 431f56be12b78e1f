import os, sys, json, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = int(os.environ.get("PROBE_B", "65536")); K = 24; NC = int(os.environ.get("NC", "4"))
dev = torch.device("cuda:0")
es = [Engine(0) for _ in range(NC)]
wl = ThresholdSigWorkload(es[0], 3, 10, B)
d_idx = torch.from_numpy(wl.idx.view(np.int64)).to(dev); d_sh = torch.from_numpy(wl.shares).to(dev)
for e in es:
    e.set_timing(False); e.combine_g2(3, d_idx, d_sh)
for e in es: e.sync()
res = {}
for rep in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = [es[i % NC].combine_g2(3, d_idx, d_sh) for i in range(K)]
    for e in es: e.sync()
    torch.cuda.synchronize(); res["rep%d_ms_per_step" % rep] = round((time.perf_counter() - t0) / K * 1e3, 3)
print(json.dumps({"B": B, "contexts": NC, **res}))
