import os, sys, json, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = 65536; K = 12; NC = int(os.environ.get("NC", "4"))
dev = torch.device("cuda:0")
es = [Engine(0) for _ in range(NC)]
wl = ThresholdSigWorkload(es[0], 3, 10, B)
sig, st = es[0].combine_g2(3, wl.idx, wl.shares)
d_sig = torch.from_numpy(sig).to(dev); d_h = torch.from_numpy(wl.hashes).to(dev); d_pk = torch.from_numpy(wl.master_pk).to(dev)
for e in es:
    e.set_timing(False); e.verify_g2(d_pk, d_sig, d_h)
for e in es: e.sync()
res = {}
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = [es[i % NC].verify_g2(d_pk, d_sig, d_h) for i in range(K)]
    for e in es: e.sync()
    torch.cuda.synchronize(); res["rep%d" % rep] = round((time.perf_counter() - t0) / K * 1e3, 2)
print(json.dumps({"verify_ms_per_step": res, "contexts": NC}))
