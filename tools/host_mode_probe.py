"""PCIe-inclusive rate: the same batch through the C ABI with HOST buffers (numpy) vs device-resident (torch)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = 65536
e = Engine(0); e.set_timing(True)
wl = ThresholdSigWorkload(e, 3, 10, B)
res = {}
for rep in range(3):
    t0 = time.perf_counter(); sig, st = e.combine_g2(3, wl.idx, wl.shares); dt = time.perf_counter() - t0
    res["host_buffers_wall_ms"] = round(dt * 1e3, 2); res["host_buffers_kernel_ms"] = round(e.last_kernel_ms(), 2)
dev = torch.device("cuda", 0)
d_idx = torch.from_numpy(wl.idx.view(np.int64)).to(dev); d_sh = torch.from_numpy(wl.shares).to(dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); s2, st2 = e.combine_g2(3, d_idx, d_sh); e.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res["device_buffers_wall_ms"] = round(dt * 1e3, 2)
t0 = time.perf_counter(); ok = e.verify_g2(wl.master_pk, sig, wl.hashes); dt = time.perf_counter() - t0
res["verify_host_buffers_wall_ms"] = round(dt * 1e3, 2); res["verify_kernel_ms"] = round(e.last_kernel_ms(), 2)
res["bytes_in_MB"] = round((wl.idx.nbytes + wl.shares.nbytes) / 1e6, 1); res["bytes_out_MB"] = round(sig.nbytes / 1e6, 1)
res["combine_per_s_host_buffers"] = round(B / res["host_buffers_wall_ms"] * 1e3)
print(json.dumps(res))
