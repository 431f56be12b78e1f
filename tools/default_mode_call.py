"""For rocprofv3 --kernel-trace: three default-mode tc_combine_g2_batch calls at the BASELINE batch (t = 3, N = 10, 65 536 jobs, every share
tested for group membership), operands resident; tools/rocpd_timeline.py then shows the last call's kernels on their two queues."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
dev = torch.device("cuda", 0)
gen = Engine(0); gen.set_input_checks(False)
wl = ThresholdSigWorkload(gen, 3, 10, 65536)
d_idx = torch.from_numpy(wl.idx.view(np.int64)).to(dev); d_sh = torch.from_numpy(wl.shares).to(dev)
e = Engine(0); e.set_input_checks(True)
for _ in range(3):
    sig, st = e.combine_g2(3, d_idx, d_sh); e.sync()
assert not bool(st.any().item())
