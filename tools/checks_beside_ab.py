"""Round 6 A/B of checked-input mode's second stream (tc_api.hip Call::run_checks) on the BASELINE batch of config 2 (t = 3, N = 10,
65 536 jobs, device-resident operands, the context's DEFAULT membership tests on): one tc_combine_g2_batch call with
    TC_CHECKS_BESIDE=0   the tests before the combination, one stream (rounds 2-5)
    TC_CHECKS_BESIDE=1   the tests on a LOW-priority second stream, released where the main stream reaches k_combine_fast (shipped)
    TC_CHECKS_BESIDE=2   the same on a normal-priority stream
each on the context's own stream and on a torch-owned stream (what bench.py hands the context), four fresh contexts per variant,
[min, median] of nine calls in ms.    python tools/checks_beside_ab.py  -> profiles/r06_checks_beside_ab.txt"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
dev = torch.device("cuda", 0)
t, N, B = 3, 10, 65536
gen = Engine(0); gen.set_input_checks(False)
wl = ThresholdSigWorkload(gen, t, N, B)
d_idx = torch.from_numpy(wl.idx.view(np.int64)).to(dev); d_sh = torch.from_numpy(wl.shares).to(dev)
def med(e, reps=9):
    ts = []
    for _ in range(reps):
        e.sync(); torch.cuda.synchronize(); t0 = time.perf_counter(); e.combine_g2(t, d_idx, d_sh); e.sync(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return round(ts[0] * 1e3, 2), round(ts[len(ts) // 2] * 1e3, 2)
res = {}
keep = []
for trial in range(4):
    for mode in ("0", "1", "2"):
        os.environ["TC_CHECKS_BESIDE"] = mode
        for main in ("own", "torch"):
            e = Engine(0); e.set_timing(False); e.set_input_checks(True)
            if main == "torch":
                s = torch.cuda.Stream(device=dev); e.set_stream(s.cuda_stream); keep.append(s)
            e.combine_g2(t, d_idx, d_sh); e.sync()
            res.setdefault("beside%s_%s" % (mode, main), []).append(med(e))
            e.close()
print(json.dumps(res))
