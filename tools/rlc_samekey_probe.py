"""Same-key signature batch (BASELINE config 3's shape): per-job pairing checks vs the opt-in random-linear-combination
path (tc_verify_g2_rlc_batch), all valid and with a bad signature in 1 of 256 groups.  usage: python tools/rlc_samekey_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
B = int(os.environ.get("PROBE_B", "65536"))
e = Engine(0); e.set_timing(True)
wl = ThresholdSigWorkload(e, 3, 10, B)
sig, st = e.combine_g2(3, wl.idx, wl.shares)
dev = torch.device("cuda", 0)
d_pk, d_sig, d_h = torch.from_numpy(wl.master_pk).to(dev), torch.from_numpy(sig).to(dev), torch.from_numpy(wl.hashes).to(dev)
bad = sig.copy(); bad[::16384] = sig[1]
d_bad = torch.from_numpy(bad).to(dev)
res = {"B": B}
for checks in (True, False):
    e.set_input_checks(checks)
    tag = "checked" if checks else "trusted"
    for rep in range(2):
        ok = e.verify_g2(d_pk, d_sig, d_h); res["per_job_ms_" + tag] = round(e.last_kernel_ms(), 3)
    assert bool(ok.all())
    for group in (64, 256):
        for rep in range(2):
            ok, nfb = e.verify_g2_rlc(d_pk, d_sig, d_h, group=group); res["rlc_g%d_ms_%s" % (group, tag)] = round(e.last_kernel_ms(), 3)
        assert bool(ok.all()) and nfb == 0
    for rep in range(2):
        ok, nfb = e.verify_g2_rlc(d_pk, d_bad, d_h, group=64); res["rlc_g64_4bad_ms_" + tag] = round(e.last_kernel_ms(), 3)
    assert int(ok.sum()) == B - 4 and nfb == 4 * 64
res["speedup_trusted_g64"] = round(res["per_job_ms_trusted"] / res["rlc_g64_ms_trusted"], 2)
res["speedup_checked_g64"] = round(res["per_job_ms_checked"] / res["rlc_g64_ms_checked"], 2)
res["rlc_verifies_per_s_trusted_g64"] = round(B / (res["rlc_g64_ms_trusted"] * 1e-3))
res["rlc_verifies_per_s_checked_g64"] = round(B / (res["rlc_g64_ms_checked"] * 1e-3))
print(json.dumps(res), flush=True)
