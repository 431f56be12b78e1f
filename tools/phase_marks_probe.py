"""Where a generic-class wave of k_combine_fast<Fq2> spends its time: shader-clock stamps at the phase boundaries
(experiment build: TC_BUILD_FLAGS=-DTC_PHASE_TIMING TC_BUILD_SUFFIX=_pt python -m threshold_crypto_amd.build;
TC_AMD_LIB=threshold_crypto_amd/libtc_amd_pt.so python tools/phase_marks_probe.py)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import ThresholdSigWorkload
from threshold_crypto_amd import _native
e = Engine(0); e.set_timing(True)
wl = ThresholdSigWorkload(e, 3, 10, 65536)
dev = torch.device("cuda", 0)
d_idx, d_sh = torch.from_numpy(wl.idx.view(np.int64)).to(dev), torch.from_numpy(wl.shares).to(dev)
for _ in range(2):
    sig, st = e.combine_g2(3, d_idx, d_sh)
e.sync(); torch.cuda.synchronize()
ms = e.last_kernel_ms()
lib = _native.load()
marks = (ctypes.c_uint64 * (4096 * 16))()
rc = lib.tc_debug_phase_marks(marks, 4096 * 16)
m = np.frombuffer(marks, dtype=np.uint64).reshape(4096, 16).astype(np.float64)
names = ["coeffs+decode(4 shares via LDS)", "short ladder (table + <=9 columns)", "division: D^-1, psi bases, SAC table", "division: 64-column ladder",
         "inversion (to affine)", "encode + store"]
out = {"kernel_ms": round(ms, 3), "rc": rc}
for lo, hi, tag in ((0, 1300, "generic_class_waves"), (1400, 1700, "D=1_class_waves"), (1750, 2040, "D=2^a_class_waves")):
    blk = m[lo:hi]
    tot = blk[:, 6] - blk[:, 0]
    ph = {}
    for k, nm in enumerate(names):
        a, b = (k, k + 1)
        d = blk[:, b] - blk[:, a]
        ph[nm] = round(float(np.median(d)) / 1e6, 3)
    out[tag] = {"median_total_Mcycles": round(float(np.median(tot)) / 1e6, 3), "phases_Mcycles": ph}
# timeline: wall-clock start / end of every workgroup (100 MHz ticks -> microseconds from the first start)
st, en = m[:2054, 8] / 100.0, m[:2054, 9] / 100.0
t0 = st.min()
st, en = st - t0, en - t0
def q(a):
    return [round(float(x), 1) for x in np.percentile(a, [0, 10, 50, 90, 100])]
out["timeline_us"] = {"kernel_span": round(float(en.max()), 1)}
for lo, hi, tag in ((0, 1024, "blocks 0-1023 (generic, first slots)"), (1024, 1372, "blocks 1024-1371 (generic, second slots)"),
                    (1372, 1702, "D=1 class"), (1702, 2054, "D=2^a class")):
    out["timeline_us"][tag] = {"start p0/10/50/90/100": q(st[lo:hi]), "end p0/10/50/90/100": q(en[lo:hi]), "duration p0/10/50/90/100": q(en[lo:hi] - st[lo:hi])}
print(json.dumps(out, indent=1))
