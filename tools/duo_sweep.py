"""One job per lane pair against two (tc_duo.h) by batch size: the checked G2 decode, hash_g2, hash_g1_g2.
    python tools/duo_sweep.py        -> profiles/r05_duo_sweep.txt (one JSON line per op and batch)
TC_DUO_MIN is read when a context is created (csrc/tc_launch.h Tuning): one context per form, same process, same data."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine, pack_messages
from threshold_crypto_amd.workload import ThresholdSigWorkload
os.environ["TC_DUO_MIN"] = str(10 ** 12); e = Engine(0)      # one job per lane pair at every size
os.environ["TC_DUO_MIN"] = "1"; e2 = Engine(0)                 # two jobs per lane pair at every size
os.environ.pop("TC_DUO_MIN")
for x in (e, e2):
    x.set_timing(True); x.set_input_checks(False)
wl = ThresholdSigWorkload(e, 3, 10, 65536)
comp, st = e.g2_compress(np.ascontiguousarray(wl.shares.reshape(-1, 192)))   # 262 144 compressed signature shares
assert not st.any()
g1 = np.tile(wl.master_pk[None], (262144, 1))


def timed(fn, eng):
    ts = []
    for _ in range(3):
        out = fn(eng); ts.append(eng.last_kernel_ms())
    return min(ts), out


for B in (16384, 32768, 65536, 131072, 262144):
    msgs = [b"tc/sweep/%08d" % i for i in range(B)]
    blob, off = pack_messages(msgs)
    for name, fn in (("g2_decompress", lambda x: x.g2_decompress(comp[:B])[0]), ("hash_g2", lambda x: x.hash_g2(blob, off)),
                     ("hash_g1_g2", lambda x: x.hash_g1_g2(g1[:B], blob, off)[0])):
        one, a = timed(fn, e)
        two, b = timed(fn, e2)
        assert (a == b).all()
        print(json.dumps({"op": name, "jobs": B, "one_job_per_pair_ms": round(one, 3), "two_jobs_per_pair_ms": round(two, 3),
                          "ratio": round(two / one, 3)}), flush=True)
