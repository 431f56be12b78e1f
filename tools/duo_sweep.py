"""One job per lane pair against two (tc_duo.h) by batch size: the checked G2 decode, hash_g2, hash_g1_g2.
    python tools/duo_sweep.py        -> profiles/r05_duo_sweep.txt (one JSON line per op and batch)
TC_DUO_MIN is read by the library at every launch, so both forms run in one process on the same data."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine, pack_messages
from threshold_crypto_amd.workload import ThresholdSigWorkload
e = Engine(0); e.set_timing(True); e.set_input_checks(False)
wl = ThresholdSigWorkload(e, 3, 10, 65536)
comp, st = e.g2_compress(np.ascontiguousarray(wl.shares.reshape(-1, 192)))   # 262 144 compressed signature shares
assert not st.any()
g1 = np.tile(wl.master_pk[None], (262144, 1))


def timed(fn):
    ts = []
    for _ in range(3):
        out = fn(); ts.append(e.last_kernel_ms())
    return min(ts), out


for B in (16384, 32768, 65536, 131072, 262144):
    msgs = [b"tc/sweep/%08d" % i for i in range(B)]
    blob, off = pack_messages(msgs)
    for name, fn in (("g2_decompress", lambda: e.g2_decompress(comp[:B])[0]), ("hash_g2", lambda: e.hash_g2(blob, off)),
                     ("hash_g1_g2", lambda: e.hash_g1_g2(g1[:B], blob, off)[0])):
        os.environ["TC_DUO_MIN"] = str(10 ** 12); one, a = timed(fn)
        os.environ["TC_DUO_MIN"] = "1"; two, b = timed(fn)
        assert (a == b).all()
        print(json.dumps({"op": name, "jobs": B, "one_job_per_pair_ms": round(one, 3), "two_jobs_per_pair_ms": round(two, 3),
                          "ratio": round(two / one, 3)}), flush=True)
