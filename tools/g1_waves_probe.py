"""Threshold decryption's combination (tc_decrypt_batch, G1, t = 3) at batches above one wave per SIMD: the 512-register
build of k_combine_fast<Fq> (one wave per SIMD whatever the batch) against the 256-register build (two waves per SIMD).
usage: TC_G1_WAVES=1|2 python tools/g1_waves_probe.py [B ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.workload import key_set, signer_subsets
from threshold_crypto_amd.api import _G1_GEN
e = Engine(0); e.set_timing(True); e.set_input_checks(False)
t, N = 3, 10
sks = key_set(t)
fr = np.stack([np.frombuffer(sks.secret_key_share(i)._bytes(), dtype=np.uint8) for i in range(N)])
u, _ = e.g1_mul(np.arange(1, 33, dtype=np.uint8).reshape(1, 32) % 61, np.frombuffer(_G1_GEN, dtype=np.uint8)[None].copy())
allsh, st = e.g1_mul(fr, np.ascontiguousarray(u[:, 0]))
master, _ = e.g1_mul(np.frombuffer(sks.poly[0].to_bytes(32, "little"), dtype=np.uint8)[None].copy(), np.ascontiguousarray(u[:, 0]))
res = {"waves": os.environ.get("TC_G1_WAVES", "auto")}
for B in [int(x) for x in sys.argv[1:]] or [65536, 131072, 262144]:
    idx = np.tile(signer_subsets(4096, N, t), (B // 4096, 1))
    shares = np.ascontiguousarray(allsh[0][idx.astype(np.int64)])
    for rep in range(2):
        out, st = e.combine_g1(t, idx, shares); ms = e.last_kernel_ms()
    assert not st.any() and (out == master[0, 0]).all()
    res["B%d_ms" % B] = round(ms, 3); res["B%d_per_s" % B] = round(B / (ms * 1e-3))
print(json.dumps(res), flush=True)
