"""Times combine/verify/sign kernels under different HIP-runtime / scratch settings.
usage: python tools/scratch_probe.py [torch|notorch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
if mode == "notorch":
    sys.modules["torch"] = None  # make `import torch` fail inside _native.load(): system ROCm runtime
import numpy as np
from threshold_crypto_amd.engine import Engine
from threshold_crypto_amd.api import _G1_GEN
e = Engine(0); e.set_timing(True)
B = int(os.environ.get("PROBE_B", "65536"))
u8 = lambda b: np.frombuffer(bytes(b), dtype=np.uint8).copy()
sk = np.zeros((4, 32), np.uint8); sk[:, 0] = [3, 5, 7, 11]; sk[:, 17] = [9, 8, 7, 6]; sk[:, 30] = 0x21
g2 = bytes.fromhex("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb80606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801")
pts = np.tile(u8(g2), (B, 1))
res = {"mode": mode, "B": B, "HSA_SCRATCH_SINGLE_LIMIT": os.environ.get("HSA_SCRATCH_SINGLE_LIMIT")}
for rep in range(2):
    shares, st = e.g2_mul(sk, pts); res["sign4xB_ms_%d" % rep] = round(e.last_kernel_ms(), 2)
    idx = np.tile(np.array([0, 1, 2, 3], np.uint64), (B, 1)); idx[1::2] = [1, 4, 6, 9]
    out, st = e.combine_g2(3, idx, np.ascontiguousarray(shares)); res["combine_ms_%d" % rep] = round(e.last_kernel_ms(), 2)
    ok = e.verify_g2(u8(_G1_GEN), np.ascontiguousarray(out), pts); res["verify_ms_%d" % rep] = round(e.last_kernel_ms(), 2)
print(res, flush=True)
