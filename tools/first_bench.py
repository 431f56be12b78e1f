"""Scratch timing of the main kernels at growing batch sizes (kernel ms via HIP events)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import tc_oracle as o
from threshold_crypto_amd.engine import Engine, pack_messages
e = Engine(0); e.set_timing(True)
u8 = lambda b: np.frombuffer(bytes(b), dtype=np.uint8).copy()
sk = np.stack([u8(o.fr_to_bytes(0x1234567 + 977 * i * 2**200)) for i in range(10)])
for B in (4096, 16384, 65536):
    base = u8(o.g2_uncompressed(o.G2_GEN))
    pts = np.tile(base, (B, 1))
    # distinct points: multiply generator by per-job scalars on the GPU itself
    ks = np.zeros((B, 32), np.uint8); ks[:, :4] = np.arange(1, B + 1, dtype=np.uint32).view(np.uint8).reshape(B, 4)
    t0 = time.time(); h, st = e.g2_mul(sk[:1], pts); print("B", B, "g2_mul 1xB ms", e.last_kernel_ms(), "wall", time.time() - t0, flush=True)
    t0 = time.time(); shares, st = e.g2_mul(sk[:4], h[:, 0]); ms = e.last_kernel_ms(); print("  sign 4xB kernel ms", ms, "muls/s", 4 * B / ms * 1e3, flush=True)
    idx = np.tile(np.array([0, 1, 2, 3], np.uint64), (B, 1))
    out, st = e.combine_g2(3, idx, np.ascontiguousarray(shares)); ms = e.last_kernel_ms(); print("  combine t=3 kernel ms", ms, "combine/s", B / ms * 1e3, "status", int(st.any()), flush=True)
    pk = u8(o.g1_uncompressed(o.E1.mul(o.G1_GEN, 5)))
    ok = e.verify_g2(pk, np.ascontiguousarray(out), np.ascontiguousarray(h[:, 0])); ms = e.last_kernel_ms(); print("  verify_g2 kernel ms", ms, "verifies/s", B / ms * 1e3, "ok_sum", int(ok.sum()), flush=True)
    if B <= 16384:
        msgs = [b"tc/msg" + int(j).to_bytes(8, "little") for j in range(B)]
        flat, off = pack_messages(msgs)
        hh = e.hash_g2(flat, off); ms = e.last_kernel_ms(); print("  hash_g2 kernel ms", ms, "hash/s", B / ms * 1e3, flush=True)
