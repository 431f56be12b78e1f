import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from threshold_crypto_amd.engine import Engine
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import tc_oracle as o
e = Engine()
B = 32
def run(tag, pts, msgs):
    flat = np.frombuffer(b"".join(msgs), np.uint8).copy()
    off = np.zeros(B + 1, np.uint64); off[1:] = np.cumsum([len(m) for m in msgs])
    g1 = np.stack([np.frombuffer(o.g1_uncompressed(P), np.uint8) for P in pts])
    h, st = e.hash_g1_g2(g1, flat, off)
    bad = [i for i in range(B) if bytes(h[i]) != o.g2_uncompressed(o.hash_g1_g2(pts[i], msgs[i]))]
    print(tag, "bad:", bad)
P0 = o.E1.mul(o.G1_GEN, 5)
pts = [o.E1.mul(o.G1_GEN, 5 + i) for i in range(B)]
m0 = b"hello"
msgs = [hashlib.sha256(b"m%d" % i).digest()[:5] for i in range(B)]
run("same pt, same msg ", [P0] * B, [m0] * B)
run("same pt, diff msg ", [P0] * B, msgs)
run("diff pt, same msg ", pts, [m0] * B)
run("diff pt, diff msg ", pts, msgs)
