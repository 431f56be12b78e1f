#!/usr/bin/env python3
"""Counts the Fq products / squarings each per-lane job body executes (host build of the device
source with -DTC_COUNT_OPS).  Source of the EXECUTED table in bench.py and DESIGN.md 4.2."""
import ctypes
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tc_oracle as o  # noqa: E402

lib = os.path.join(ROOT, "tests", "hostsim", "libtc_hostsim_cnt.so")
subprocess.run(["g++", "-O2", "-std=c++17", "-DTC_COUNT_OPS", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "threshold_crypto_amd", "csrc"),
                os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp"), "-o", lib], check=True)
L = ctypes.CDLL(lib)


def cnt():
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    L.hs_op_counts(ctypes.byref(a), ctypes.byref(b), 1)
    return a.value, b.value


rnd = random.Random(1)
buf = ctypes.create_string_buffer
P2 = o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R))
P1 = o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R))
res = {}
cnt()
L.hs_g2_mul(o.fr_to_bytes(rnd.randrange(o.R)), o.g2_uncompressed(P2), buf(192)); res["g2_mul"] = cnt()
L.hs_g1_mul(o.fr_to_bytes(rnd.randrange(o.R)), o.g1_uncompressed(P1), buf(96)); res["g1_mul"] = cnt()
poly = [rnd.randrange(o.R) for _ in range(4)]
ids = [1, 4, 6, 9]
idx = (ctypes.c_uint64 * 4)(*ids)
sh = b"".join(o.g2_uncompressed(o.E2.mul(P2, o.secret_key_share(poly, i))) for i in ids)
cnt()
L.hs_combine_g2(3, idx, sh, buf(192)); res["combine_g2_t3_fast"] = cnt()
L.hs_force_general_combine(1)
L.hs_combine_g2(3, idx, sh, buf(192)); res["combine_g2_t3_general"] = cnt()
L.hs_force_general_combine(0)
sh1 = b"".join(o.g1_uncompressed(o.E1.mul(P1, o.secret_key_share(poly, i))) for i in ids)
L.hs_combine_g1(3, idx, sh1, buf(96)); res["combine_g1_t3"] = cnt()
a = rnd.randrange(o.R)
L.hs_pairing_check(o.g1_uncompressed(o.E1.mul(o.G1_GEN, a)), o.g2_uncompressed(P2), o.g1_uncompressed(o.G1_GEN),
                   o.g2_uncompressed(o.E2.mul(P2, a))); res["pairing_check"] = cnt()
hs = []
for j in range(40):
    m = b"tc/msg" + j.to_bytes(8, "little")
    L.hs_hash_g2(m, len(m), buf(192)); hs.append(cnt())
res["hash_g2_avg40"] = (sum(h[0] for h in hs) // 40, sum(h[1] for h in hs) // 40)
L.hs_decompress_g2(o.g2_compressed(P2), buf(192)); res["g2_decompress"] = cnt()
L.hs_decompress_g1(o.g1_compressed(P1), buf(96)); res["g1_decompress"] = cnt()
for k, (m, s) in res.items():
    print("%-24s products %6d  squarings %5d  total %6d  v_mad %9d" % (k, m, s, m + s, m * 450 + s * 345))
