#!/usr/bin/env python3
"""Counts the limb multiply-adds each per-lane job body executes (host build of the device source
with -DTC_COUNT_OPS).  Source of the EXECUTED table in bench.py and DESIGN.md.

One 14x14 limb product or one Montgomery reduction is 196 v_mad:
    Fq product 392, Fq squaring 301 (105 + 196), one coefficient of an Fq2 product (two products,
    one reduction, tc_field.h fq_mul2) 588.
In a G2 kernel two lanes work on a job: operations inside Fq2 methods are split between them
(counted once), Fq operations outside (inversions, square-root exponentiations) run on both lanes
(counted twice).  G1 kernels run one lane per job."""
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
    a = (ctypes.c_uint64 * 5)()
    L.hs_op_counts5(a, 1)
    return tuple(a)


def macs(c, lanes):
    mul2, smul, ssqr, mul, sqr = c
    local_mul, local_sqr = mul - smul, sqr - ssqr
    return mul2 * 588 + smul * 392 + ssqr * 301 + lanes * (local_mul * 392 + local_sqr * 301)


rnd = random.Random(1)
buf = ctypes.create_string_buffer
P2 = o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R))
P1 = o.E1.mul(o.G1_GEN, rnd.randrange(1, o.R))
res = {}
cnt()
L.hs_g2_mul(o.fr_to_bytes(rnd.randrange(o.R)), o.g2_uncompressed(P2), buf(192)); res["g2_mul"] = (cnt(), 2)
L.hs_g1_mul(o.fr_to_bytes(rnd.randrange(o.R)), o.g1_uncompressed(P1), buf(96)); res["g1_mul"] = (cnt(), 1)
L.hs_g2_mul_shared(b"".join(o.fr_to_bytes(rnd.randrange(o.R)) for _ in range(4)), 4, o.g2_uncompressed(P2), buf(768), buf(4))
res["g2_mul_4_scalars_per_point"] = (tuple(x // 4 for x in cnt()), 2)
# share generation (k_g2_mul_gather): 68 signers of a message in tables of kGatherShare scalars -- full tables + the remainder
_g = L.hs_gather_share()
_sk = b"".join(o.fr_to_bytes(rnd.randrange(o.R)) for _ in range(_g))
_idx = (ctypes.c_uint64 * _g)(*range(_g))
L.hs_g2_mul_gather(_sk, _g, _idx, _g, o.g2_uncompressed(P2), buf(192 * _g), buf(_g)); _full = cnt()
_rem = 68 % _g
if _rem:
    L.hs_g2_mul_gather(_sk, _g, _idx, _rem, o.g2_uncompressed(P2), buf(192 * _g), buf(_g)); _part = cnt()
else:
    _part = tuple(0 for _ in _full)
res["g2_mul_gather_68_signers"] = (tuple(((68 // _g) * a + b) // 68 for a, b in zip(_full, _part)), 2)
# the same 68 shares through the per-message comb (k_comb_tables + k_comb_sign: what n >= 24 signers run)
_sk68 = b"".join(o.fr_to_bytes(rnd.randrange(o.R)) for _ in range(68))
_idx68 = (ctypes.c_uint64 * 68)(*range(68))
cnt()
L.hs_comb_sign(_sk68, 68, _idx68, 68, o.g2_uncompressed(P2), buf(192 * 68), buf(68))
res["g2_sign_comb_68_signers"] = (tuple(x // 68 for x in cnt()), 2)
poly = [rnd.randrange(o.R) for _ in range(4)]
shares_g2 = {i: o.g2_uncompressed(o.E2.mul(P2, o.secret_key_share(poly, i))) for i in range(10)}
shares_g1 = {i: o.g1_uncompressed(o.E1.mul(P1, o.secret_key_share(poly, i))) for i in range(10)}


def avg_combine(g2, general, n=48):
    """average over random 4-subsets of 10 signers (the bench's distribution); a wave additionally
    pads the short ladder of the fast path to its longest job (not counted here)"""
    tot = [0] * 5
    L.hs_force_general_combine(1 if general else 0)
    for _ in range(n):
        ids = sorted(rnd.sample(range(10), 4))
        idx = (ctypes.c_uint64 * 4)(*ids)
        if g2:
            L.hs_combine_g2(3, idx, b"".join(shares_g2[i] for i in ids), buf(192))
        else:
            L.hs_combine_g1(3, idx, b"".join(shares_g1[i] for i in ids), buf(96))
        tot = [x + y for x, y in zip(tot, cnt())]
    L.hs_force_general_combine(0)
    return tuple(x // n for x in tot)


def _denominator(ids):
    """the common denominator D of the small-index fast path (tc_threshold.h lagrange_small_coeffs)"""
    from math import gcd
    xs = [i + 1 for i in ids]
    den = []
    for i in range(len(xs)):
        d = 1
        for j in range(len(xs)):
            if j != i:
                d *= xs[j] - xs[i]
        den.append(abs(d))
    D = 1
    for d in den:
        D = D * d // gcd(D, d)
    g = D
    for i, d in enumerate(den):
        c = D // d
        for j in range(len(xs)):
            if j != i:
                c *= xs[j]
        g = gcd(g, c)
    return D // g


def class_combine(pred, n=24):
    """average over random 4-subsets whose denominator satisfies pred"""
    tot, done = [0] * 5, 0
    while done < n:
        ids = sorted(rnd.sample(range(10), 4))
        if not pred(_denominator(ids)):
            continue
        idx = (ctypes.c_uint64 * 4)(*ids)
        L.hs_combine_g2(3, idx, b"".join(shares_g2[i] for i in ids), buf(192))
        tot = [x + y for x, y in zip(tot, cnt())]
        done += 1
    return tuple(x // n for x in tot)


def all_subsets_combine():
    """exact average over all C(10, 4) signer subsets (the workload draws them uniformly)"""
    import itertools
    tot, n = [0] * 5, 0
    for ids in itertools.combinations(range(10), 4):
        idx = (ctypes.c_uint64 * 4)(*ids)
        L.hs_combine_g2(3, idx, b"".join(shares_g2[i] for i in ids), buf(192))
        tot = [x + y for x, y in zip(tot, cnt())]
        n += 1
    return tuple(x // n for x in tot)


cnt()
res["combine_g2_t3_fast"] = (all_subsets_combine(), 2)
_pow2 = lambda d: d & (d - 1) == 0
res["combine_g2_t3_fast_general_denominator"] = (class_combine(lambda d: not _pow2(d)), 2)
res["combine_g2_t3_fast_denominator_1"] = (class_combine(lambda d: d == 1, 8), 2)
res["combine_g2_t3_fast_denominator_pow2"] = (class_combine(lambda d: d > 1 and _pow2(d), 8), 2)
res["combine_g2_t3_general"] = (avg_combine(True, True, 8), 2)
res["combine_g1_t3_fast"] = (avg_combine(False, False), 1)
res["combine_g1_t3_general"] = (avg_combine(False, True, 8), 1)
# BASELINE config 5 shape: one t = 67 combination through the two-stage path (k_lagrange_all's Fr work is not
# v_mad_i64_i32 work and is not counted: the saturated 8 x 32 Fr multiplier issues v_mad_u64_u32)
poly67 = [rnd.randrange(o.R) for _ in range(68)]
ids67 = sorted(rnd.sample(range(200), 68))
_sh = []
for i in ids67:
    b_ = buf(192)
    L.hs_g2_mul(o.fr_to_bytes(o.secret_key_share(poly67, i)), o.g2_uncompressed(P2), b_)
    _sh.append(b_.raw)
cnt()
_out = buf(192)
assert L.hs_combine_g2(67, (ctypes.c_uint64 * 68)(*ids67), b"".join(_sh), _out) == 0
res["combine_g2_t67_msm"] = (cnt(), 2)
assert _out.raw == o.g2_uncompressed(o.E2.mul(P2, poly67[0]))
# the same shape in G1 (threshold decryption at t = 67): tc_msm.h job_msm_tables_g1 + job_msm_ladder_g1, one lane per job
L.hs_msm_g1.argtypes = [ctypes.c_size_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
_lam = o.lagrange_coeffs(67, [o.into_fr_plus_1(i) for i in ids67])
_sh1 = b"".join(o.g1_uncompressed(o.E1.mul(P1, o.secret_key_share(poly67, i))) for i in ids67)
_words = (ctypes.c_uint32 * (8 * 68))(*[(k >> (32 * i)) & 0xffffffff for k in _lam for i in range(8)])
cnt()
_out1 = buf(96)
assert L.hs_msm_g1(68, _sh1, _words, _out1, 1) == 0 and _out1.raw == o.g1_uncompressed(o.E1.mul(P1, poly67[0]))
res["combine_g1_t67_msm"] = (cnt(), 1)
_m = b"tc/enc-payload-0123456789abcdef0123456789abcdef"[:32]
assert L.hs_hash_g1_g2(o.g1_uncompressed(P1), _m, ctypes.c_size_t(len(_m)), buf(192)) == 0
res["hash_g1_g2"] = (cnt(), 2)
a = rnd.randrange(o.R)
L.hs_pairing_check(o.g1_uncompressed(o.E1.mul(o.G1_GEN, a)), o.g2_uncompressed(P2), o.g1_uncompressed(o.G1_GEN),
                   o.g2_uncompressed(o.E2.mul(P2, a))); res["verify_g2"] = (cnt(), 2)
# the prepared form (k_miller_lines + k_miller_accumulate + k_final_exp: what batches above 16 384 checks run)
L.hs_pairing_check_prepared(o.g1_uncompressed(o.E1.mul(o.G1_GEN, a)), o.g2_uncompressed(P2), o.g1_uncompressed(o.G1_GEN),
                            o.g2_uncompressed(o.E2.mul(P2, a))); res["verify_g2_prepared"] = (cnt(), 2)
tot = [0] * 5
N = 64
for j in range(N):
    m = b"tc/msg" + j.to_bytes(8, "little")
    L.hs_hash_g2(m, len(m), buf(192))
    tot = [x + y for x, y in zip(tot, cnt())]
res["hash_g2"] = (tuple(x // N for x in tot), 2)
L.hs_g1_fixed_base_mul(o.fr_to_bytes(1), buf(96)); cnt()   # (builds the host copy of the window table)
L.hs_g1_fixed_base_mul(o.fr_to_bytes(rnd.randrange(o.R)), buf(96)); res["g1_fixed_base_mul"] = (cnt(), 1)
L.hs_decompress_g2(o.g2_compressed(P2), buf(192)); res["g2_decompress"] = (cnt(), 2)
L.hs_decompress_g1(o.g1_compressed(P1), buf(96)); res["g1_decompress"] = (cnt(), 1)
# ---- two jobs per lane pair (tc_duo.h): per JOB = half of what the pair's call executes ----
_half = lambda c: tuple(x // 2 for x in c)
Q2 = o.E2.mul(o.G2_GEN, rnd.randrange(1, o.R))
cnt()
L.hs_decompress_g2_x2(o.g2_compressed(P2), o.g2_compressed(Q2), buf(192), buf(192)); res["g2_decompress_x2"] = (_half(cnt()), 2)
L.hs_hash_g2_x2.restype = None
tot = [0] * 5
for j in range(0, N, 2):
    ma, mb = b"tc/msg" + j.to_bytes(8, "little"), b"tc/msg" + (j + 1).to_bytes(8, "little")
    L.hs_hash_g2_x2(ma, ctypes.c_size_t(len(ma)), mb, ctypes.c_size_t(len(mb)), buf(192), buf(192), 1)
    tot = [x + y for x, y in zip(tot, cnt())]
res["hash_g2_x2"] = (tuple(x // N for x in tot), 2)
L.hs_hash_g1_g2_x2(o.g1_uncompressed(P1), _m, ctypes.c_size_t(len(_m)), o.g1_uncompressed(P1), _m, ctypes.c_size_t(len(_m)), buf(192), buf(192), 1)
res["hash_g1_g2_x2"] = (_half(cnt()), 2)
if "--json" in sys.argv or "--json-useful" in sys.argv:
    import json
    useful = "--json-useful" in sys.argv
    print(json.dumps({k: macs(c, 1 if useful else lanes) for k, (c, lanes) in res.items()}, indent=1))
    sys.exit(0)
print("%-40s %8s %8s %8s %8s %8s %12s %12s %6s" % ("job", "mul2", "split_m", "split_s", "all_mul", "all_sqr", "device v_mad", "useful", "dup %"))
for k, (c, lanes) in res.items():
    ex, us = macs(c, lanes), macs(c, 1)
    print("%-40s %8d %8d %8d %8d %8d %12d %12d %6.1f" % ((k,) + tuple(c) + (ex, us, 100.0 * (ex - us) / ex)))
