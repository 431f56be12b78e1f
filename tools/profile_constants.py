#!/usr/bin/env python3
"""profiles/<tag>_<leg>_rocprofv3_summary.csv (tools/capture_legs.sh: one capture per leg of the bench line, the leg's timed
launches only) -> profiles/profile_constants.json: per leg the L2<->fabric traffic, VALU instruction count and kernel time PER
LAUNCH that bench.py prints next to its live timings (`traffic`, `executed_cross_check`, `profile_kernel_ms`).
usage: python tools/profile_constants.py <tag> [units]        e.g. r05_c"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
units = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
# leg -> (capture the figures come from, the kernels of ONE step of the leg, units per launch relative to the batch)
LEGS = {
    "combine_g2_t3": ("combine", ("k_combine_fast<tc::Fq2>", "k_combine_classify", "k_combine_scatter"), 1),
    "pairing_check": ("verify_g2", ("k_miller_lines", "k_miller_accumulate", "k_final_exp"), 1),
    "hash_g2": ("hash_g2", ("k_hash_g2",), 1),
    "g2_sign": ("g2_sign", ("k_g2_mul_shared",), 4),
    "threshold_decrypt": ("threshold_decrypt", ("k_combine_fast_g1_arena", "k_xor_with_hash"), 1),
    "ciphertext_verify": ("ciphertext_verify", ("k_hash_g1_g2", "k_miller_lines", "k_miller_accumulate", "k_final_exp"), 1),
    "wire": ("wire", ("k_decompress_take_g2_x2", "k_combine_fast<tc::Fq2>", "k_combine_classify", "k_combine_scatter", "k_compress<tc::Fq2>"), 1),
    "general_path": ("general_path", ("k_lagrange", "k_msm_tables", "k_msm_ladder"), 1),
    # BASELINE config 5, one rank's slice (131 072 jobs, t = 67: 68 signers per message) -- `python bench.py --config 5` under the
    # profiler; every kernel of the step runs once per step, so the per-kernel averages of the whole process are per-step figures
    "config5_sign": ("config5", ("k_comb_tables", "k_comb_sign"), 2 * 68),
    "config5_combine": ("config5", ("k_lagrange_den", "k_lagrange_finish", "k_msm_tables", "k_msm_ladder"), 2),
    "config5_verify": ("config5", ("k_miller_lines", "k_miller_accumulate", "k_final_exp"), 2),
}
COUNTERS = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_LDS",
            "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU")


def read(path):
    """per (kernel, counter): (launches, sum over them); per kernel: (launches, total ms) of the stats pass (the file's first table)"""
    vals, ms = {}, {}
    for line in open(path):
        f = line.rstrip("\n").split(",")
        name = f[0].replace("void ", "").strip()
        if len(f) == 5 and f[1] in COUNTERS:
            vals[(name, f[1])] = (int(f[2]), float(f[4]))
        elif len(f) == 12 and f[1].isdigit() and name not in ms:
            ms[name] = (int(f[1]), float(f[2]))
    return vals, ms


out = {"static_mad_share": {"fq2p_mul_call": round(588 / 732, 3), "fq2p_sqr_call": round(392 / 522, 3), "fq_mul_call": round(392 / 479, 3),
                            "is": "v_mad / all instructions of the out-of-line multiplier bodies (llvm-objdump of the shipped code object)"}}
for key, (leg, kernels, mult) in LEGS.items():
    path = os.path.join(ROOT, "profiles", "%s_%s_rocprofv3_summary.csv" % (tag, leg))
    if not os.path.exists(path):
        continue
    vals, ms = read(path)
    ks = ["tc::" + n for n in kernels]
    have = [k for k in ks if (k, "FETCH_SIZE") in vals]
    if not have:
        continue
    # one STEP of the leg may launch a kernel more than once (tiles): sums over all launches / steps, steps = the launches of the
    # kernel that runs once per step (the fewest launches among the leg's kernels)
    def tot(c):
        have_c = [vals[(k, c)] for k in ks if (k, c) in vals]
        if not have_c:
            return 0.0
        steps = min(n for n, _ in have_c)
        return sum(sm for _, sm in have_c) / steps
    fetch_kb, write_kb = tot("FETCH_SIZE"), tot("WRITE_SIZE")
    ms_have = [ms[k] for k in ks if k in ms]
    steps_ms = min(n for n, _ in ms_have) if ms_have else 1
    out[key] = {"kernel": " + ".join(kernels), "units": units * mult, "source": os.path.relpath(path, ROOT),
                "profile_kernel_ms": round(sum(tm for _, tm in ms_have) / steps_ms, 3),
                "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
                "traffic_bytes": int((2 * fetch_kb + write_kb) * 1024),
                "traffic_is": "2 x FETCH_SIZE + WRITE_SIZE (KB counters, FETCH doubled per the gfx950 note of the micro-architecture guide), separate PMC passes, averaged over the leg's timed launches",
                "sq_insts_valu": int(tot("SQ_INSTS_VALU")),
                "sq_wait_any_frac": round(tot("SQ_WAIT_ANY") / max(tot("SQ_WAVE_CYCLES"), 1), 4),
                "sq_insts_lds": int(tot("SQ_INSTS_LDS")),
                "sq_insts_scratch": int(tot("SQ_INSTS_VMEM_RD") + tot("SQ_INSTS_VMEM_WR"))}
json.dump(out, open(os.path.join(ROOT, "profiles", "profile_constants.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
