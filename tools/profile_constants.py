#!/usr/bin/env python3
"""profiles/<tag>_rocprofv3_summary.csv (tools/capture_r02.sh) -> profiles/profile_constants.json: the per-launch
L2<->fabric traffic and VALU instruction counts bench.py prints next to its live timings.
usage: python tools/profile_constants.py profiles/r02_a_head_rocprofv3_summary.csv [units]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
units = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
# per key: alternatives in order of preference; an alternative is one kernel or a tuple of kernels whose counters are summed
# (the pairing check runs as k_miller_loop + k_final_exp since round 3)
# (r04: k_miller_lines + k_miller_accumulate + k_final_exp above 16 384 checks)
KERNELS = {"combine_g2_t3": ("k_combine_fast<tc::Fq2>", "k_combine<tc::Fq2>"),
           "pairing_check": (("k_miller_lines", "k_miller_accumulate", "k_final_exp"), ("k_miller_loop", "k_final_exp"), "k_pairing_check")}
vals = {}
for line in open(src):
    f = line.rstrip("\n").split(",")
    if len(f) == 5 and f[1] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_LDS",
                                "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        vals[(f[0].replace("void ", "").strip(), f[1])] = float(f[3])     # avg per launch
out = {"static_mad_share": {"fq2p_mul_call": round(588 / 732, 3), "fq2p_sqr_call": round(392 / 522, 3), "fq_mul_call": round(392 / 479, 3),
                            "is": "v_mad / all instructions of the out-of-line multiplier bodies (llvm-objdump of the shipped code object)"}}
for key, names in KERNELS.items():
    for alt in names:
        parts = alt if isinstance(alt, tuple) else (alt,)
        ks = ["tc::" + n for n in parts]
        if all((k, "FETCH_SIZE") in vals for k in ks):
            tot = lambda c: sum(vals.get((k, c), 0) for k in ks)
            fetch_kb, write_kb = tot("FETCH_SIZE"), tot("WRITE_SIZE")
            out[key] = {"kernel": " + ".join(parts), "units": units, "source": os.path.relpath(src, ROOT),
                        "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
                        "traffic_bytes": int((2 * fetch_kb + write_kb) * 1024),
                        "sq_insts_valu": int(tot("SQ_INSTS_VALU")),
                        "sq_wait_any_frac": round(tot("SQ_WAIT_ANY") / max(tot("SQ_WAVE_CYCLES"), 1), 4),
                        "sq_insts_lds": int(tot("SQ_INSTS_LDS")),
                        "sq_insts_scratch": int(tot("SQ_INSTS_VMEM_RD") + tot("SQ_INSTS_VMEM_WR"))}
            break
json.dump(out, open(os.path.join(ROOT, "profiles", "profile_constants.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
