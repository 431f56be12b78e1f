#!/usr/bin/env python3
"""Static census of a gfx950 code object: per function the instruction counts (all / VALU / v_mov / scratch / calls)
and, for every loop (a backward branch), the same counts for its body -- the numbers behind DESIGN.md 4.1 / 5.2
("1 283 instructions per round", "per column 1 475 VALU inline + 22 calls").

    tools/kernel_meta.sh k_msm                # leaves /tmp/tc_meta/k_msm.elf
    python tools/isa_loops.py k_msm [regex]   # functions whose mangled name matches the regex (default: all)
"""
import collections
import re
import subprocess
import sys

unit = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "/tmp/tc_meta/%s.elf" % unit], check=True, capture_output=True,
                     text=True).stdout.split("\n")
funcs, cur = collections.OrderedDict(), None
for line in dis:
    m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
    if m:
        cur = funcs.setdefault(m.group(1), [])
        continue
    if cur is not None and re.search(r"//\s*[0-9A-F]+:", line):
        cur.append(line)


def census(seg):
    ops = [x.split()[0] for x in seg]
    return {"total": sum(1 for o in ops if o != "s_nop"), "valu": sum(1 for o in ops if o.startswith("v_")),
            "mov": sum(1 for o in ops if o.startswith("v_mov") or o.startswith("v_accvgpr")),
            "mad": sum(1 for o in ops if o.startswith("v_mad_")),
            "scratch": sum(1 for o in ops if o.startswith("scratch_")), "global": sum(1 for o in ops if o.startswith("global_")),
            "calls": sum(1 for o in ops if o.startswith("s_swappc"))}


for name, body in funcs.items():
    if pat and not pat.search(name):
        continue
    if not body:
        continue
    print(name[:120], census(body))
    addrs = [int(re.search(r"//\s*([0-9A-F]+):", x).group(1), 16) for x in body]
    for i, line in enumerate(body):
        m = re.search(r"(s_cbranch_\w+|s_branch)\s+\S+\s+//.*<[^+>]+\+0x([0-9a-f]+)>", line)
        if not m:
            continue
        tgt = addrs[0] + int(m.group(2), 16)
        if tgt < addrs[i] and tgt in addrs:
            j = addrs.index(tgt)
            if i - j >= 64:
                print("    loop %5d..%5d %s" % (j, i, census(body[j:i + 1])))
