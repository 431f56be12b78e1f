#!/bin/bash
# Kernel resource metadata (scratch bytes per lane, spills, LDS bytes, registers) of one translation
# unit, read from the gfx950 code object:  tools/kernel_meta.sh k_combine [extra hipcc flags]
# Output: one line per kernel; also leaves /tmp/tc_meta/<unit>.elf for llvm-objdump -d.
set -e
unit=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/tc_meta; mkdir -p $out
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc --cuda-device-only "$@" -c $root/threshold_crypto_amd/csrc/$unit.hip -o $out/$unit.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$out/$unit.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$out/$unit.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out/$unit.elf | python3 -c '
import sys, re
cur = {}
rows = []
for line in sys.stdin:
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count" and cur.get("name"):
        rows.append(cur); cur = {}
    cur[k] = v
    if k == "wavefront_size":
        rows.append(cur); cur = {}
for r in rows:
    if "name" not in r: continue
    print("%-60s scratch %6s B/lane  lds %6s B  vgpr %4s agpr %4s  vgpr_spill %5s sgpr_spill %4s" % (
        r["name"][:60], r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size"), r.get("vgpr_count"),
        r.get("agpr_count"), r.get("vgpr_spill_count"), r.get("sgpr_spill_count")))
'
