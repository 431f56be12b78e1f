#!/usr/bin/env python3
"""Post-processes the gfx950 assembly hipcc emits for one translation unit: the out-of-line field multipliers stop waiting for the
caller's memory operations at their ENTRY and wait for them at their EXIT instead.

Why.  LLVM's AMDGPU backend opens every non-kernel function with `s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)` (SIInsertWaitcnts: the
callee cannot know which of its input registers a pending load still owes) and, in return, lets the CALLER assume that nothing
is outstanding once a call has returned.  The kernels of this library call `fq2p_mul_call` / `fq2p_sqr_call` / `fq_mul_call` /
`fq_sqr_call` every few hundred instructions -- register-only routines of ~3 000 cycles -- so every spill reload, table look-up
or line-coefficient load issued before a product stalled the wave at the product's first instruction for its full latency
instead of landing while the product runs (14-20 % of the wave time of the pairing kernels in `s_waitcnt`).

What.  For the functions named in CALLEES (pure register routines: no memory access of their own):
  * the entry wait is removed and the same full wait is placed before every `s_setpc_b64` of the function -- the caller's
    assumption "nothing outstanding after the call" still holds;
  * before EVERY `s_swappc_b64` of the unit a wait is inserted when a load that may still be in flight writes one of the
    argument registers v0..v31 (the one case the entry wait protected: a reload straight into an argument register).  Pending
    loads into other registers are harmless: a register the callee clobbers cannot hold a value the caller needs after the
    call (IPRA), and one the callee does not touch may receive its data at any time.
The pending set at a call is reconstructed per basic block (from the previous call, full wait or label -- a label with calls
behind it gets a conservative full wait); vmcnt retires in issue order on gfx9 (the compiler's own `vmcnt(N)` waits rely on
it), FLAT accesses may not and force a full wait.

usage: python tools/asm_entry_wait.py in.s out.s      (prints a one-line summary as JSON)"""
import json
import re
import sys

CALLEES = ("fq2p_mul_call", "fq2p_sqr_call", "fq_mul_call", "fq_sqr_call")
ARG_LAST = 31
FULL = "\ts_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)"
VMEM = re.compile(r"^\s+(scratch_|global_|buffer_|tbuffer_)(load|store|atomic)")
FLAT = re.compile(r"^\s+flat_(load|store|atomic)")
LGKM = re.compile(r"^\s+(ds_|s_load|s_buffer_load|s_memtime|s_memrealtime|s_sendmsg)")
LABEL = re.compile(r"^[.\w$]+:")


def dest_range(line):
    """(lo, hi) VGPR range a load writes, or None (stores, scalar loads)"""
    m = re.match(r"^\s+(\w+)\s+(.*)$", line)
    if not m:
        return None
    op, rest = m.group(1), m.group(2)
    if "store" in op or op.startswith("s_") or op.startswith("ds_write"):
        return None
    first = rest.split(",")[0].strip()
    r = re.fullmatch(r"v(\d+)", first)
    if r:
        return (int(r.group(1)), int(r.group(1)))
    r = re.fullmatch(r"v\[(\d+):(\d+)\]", first)
    if r:
        return (int(r.group(1)), int(r.group(2)))
    return None


def hits_args(rng):
    return rng is not None and rng[0] <= ARG_LAST


def transform(lines):
    out = []
    stats = {"callees_rewritten": 0, "returns_guarded": 0, "calls": 0, "arg_waits_inserted": 0, "block_entry_waits": 0, "lgkm_waits": 0,
             "tail_calls_and_long_jumps": 0}
    in_callee = False
    expect_entry_wait = False
    pend = []          # vmcnt queue since the last sync point of this block: (is_load_into_args, is_flat)
    lgkm_args = False  # a DS / FLAT read into an argument register since the last lgkmcnt(0)
    known = False      # does `pend` describe everything that can be outstanding?  (false right after a label)
    for line in lines:
        stripped = line.rstrip("\n")
        m = re.match(r"^(_Z\w+):", stripped)
        if m:
            in_callee = any(m.group(1).startswith("_ZN2tc%d%sE" % (len(c), c)) for c in CALLEES)
            expect_entry_wait = in_callee
        if LABEL.match(stripped):
            pend, lgkm_args, known = [], False, False
            out.append(line)
            continue
        code = stripped.split(";")[0].rstrip()
        if not code.strip() or code.lstrip().startswith("."):
            out.append(line)
            continue
        ins = code.strip()
        if expect_entry_wait and ins.startswith("s_"):
            if ins.replace(" ", "") == "s_waitcntvmcnt(0)expcnt(0)lgkmcnt(0)":
                stats["callees_rewritten"] += 1
                expect_entry_wait = False
                known = True          # the caller guarantees the argument registers; nothing else concerns this routine
                continue              # dropped: re-issued before the return
            raise SystemExit("unexpected first instruction of a callee: " + ins)
        if in_callee and ins.startswith("s_setpc_b64"):
            out.append(FULL + "\n")
            stats["returns_guarded"] += 1
            out.append(line)
            continue
        is_call = ins.startswith("s_swappc_b64")
        # a tail call (Fq::inv ends in one to fq_mul_call) or a long jump: the same care for the argument registers, no sync after it
        is_jump = ins.startswith("s_setpc_b64") and "s[30:31]" not in ins
        if in_callee and (VMEM.match(code) or FLAT.match(code) or LGKM.match(code)):
            raise SystemExit("a callee of the rewritten set touches memory: " + ins)
        if ins.startswith("s_waitcnt"):
            mv = re.search(r"vmcnt\((\d+)\)", ins)
            if mv:
                k = int(mv.group(1))
                pend = pend[len(pend) - k:] if k < len(pend) else pend
                if k == 0:
                    known = True
            ml = re.search(r"lgkmcnt\((\d+)\)", ins)
            if ml and int(ml.group(1)) == 0:
                lgkm_args = False
            out.append(line)
            continue
        if FLAT.match(code):
            rng = dest_range(code)
            pend.append((hits_args(rng), True))
            lgkm_args = lgkm_args or hits_args(rng)
        elif VMEM.match(code):
            pend.append((hits_args(dest_range(code)) if "load" in ins.split()[0] or "atomic" in ins.split()[0] else False, False))
        elif LGKM.match(code):
            if ins.startswith("ds_") and not ins.startswith("ds_write") and hits_args(dest_range(code)):
                lgkm_args = True
        if is_call or is_jump:
            stats["calls" if is_call else "tail_calls_and_long_jumps"] += 1
            need_vm = None
            if not known:
                need_vm = 0           # first call after a label: what the predecessors left in flight is not visible here
                stats["block_entry_waits"] += 1
            else:
                idx = [i for i, (a, _) in enumerate(pend) if a]
                if idx:
                    need_vm = 0 if any(f for _, f in pend) else min(len(pend) - 1 - idx[-1], 62)
                    stats["arg_waits_inserted"] += 1
            if need_vm is not None or lgkm_args:
                parts = []
                if need_vm is not None:
                    parts.append("vmcnt(%d)" % need_vm)
                if lgkm_args or need_vm == 0 and not known:
                    parts.append("lgkmcnt(0)")
                    stats["lgkm_waits"] += 1
                out.append("\ts_waitcnt " + " ".join(parts) + "\n")
            out.append(line)
            pend, lgkm_args, known = [], False, is_call  # every callee waits for everything before it returns
            continue
        out.append(line)
    if stats["callees_rewritten"] and stats["returns_guarded"] < stats["callees_rewritten"]:
        raise SystemExit("a rewritten callee has no guarded return")
    return out, stats


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out, stats = transform(open(src).readlines())
    open(dst, "w").writelines(out)
    print(json.dumps(stats))


if __name__ == "__main__":
    main()
