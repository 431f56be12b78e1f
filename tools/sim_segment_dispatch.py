#!/usr/bin/env python3
"""Would cutting the expensive ladders of the headline launch into segments, dispatched as separate workgroups in
segment-major order (workgroup s*C + c runs segment s of chain c once segment s-1 has finished), balance the SIMDs?

Model: 1024 SIMDs x 2 wave slots; a wave advances at 1.0 alone and at 0.667 beside another RUNNING wave (two
co-resident waves = 1.33 x one); a wave that waits for its predecessor does not slow its sibling; workgroups are dispatched in
index order into free slots.  C = 1372 chains of W = 2.03 ms of lone-wave work (the 64-column ladder of the expensive class).

Answer: no -- the makespan stays W / 0.667 for every segment count.  1372 runners + 676 waiters fill the 2048 slots, a runner
sits beside another runner with probability 0.67, and with 1372 chains some chain is paired in EVERY segment; re-placement at
random evens out the MEAN (0.78 of the lone speed), not the slowest chain.  Only a scheduler that places the laggards alone
would (persistent workers + a least-progress-first queue) -- and the second model below says what that is worth: a greedy
least-progress-first assignment at every segment boundary (laggards onto SIMDs whose other slot is idle, the most advanced
chains paired) ends the ladder phase at 2.75-2.8 ms instead of 3.04 (-8 %, about -5 % of the whole launch): priced in
DESIGN.md 5.2, not built."""
import sys


def sim(C, S, W, nsimd=1024, paired=0.667, dt=0.002):
    seg = W / S
    slots = [[None, None] for _ in range(nsimd)]
    done_seg = [-1] * C
    nxt, total, finished, t = 0, C * S, 0, 0.0
    free = [(i, k) for k in range(2) for i in range(nsimd)]
    free.reverse()
    while finished < total:
        while nxt < total and free:
            i, k = free.pop()
            slots[i][k] = [nxt % C, nxt // C, seg]
            nxt += 1
        for i in range(nsimd):
            a, b = slots[i]
            ra = a is not None and done_seg[a[0]] >= a[1] - 1
            rb = b is not None and done_seg[b[0]] >= b[1] - 1
            rate = paired if (ra and rb) else 1.0
            if ra:
                a[2] -= rate * dt
            if rb:
                b[2] -= rate * dt
        t += dt
        for i in range(nsimd):
            for k in range(2):
                w = slots[i][k]
                if w is not None and w[2] <= 0:
                    done_seg[w[0]] = max(done_seg[w[0]], w[1])
                    finished += 1
                    slots[i][k] = None
                    free.append((i, k))
    return t


def sim_least_progress_first(C, S, W, nsimd=1024, paired=0.667, dt=0.002):
    seg = W / S
    prog = [0] * C
    running = [False] * C
    slots = [[None, None] for _ in range(nsimd)]
    t, done = 0.0, 0
    while done < C:
        idle = sorted((c for c in range(C) if not running[c] and prog[c] < S), key=lambda c: prog[c])
        if idle:
            empty = [i for i in range(nsimd) if slots[i][0] is None and slots[i][1] is None]
            half = [(i, k) for i in range(nsimd) for k in range(2) if slots[i][k] is None and slots[i][1 - k] is not None]
            for c in idle:
                if empty:
                    i = empty.pop()
                    slots[i][0] = [c, seg]
                    half.append((i, 1))
                    running[c] = True
                elif half:
                    i, k = half.pop(0)
                    slots[i][k] = [c, seg]
                    running[c] = True
        for i in range(nsimd):
            a, b = slots[i]
            rate = paired if (a is not None and b is not None) else 1.0
            if a is not None:
                a[1] -= rate * dt
            if b is not None:
                b[1] -= rate * dt
        t += dt
        for i in range(nsimd):
            for k in range(2):
                w = slots[i][k]
                if w is not None and w[1] <= 0:
                    prog[w[0]] += 1
                    running[w[0]] = False
                    slots[i][k] = None
                    if prog[w[0]] == S:
                        done += 1
    return t


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "--split"):
    W = float(sys.argv[1]) if len(sys.argv) > 1 else 2.03
    for S in (1, 2, 4, 8):
        print({"segments": S, "makespan_ms": round(sim(1372, S, W), 3), "one_wave_per_chain_ms": round(W / 0.667, 3)})
    for S in (4, 8):
        print({"policy": "least progress first", "segments": S, "makespan_ms": round(sim_least_progress_first(1372, S, W), 3)})


# ---- r05: the [1/D] ladder of a FRACTION of the expensive jobs on two lane pairs --------------------------
# A split job runs on two adjacent lane pairs of ONE wave (they merge with a DPP exchange), so the two halves execute in lockstep:
# with pair A on the top k columns (+ 64 - k positioning doublings) and pair B on the low 64 - k, the wave issues
# (64 - k) x (doubling + addition) + k x doubling -- least at k = 32: 32 x 8 820 + 32 x 3 136 = 0.68 of the 64-column ladder.  Both
# pairs run the short joint ladder and build the 8-entry psi table; one addition merges.  Per lane (multiply-adds of the generic
# class, profiles/executed_macs.json / 2): short ladder 0.128 M + table 0.075 M + ladder 0.564 M + rest = 0.775 M, so a split wave
# lasts 0.165 + 0.097 + 0.68 x 0.728 + 0.015 = 0.77 of a full one for HALF the jobs: 1.54 x the work per job, in shorter waves
# that the dispatcher can back-fill behind the cheap classes.
# Lone-wave times from the measured launch: an expensive wave paired for its whole life ends at 5.0 ms => 3.33 ms
# alone; the classes scale with their executed multiply-adds (1.55 M / 0.736 M / 0.256 M).
def sim_split(frac, split_last=True, nsimd=1024, paired=0.667, W=3.33):
    w_d1, w_p2 = W * 0.256 / 1.55, W * 0.736 / 1.55
    w_split = 0.77 * W
    n_exp, n_p2, n_d1 = 1345, 381, 322           # waves per class of the 65 536-job batch (65.7 / 18.6 / 15.7 % of 2048)
    n_split_jobs = int(round(n_exp * frac))
    unsplit = [W] * (n_exp - n_split_jobs)
    split = [w_split] * (2 * n_split_jobs)
    cheap = [w_p2] * n_p2 + [w_d1] * n_d1
    order = unsplit + cheap + split if split_last else unsplit + split + cheap
    slots = [[None, None] for _ in range(nsimd)]
    free = [(i, k) for k in (1, 0) for i in range(nsimd - 1, -1, -1)]   # slot 0 of every SIMD first, then slot 1
    nxt, t, running = 0, 0.0, 0
    while nxt < len(order) or running:
        while nxt < len(order) and free:
            i, k = free.pop()
            slots[i][k] = order[nxt]
            nxt += 1
            running += 1
        # advance to the next completion
        dt = min((w / (paired if (s[0] is not None and s[1] is not None) else 1.0)) for s in slots for w in s if w is not None)
        t += dt
        for i, s in enumerate(slots):
            rate = paired if (s[0] is not None and s[1] is not None) else 1.0
            for k in (0, 1):
                if s[k] is not None:
                    s[k] -= rate * dt
                    if s[k] <= 1e-9:
                        s[k] = None
                        free.append((i, k))
                        running -= 1
    return t


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--split":
    import json
    rows = [{"fraction_of_expensive_jobs_split": f, "launch_ms_split_waves_last": round(sim_split(f, True), 2),
             "launch_ms_split_waves_after_the_unsplit_ones": round(sim_split(f, False), 2)} for f in (0.0, 0.1, 0.2, 0.25, 0.3, 0.4, 0.5, 0.75, 1.0)]
    best = min(min(r["launch_ms_split_waves_last"], r["launch_ms_split_waves_after_the_unsplit_ones"]) for r in rows)
    print(json.dumps({"model": "1024 SIMDs x 2 slots, in-order dispatch, two co-resident waves at 0.667 each; lone expensive wave 3.33 ms",
                      "rows": rows, "best_ms": best, "gate_ms": 4.4, "built": best <= 4.4}))
    sys.exit(0)
