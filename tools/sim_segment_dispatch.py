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


if __name__ == "__main__":
    W = float(sys.argv[1]) if len(sys.argv) > 1 else 2.03
    for S in (1, 2, 4, 8):
        print({"segments": S, "makespan_ms": round(sim(1372, S, W), 3), "one_wave_per_chain_ms": round(W / 0.667, 3)})
    for S in (4, 8):
        print({"policy": "least progress first", "segments": S, "makespan_ms": round(sim_least_progress_first(1372, S, W), 3)})
