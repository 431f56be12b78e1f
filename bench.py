#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|5] [--batch B] [--t T] [--signers N]

--config 2 (default; BASELINE "t=3, N=10, batch=65 536 threshold signatures on 1xMI355X"): a "step" is
one pass of PublicKeySet::combine_signatures (src/lib.rs:608-615) over `batch` independent
(message, share-set) jobs through the C ABI (tc_combine_g2_batch) with every input already resident in
HBM.  The K timed steps are issued without host synchronisation and alternate between two contexts (--in-flight, one
HIP stream each), so consecutive launches overlap at their ends -- each step is still a full pass over the batch; the
JSON line also carries the one-context, host-waits-every-step figure (`sequential`), and the roofline's per-launch
kernel time is measured on that leg.  The same batch is then verified (PublicKey::verify_g2, src/lib.rs:108-110; every 16th signature
replaced by its neighbour's, so the expected ok-vector is known), re-signed, verified with hashing on the
device, and run through the threshold-decryption path (BASELINE configs 3 and 4); each of those legs
carries its own roofline object and ANY failing leg fails the run.

--config 5 (BASELINE "t=67, N=200, batch=1 048 576 mixed sign+combine+verify sharded across 8 GPUs"):
one step = sign the t+1 selected shares of every job ON the device, combine them, verify the result;
131 072 jobs per GPU (weak scaling: --gpus 8 is the BASELINE batch), nothing larger than the key set
crosses PCIe or xGMI.

One process per GPU; for N > 1 the batch is per-rank (weak scaling), the key-set parameters are broadcast
from rank 0 over RCCL and the per-rank valid counts are all-reduced; there is no data-path collective
(jobs are independent).  Prints ONE JSON line (rank 0).  The oracle (oracle/) is used only for the
cpu_baseline leg and to check the GPU output of the timed batch bit-for-bit.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# frozen reference-algorithm work constants (Fq multiplications+squarings per unit), measured with
# Oracle B's counter (oracle/c/tc_oracle.c or_fq_mul_count; DESIGN.md "Work constants"); 1 Fq-mul = 300
# 32-bit multiply-adds (12x12 product + 12x12 reduction + 12 quotient digits, CIOS: SURVEY 8d)
W_FQMUL = {"combine_g2_t3": 31148, "g2_mul": 7771, "verify_g2": 41226, "hash_g2": 19604, "combine_g1_t3": 12800,
           "combine_g2_t67": 530000}
MAC_PER_FQMUL = 300
# what the kernels execute per unit, in v_mad_i64_i32 lane-instructions (one 14x14 limb product or one
# Montgomery reduction = 196): counted by running the same per-lane job bodies in the host build
# (tests/count_ops.py).  Two lanes work on a G2 job; work inside Fq2 operations is split between them, Fq
# work outside (inversions, root exponentiations) is done by both and counted twice.
EXECUTED_MACS = json.load(open(os.path.join(ROOT, "profiles", "executed_macs.json")))
# algorithmic bytes per unit (canonical uncompressed affine I/O)
ALG_BYTES = {"combine_g2": lambda t: (t + 1) * (192 + 8) + 192, "verify_g2": lambda t: 385, "g2_mul": lambda t: 192,
             "hash_g2": lambda t: 207, "combine_g1": lambda t: (t + 1) * (96 + 8) + 96 + 32}
# Roofline peak: the issue rate of the multiplier's own instruction (v_mad_i64_i32) with every SIMD full,
# measured LIVE by tools/ubench_clock --peak when that binary is present (sustained ~25 ms launches, clock
# read from s_memtime / s_memrealtime); otherwise the value recorded in profiles/r02_ubench_clock.txt.
PEAK_RECORDED = {"tmacs": 37.5, "clock_ghz": 2.1, "source": "profiles/r02_ubench_chain.txt (v_mad_i64_i32 chains, >= 2 waves/SIMD)"}
HBM_PEAK_GBPS = 8000.0
# L2<->fabric bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate PMC passes, FETCH doubled per the
# gfx950 note of MI355X_MICROARCH.md) and SQ_INSTS_VALU per launch at batch 65 536, from the profile named
PROFILE = json.load(open(os.path.join(ROOT, "profiles", "profile_constants.json")))


def measure_peak():
    exe = os.path.join(ROOT, "tools", "ubench_chain")
    try:
        out = subprocess.run([exe, "--peak"], capture_output=True, text=True, timeout=120).stdout
        for line in out.splitlines():
            d = json.loads(line)
            if d.get("op") == "v_mad_i64_i32_chained" and d.get("waves_per_simd") == 4:
                clk = None
                try:   # the clock under the same kind of load, from the s_memtime / s_memrealtime microbenchmark
                    o2 = subprocess.run([os.path.join(ROOT, "tools", "ubench_clock"), "--peak"], capture_output=True, text=True, timeout=120).stdout
                    clk = [json.loads(l) for l in o2.splitlines() if "effective_clock_GHz_median" in l][0]["effective_clock_GHz_median"]
                except Exception:
                    pass
                return {"tmacs": d["T_mad_per_s"], "clock_ghz": clk,
                        "source": "measured live: tools/ubench_chain --peak (v_mad_i64_i32, stable multiplicands + two chained "
                                  "accumulators as in the field multiplier, 4 waves/SIMD, %.1f ms launch)" % d["ms"]}
    except Exception:
        pass
    return dict(PEAK_RECORDED)


def roofline(kernel, unit_key, ref_key, alg_key, t, units, kernel_ms, peak, traffic_key=None, extra=None):
    """frac = the kernel's OWN multiply-add count / time / peak (utilisation of the integer multiplier);
    algorithmic_speedup = reference-algorithm work / executed work (what the smarter algorithm buys)."""
    sec = kernel_ms * 1e-3
    executed = EXECUTED_MACS[unit_key] * units
    ref = W_FQMUL[ref_key] * MAC_PER_FQMUL * units if ref_key else None
    ach = executed / sec / 1e12
    alg_bytes = ALG_BYTES[alg_key](t) * units
    prof = PROFILE.get(traffic_key or "", {})
    r = {"bound": "valu_int32_mac", "kernel": kernel, "kernel_ms": round(kernel_ms, 3), "units_per_launch": units,
         "achieved": round(ach, 3), "peak": peak["tmacs"], "unit": "TMAC/s", "frac": round(ach / peak["tmacs"], 4),
         "peak_clock_GHz": peak["clock_ghz"], "peak_source": peak["source"],
         "executed_macs_per_unit": EXECUTED_MACS[unit_key],
         "achieved_is": "v_mad the kernel executes per unit (tests/count_ops.py) x units / HIP-event kernel time",
         "algorithmic_bytes_per_launch": alg_bytes,
         "hbm_achieved_GBps": round(alg_bytes / sec / 1e9, 3), "hbm_peak_GBps": HBM_PEAK_GBPS,
         "hbm_frac": round(alg_bytes / sec / 1e9 / HBM_PEAK_GBPS, 6),
         "traffic": prof.get("traffic_bytes") if units == prof.get("units") else None,
         "traffic_is": ("L2<->fabric bytes per launch (2 x FETCH_SIZE + WRITE_SIZE), %s" % prof.get("source")) if prof else None}
    if ref:
        r["reference_work_TMACs"] = round(ref / sec / 1e12, 3)
        r["algorithmic_speedup"] = round(ref / executed, 3)
    if prof.get("sq_insts_valu") and units == prof.get("units"):
        lane_instr = prof["sq_insts_valu"] * 64
        r["executed_cross_check"] = {"sq_insts_valu_wave_instr_per_launch": prof["sq_insts_valu"],
                                     "executed_v_mad_lane_instr_per_launch": executed,
                                     "implied_v_mad_share_of_valu": round(executed / lane_instr, 3),
                                     "static_v_mad_share_of_the_multiplier_bodies": PROFILE.get("static_mad_share")}
    if extra:
        r.update(extra)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 20 (config 2), 6 (config 5)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=[2, 5])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--t", type=int, default=None)
    ap.add_argument("--signers", type=int, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="contexts (one HIP stream each) the timed steps of config 2 alternate between; 1 = one context, "
                         "the host waits for every step")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary legs (configs 3 and 4, PCIe-inclusive rate)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20 if args.config == 2 else 6

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    from threshold_crypto_amd.engine import Engine
    eng = Engine(local_rank)
    peak = measure_peak() if rank == 0 else dict(PEAK_RECORDED)
    if args.config == 5:
        from threshold_crypto_amd import config5
        result = config5.run_bench(args, eng, dev, rank, world, peak, roofline, cpu_baseline=cpu_baseline_config5)
    else:
        result = run_config2(args, eng, dev, rank, world, peak)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return result


def run_config2(args, eng, dev, rank, world, peak):
    import numpy as np
    import torch
    from threshold_crypto_amd.workload import ThresholdSigWorkload, ThresholdEncWorkload
    from threshold_crypto_amd.parallel import broadcast_key_set, shard_range, total_count
    if world > 1:
        import torch.distributed as dist

    t = 3 if args.t is None else args.t
    N = 10 if args.signers is None else args.signers
    B = 65536 if args.batch is None else args.batch
    start, _ = shard_range(B * world, world, rank)      # weak scaling: B jobs per rank
    wl = ThresholdSigWorkload(eng, t, N, B, start=start)

    # key-set parameters (commitment) travel rank 0 -> all ranks over RCCL/xGMI
    commit = torch.from_numpy(np.frombuffer(b"".join(wl.sks.public_keys(eng).commit), dtype=np.uint8).copy()).to(dev)
    commit = broadcast_key_set(commit, world)
    master_pk = commit[:96].contiguous()

    d_idx = torch.from_numpy(wl.idx.view(np.int64)).to(dev)
    d_shares = torch.from_numpy(wl.shares).to(dev)
    d_hashes = torch.from_numpy(wl.hashes).to(dev)
    d_sk = torch.from_numpy(np.stack([np.frombuffer(s._bytes(), dtype=np.uint8) for s in wl.shares_sk[: t + 1]])).to(dev)

    def sync():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- headline: combine_signatures ----------------------------------------------------
    # The timed region issues its K steps WITHOUT host synchronisation, alternating between `--in-flight` contexts (one
    # HIP stream each, the same device-resident operands, separate outputs): every step is a full pass over the batch,
    # and the tail of one launch -- the 65 536-job batch is a single resident round that ends with a third of the SIMDs
    # idle -- overlaps the head of the next.  This is how a caller with a stream of batches drives the
    # library; `sequential` below is the same work on one context with the host waiting for each step,
    # and the per-launch kernel time of the roofline comes from that leg (un-overlapped launches).
    from threshold_crypto_amd.engine import Engine
    in_flight = max(1, int(args.in_flight))
    engines = [eng] + [Engine(dev.index if dev.index is not None else 0) for _ in range(in_flight - 1)]
    for e in engines:
        e.set_timing(False)
        for _ in range(max(1, args.warmup)):
            sig, st = e.combine_g2(t, d_idx, d_shares)
    for e in engines:
        e.sync()
    sync()
    t0 = time.perf_counter()
    outs = []
    for i in range(args.steps):
        outs.append(engines[i % in_flight].combine_g2(t, d_idx, d_shares))
        if len(outs) > 2 * in_flight:
            outs.pop(0)
    for e in engines:
        e.sync()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = B * world / (dt / args.steps)
    for o_sig, o_st in outs:
        assert int(o_st.to(torch.int32).sum().item()) == 0, "combine reported per-job errors"
    # the same steps one at a time (host waits for each): per-launch kernel times for the roofline
    eng.set_timing(True)
    seq_steps = args.steps
    kernel_ms = []
    sync()
    t0 = time.perf_counter()
    for _ in range(seq_steps):
        sig, st = eng.combine_g2(t, d_idx, d_shares)
        kernel_ms.append(eng.last_kernel_ms())
    sync()
    seq_dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([seq_dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        seq_dt = float(tt.item())
    assert int(st.to(torch.int32).sum().item()) == 0, "combine reported per-job errors"
    for o_sig, _ in outs:
        assert bool((o_sig == sig).all().item()), "steps in flight and the sequential step disagree"
    del outs
    seq_ms_per_step = seq_dt / seq_steps * 1e3
    if seq_ms_per_step < ms_per_step:
        # (never observed with two contexts; with four or more, concurrent queues can start trading scratch reservations
        # -- INTEGRATION.md -- and then the one-context region is the honest headline)
        in_flight, ms_per_step, value = 1, seq_ms_per_step, B * world / (seq_dt / seq_steps)

    # ---- config 3: verify the combined signatures; every 16th one replaced by its neighbour's ------------
    bad = sig.clone()
    expect_ok = torch.ones(B, dtype=torch.uint8, device=dev)
    if B >= 2:
        planted = torch.arange(0, B, 16, device=dev)
        bad[planted] = sig[(planted + 1) % B]
        expect_ok[planted] = 0
    ok = eng.verify_g2(master_pk, bad, d_hashes)
    sync()
    v0 = time.perf_counter()
    ok = eng.verify_g2(master_pk, bad, d_hashes)
    verify_kernel_ms = eng.last_kernel_ms()
    sync()
    verify_dt = time.perf_counter() - v0
    assert bool((ok == expect_ok).all().item()), "verify_g2 ok-vector differs from the planted corruption pattern"
    # the same check with calls in flight (as the headline): 8 steps alternating between the contexts
    verify_if_dt = verify_dt
    if len(engines) > 1:
        eng.set_timing(False)
        for e in engines:
            e.verify_g2(master_pk, bad, d_hashes)
        for e in engines:
            e.sync()
        sync()
        v0 = time.perf_counter()
        oks = [engines[i % len(engines)].verify_g2(master_pk, bad, d_hashes) for i in range(8)]
        for e in engines:
            e.sync()
        sync()
        verify_if_dt = (time.perf_counter() - v0) / 8
        for o_ok in oks:
            assert bool((o_ok == expect_ok).all().item()), "verify_g2 in flight: ok-vector differs from the planted corruption pattern"
        del oks
        eng.set_timing(True)
    for e in engines[1:]:
        e.close()
    n_valid = total_count(int(ok.to(torch.int64).sum().item()), world, dev)   # per-rank valid counts, summed over RCCL
    assert n_valid == int(expect_ok.sum().item()) * world
    s0 = time.perf_counter()
    _sh, _st = eng.g2_mul(d_sk, d_hashes)
    sign_kernel_ms = eng.last_kernel_ms()
    sync()
    sign_dt = time.perf_counter() - s0
    # size-independent property at full size: the master key's own signature equals the combination
    msig, _ = eng.g2_mul(torch.from_numpy(wl.master_sk_fr[None].copy()).to(dev), d_hashes)
    sync()
    assert bool((msig[:, 0] == sig).all().item()), "combine != master-key signature"

    extras, legs = {}, {}
    if not args.no_extras:
        # ---- verify incl. hashing on the device, hash_g2 alone --------------------------------------------
        d_msgs = torch.from_numpy(wl.msg_flat).to(dev)
        d_off = torch.from_numpy(wl.msg_off.view(np.int64)).to(dev)
        ok2 = eng.verify_sig(master_pk, bad, d_msgs, d_off)
        sync()
        e0 = time.perf_counter()
        ok2 = eng.verify_sig(master_pk, bad, d_msgs, d_off)
        sync()
        extras["verifies_with_hash_per_s"] = round(B * world / (time.perf_counter() - e0), 1)
        assert bool((ok2 == expect_ok).all().item()), "verify (hash on device) ok-vector differs from the pattern"
        e0 = time.perf_counter()
        hh = eng.hash_g2(d_msgs, d_off)
        hash_kernel_ms = eng.last_kernel_ms()
        sync()
        extras["hash_g2_per_s"] = round(B * world / (time.perf_counter() - e0), 1)
        assert bool((hh == d_hashes).all().item())
        legs["hash_g2"] = roofline("k_hash_g2", "hash_g2", "hash_g2", "hash_g2", t, B, hash_kernel_ms, peak)
        # ---- config 4: threshold decryption = Ciphertext::verify + G1 combine + keystream -------------------
        we = ThresholdEncWorkload(eng, t, N, B, start=start)
        du, dv, dw = torch.from_numpy(we.u).to(dev), torch.from_numpy(we.v).to(dev), torch.from_numpy(we.w).to(dev)
        doff = torch.from_numpy(we.off.view(np.int64)).to(dev)
        didx, dsh = torch.from_numpy(we.idx.view(np.int64)).to(dev), torch.from_numpy(we.shares).to(dev)
        okc = eng.ciphertext_verify(du, dv, doff, dw)
        out, dst = eng.decrypt(t, didx, dsh, dv, doff)
        sync()
        e0 = time.perf_counter()
        okc = eng.ciphertext_verify(du, dv, doff, dw)
        extras["ciphertext_verify_kernel_ms"] = round(eng.last_kernel_ms(), 3)
        sync()
        e1 = time.perf_counter()
        out, dst = eng.decrypt(t, didx, dsh, dv, doff)
        dec_kernel_ms = eng.last_kernel_ms()
        sync()
        e2 = time.perf_counter()
        extras["ciphertext_verifies_per_s"] = round(B * world / (e1 - e0), 1)
        extras["threshold_decrypts_per_s"] = round(B * world / (e2 - e1), 1)
        extras["threshold_decryptions_incl_ciphertext_verify_per_s"] = round(B * world / (e2 - e0), 1)
        assert int(okc.to(torch.int32).sum().item()) == B and int(dst.to(torch.int32).sum().item()) == 0
        assert bool((out.cpu() == torch.from_numpy(we.plain_flat)[: out.numel()]).all().item()), "threshold decryption returned wrong plaintext"
        legs["threshold_decrypt"] = roofline("k_combine_fast<Fq> + k_xor_with_hash", "combine_g1_t3_fast", "combine_g1_t3",
                                             "combine_g1", t, B, dec_kernel_ms, peak)
        # ---- the same combine with HOST buffers at the C ABI (pageable numpy memory): PCIe-inclusive ---------
        eng.combine_g2(t, wl.idx, wl.shares)
        best = 1e9
        for _ in range(2):
            p0 = time.perf_counter()
            hsig, hst = eng.combine_g2(t, wl.idx, wl.shares)
            best = min(best, time.perf_counter() - p0)
        assert not hst.any() and bool((torch.from_numpy(hsig).to(dev) == sig).all().item())
        extras["pcie_inclusive_combine_per_s"] = round(B / best, 1)
        extras["pcie_inclusive_is"] = "tc_combine_g2_batch with host pointers: H2D of %d B + kernels + D2H of %d B, wall clock, one GPU" % (
            wl.idx.nbytes + wl.shares.nbytes, hsig.nbytes + hst.nbytes)

    if rank != 0:
        return None
    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
    fast = 1 <= t <= 3
    unit_key = "combine_g2_t3_fast" if t == 3 else None
    head = None
    if unit_key:
        head = roofline("k_lagrange + 3 grouping kernels + k_combine_fast<Fq2> + k_combine_general<Fq2>", unit_key, "combine_g2_t3",
                        "combine_g2", t, B, avg_kernel_ms, peak, traffic_key="combine_g2_t3",
                        extra={"frac_slowest_class": round(EXECUTED_MACS["combine_g2_t3_fast_general_denominator"] * B
                                                           / (avg_kernel_ms * 1e-3) / 1e12 / peak["tmacs"], 4),
                               "frac_timed_region": round(EXECUTED_MACS[unit_key] * B / (ms_per_step * 1e-3) / 1e12 / peak["tmacs"], 4),
                               "frac_timed_region_is": "executed multiply-adds of the K timed steps / their wall time / peak: the "
                                                       "machine's utilisation with steps in flight (frac is per launch)",
                               "frac_is": "average job over the 4-of-10 subsets; the 65 536-job batch is ONE resident round "
                                          "of 2048 waves and lasts as long as its slowest denominator class, whose waves run "
                                          "at frac_slowest_class (DESIGN.md 5.2)"})
    legs["pairing_check"] = roofline("k_pairing_check", "verify_g2", "verify_g2", "verify_g2", t, B, verify_kernel_ms, peak,
                                     traffic_key="pairing_check")
    legs["g2_sign"] = roofline("k_g2_mul_shared", "g2_mul_4_scalars_per_point", "g2_mul", "g2_mul", t, (t + 1) * B, sign_kernel_ms, peak)
    # the CPU leg runs on rank 0 of a one-GPU run only (N ranks would time N oracles against each other on one host)
    cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline(wl, sig.cpu().numpy(), t, args.cpu_seconds)
    return {
        "metric": "combine_signatures/sec", "value": round(value, 1), "unit": "combine_signatures/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i32 limbs (Fq = 14 x 28-bit signed, Montgomery R=2^392; 64-bit column accumulators; one Fq2 coefficient per lane of a lane pair)",
        "data": "synthetic",
        "config": {"workload": "t=%d,N=%d,batch=%d threshold signatures (combine_signatures, G2), per GPU" % (t, N, B),
                   "t": t, "N": N, "batch_per_gpu": B, "parallelism": "jobs sharded, dp%d" % world,
                   "fast_path": fast, "steps_in_flight": in_flight,
                   "steps_in_flight_is": "the K timed steps alternate between this many contexts (one HIP stream each) and are "
                                         "not synchronised with the host inside the timed region; each is a full pass over "
                                         "the batch (bench.py --in-flight 1: one context, host waits for every step)"},
        "sequential": {"value": round(B * world / (seq_dt / seq_steps), 1), "ms_per_step": round(seq_dt / seq_steps * 1e3, 3),
                       "steps": seq_steps, "is": "the same step on ONE context, the host waiting for each step; the roofline's "
                                                 "per-launch kernel time is measured here (launches do not overlap)"},
        "pairing_verifies_per_s": round(B * world / min(verify_dt, verify_if_dt), 1),
        "pairing_verifies_per_s_is": "verify_g2 over the batch, calls in flight over the same contexts as the headline (8 steps); "
                                     "one call at a time: pairing_verifies_sequential_per_s",
        "pairing_verifies_sequential_per_s": round(B * world / verify_dt, 1),
        "pairing_verify_kernel_ms": round(verify_kernel_ms, 3),
        "pairing_verify_valid_count_all_ranks": n_valid,
        "share_signs_per_s": round((t + 1) * B * world / sign_dt, 1),
        "share_sign_kernel_ms": round(sign_kernel_ms, 3),
        "verified_all": True,
        "extras": extras,
        "roofline": head,
        "secondary_rooflines": legs,
        "cpu_baseline": cpu,
    }


def cpu_baseline(wl, gpu_sigs, t, seconds):
    """Oracle B (plain-C port of the reference algorithm, pthreads over the host cores) on a bounded sample
    of the SAME jobs; also the bit-exact check of the GPU output on that sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    c_oracle.load()
    threads = c_oracle.host_threads()
    n1 = 8
    t0 = time.perf_counter()
    out1, rc1 = c_oracle.combine_g2_batch(t, wl.idx[:n1], wl.shares[:n1], 1)
    per = max((time.perf_counter() - t0) / n1, 1e-5)
    n = int(max(threads, min(wl.B, seconds / per * min(threads, 24))))   # ~seconds of wall clock if ~24 cores are real
    t0 = time.perf_counter()
    out, rc = c_oracle.combine_g2_batch(t, wl.idx[:n], wl.shares[:n], threads)
    dt = time.perf_counter() - t0
    mism = int(rc.any()) + int((out != gpu_sigs[:n]).any(axis=1).sum())
    if mism:
        raise AssertionError("GPU combine differs from the CPU oracle on %d of %d sampled jobs" % (mism, n))
    return {"value": round(n / dt, 2), "unit": "combine_signatures/s", "cores": threads, "kind": "port",
            "cores_is": "threads started = len(sched_getaffinity) capped by the cgroup CPU quota (os.cpu_count() = %d)" % (os.cpu_count() or 0),
            "thread_scaling": round((n / dt) * per, 2),
            "thread_scaling_is": "all-thread rate / single-thread rate: the number of cores the lease really delivers",
            "sample": "first %d jobs of the timed batch on %d pthreads (oracle/c/tc_oracle.c, gcc -O3 x86-64-v3); "
                      "every sampled job compared bit-exact with the GPU output" % (n, threads),
            "single_thread_per_s": round(1.0 / per, 2)}


def cpu_baseline_config5(res, t):
    """Oracle B on a handful of the rank's jobs (a t=67 combination takes ~0.1 s per core): bit-exact check + rate."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    c_oracle.load()
    threads = c_oracle.host_threads()
    n = max(threads, 16)
    idx = res["idx"][:n]
    km = res["key_material"]
    shares = np.empty((n, t + 1, 192), dtype=np.uint8)
    hashes = res["hashes"].cpu().numpy() if hasattr(res["hashes"], "cpu") else np.asarray(res["hashes"])
    t0 = time.perf_counter()
    for j in range(n):
        for k in range(t + 1):
            rc, out = c_oracle.g2_mul(bytes(km.sk_table[int(idx[j, k])]), bytes(hashes[j]))
            shares[j, k] = np.frombuffer(out, dtype=np.uint8)
    sign_dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    out, rc = c_oracle.combine_g2_batch(t, idx, shares, threads)
    dt = time.perf_counter() - t0
    assert not rc.any() and (out == res["sig"][:n]).all(), "GPU config-5 signatures differ from the CPU oracle"
    return {"value": round(n / dt, 2), "unit": "combine_signatures/s", "cores": threads, "kind": "port",
            "sample": "first %d jobs of rank 0 (t=%d): shares signed by Oracle B single-threaded (%.1f s), combined on %d pthreads; "
                      "every sampled signature compared bit-exact with the GPU output" % (n, t, sign_dt, threads)}


if __name__ == "__main__":
    main()
