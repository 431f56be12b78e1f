"""BASELINE config 5: "t=67, N=200, batch=1 048 576 mixed sign+combine+verify sharded across 8 GPUs".

One step of one rank, everything resident in HBM:
    shares[j, k] = sk[idx[j, k]] * H_j      tc_sign_shares_g2_batch   (SecretKeyShare::sign_g2, src/lib.rs:442-444)
    sig[j]       = interpolate(shares[j])    tc_combine_g2_batch       (combine_signatures, src/lib.rs:608-615)
    ok[j]        = pk.verify_g2(sig[j], H_j) tc_verify_g2_batch        (PublicKey::verify_g2, src/lib.rs:108-110)
The shares are generated ON the device from the key set (the raw share input of the full batch would be
1 048 576 x 68 x 192 B = 13.7 GB): per rank only the key set (6.5 KB commitment + 6.4 KB secret share
table) arrives over RCCL and the per-job subsets / messages are derived from the GLOBAL job index, so a rank's
slice is the same whatever the world size.  Valid counts are all-reduced, one record per rank is gathered.
The module is engine-agnostic: tests/test_host_logic.py drives the same code on two gloo ranks with the
host-compiled device source standing in for the GPU.
"""
import time

import numpy as np

from . import parallel
from .workload import SEED, key_set, messages

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def signer_subsets_np(B, N, t, seed=SEED, start=0):
    """Vectorised twin of workload.signer_subsets (same splitmix64 / partial Fisher-Yates, same output)."""
    with np.errstate(over="ignore"):
        state = (np.uint64(seed) ^ (np.arange(start, start + B, dtype=np.uint64)))
        perm = np.tile(np.arange(N, dtype=np.int64), (B, 1))
        rows = np.arange(B)
        for i in range(N - 1, N - 2 - t, -1):
            state = state + np.uint64(0x9E3779B97F4A7C15)
            z = state.copy()
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            r = z ^ (z >> np.uint64(31))
            k = (r % np.uint64(i + 1)).astype(np.int64)
            a = perm[rows, i].copy()
            perm[rows, i] = perm[rows, k]
            perm[rows, k] = a
    return np.sort(perm[:, N - 1 - t:], axis=1).astype(np.uint64)


class KeyMaterial:
    """What rank 0 creates and every rank receives: commitment ((t+1) x 96) and secret share table (N x 32)."""

    def __init__(self, commit, sk_table):
        self.commit, self.sk_table = commit, sk_table

    @staticmethod
    def create(engine, t, N):
        sks = key_set(t)
        fr = np.stack([np.frombuffer(c.to_bytes(32, "little"), dtype=np.uint8) for c in sks.poly])
        commit, st = engine.g1_commitment(fr)            # Poly::commitment: fixed-base kernel
        assert not np.asarray(st).any()
        sk_table = np.stack([np.frombuffer(sks.secret_key_share(i)._bytes(), dtype=np.uint8) for i in range(N)])
        return KeyMaterial(np.ascontiguousarray(commit), sk_table), sks


def distribute_key_material(km, t, N, rank, world, to_device, from_device):
    """rank 0 -> all ranks: one broadcast of commitment || share table (RCCL on GPUs, gloo in the CPU test)."""
    import torch
    n_bytes = (t + 1) * 96 + N * 32
    if rank == 0:
        blob = np.concatenate([km.commit.reshape(-1), km.sk_table.reshape(-1)])
    else:
        blob = np.zeros(n_bytes, dtype=np.uint8)
    tens = to_device(torch.from_numpy(blob))
    parallel.broadcast_key_set(tens, world)
    got = from_device(tens)
    return KeyMaterial(got[: (t + 1) * 96].reshape(t + 1, 96).copy(), got[(t + 1) * 96:].reshape(N, 32).copy())


def run_step(engine, t, sk_table, idx, hashes, master_pk):
    """sign the selected shares -> combine -> verify; returns (sig, status, ok, (ms_sign, ms_combine, ms_verify))"""
    ms = []
    shares, st0 = engine.sign_shares_g2(sk_table, idx, hashes)
    ms.append(engine.last_kernel_ms())
    sig, st = engine.combine_g2(t, idx, shares)
    ms.append(engine.last_kernel_ms())
    ok = engine.verify_g2(master_pk, sig, hashes)
    ms.append(engine.last_kernel_ms())
    return sig, st0, st, ok, tuple(ms)


def run_pipeline(engine, t, N, B, rank, world, device=None, steps=1, warmup=0, sync=None, emulated=None):
    """The whole per-rank flow (shared by bench.py --config 5 and the gloo CPU test).  Returns a dict; rank 0's
    holds the gathered per-rank records.

    emulated = (KeyMaterial, SecretKeySet): this process plays rank `rank` of a `world`-rank job BY ITSELF (run_emulated_world
    below: every slice of the BASELINE batch through one GPU, one after the other) -- the slice is cut from the global job
    range exactly as on a real rank, the key material is the one rank 0 made, and no collective runs."""
    import torch
    to_dev = (lambda x: x.to(device)) if device is not None else (lambda x: x)
    to_np = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
    sync = sync or (lambda: None)
    start, stop = parallel.shard_range(B * world, world, rank)       # weak scaling: B jobs per rank
    km = sks = None
    if emulated is not None:
        km, sks = emulated
        world = 1                                                     # collectives below: this process alone
    else:
        if rank == 0:
            km, sks = KeyMaterial.create(engine, t, N)
        km = distribute_key_material(km, t, N, rank, world, to_dev, to_np)
    master_pk = to_dev(torch.from_numpy(km.commit[0].copy())) if device is not None else km.commit[0].copy()
    sk_table = to_dev(torch.from_numpy(km.sk_table)) if device is not None else km.sk_table
    from .engine import pack_messages
    flat, off = pack_messages(messages(B, start))
    idx_np = signer_subsets_np(B, N, t, SEED, start)
    if device is not None:
        d_flat, d_off = to_dev(torch.from_numpy(flat)), to_dev(torch.from_numpy(off.view(np.int64)))
        idx = to_dev(torch.from_numpy(idx_np.view(np.int64)))
    else:
        d_flat, d_off, idx = flat, off, idx_np
    hashes = engine.hash_g2(d_flat, d_off)
    for _ in range(warmup):
        run_step(engine, t, sk_table, idx, hashes, master_pk)
    sync()
    t0 = time.perf_counter()
    phase_ms = []
    for _ in range(steps):
        sig, st0, st, ok, ms = run_step(engine, t, sk_table, idx, hashes, master_pk)
        phase_ms.append(ms)
    sync()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, world, device)
    n_bad = int(to_np(st0).astype(np.int64).sum()) + int(to_np(st).astype(np.int64).sum())
    n_valid_local = int(to_np(ok).astype(np.int64).sum())
    n_valid = parallel.total_count(n_valid_local, world, device)
    sig_np = to_np(sig)
    records = parallel.gather_records([start, stop - start, n_valid_local, parallel.digest64(sig_np.tobytes())], world, device)
    return {"rank": rank, "start": start, "jobs": stop - start, "seconds": dt, "steps": steps, "phase_ms": phase_ms,
            "status_errors": n_bad, "valid_local": n_valid_local, "valid_total": n_valid, "records": records,
            "sig": sig_np, "hashes": hashes, "idx": idx_np, "key_material": km, "secret_key_set": sks, "master_pk": master_pk}


def run_emulated_world(engine, t, N, B, world, device=None, steps=1, warmup=0, sync=None, keep=(), on_slice=None):
    """All `world` slices of a `world`-rank config-5 job through ONE engine, one after the other: BASELINE config 5 at its
    stated 1 048 576 jobs (world = 8, B = 131 072) on a single GPU.  Rank r's slice is global jobs [r B, (r + 1) B), with
    the subsets and messages a real rank r derives; the key material is created once (rank 0's) and handed on as the
    broadcast would.  on_slice(res) sees every slice's full result (signatures, hash points, subsets) before the large
    arrays are dropped (unless named in `keep`); returns the per-slice records and the totals."""
    emu = KeyMaterial.create(engine, t, N)
    slices, seconds, valid, errors = [], 0.0, 0, 0
    for r in range(world):
        res = run_pipeline(engine, t, N, B, r, world, device=device, steps=steps, warmup=warmup, sync=sync, emulated=emu)
        assert (res["start"], res["jobs"]) == (r * B, B)
        if on_slice is not None:
            on_slice(res)
        seconds += res["seconds"]
        valid += res["valid_local"]
        errors += res["status_errors"]
        rec = res["records"][0]
        slim = {k: v for k, v in res.items() if k in ("rank", "start", "jobs", "seconds", "phase_ms", "status_errors", "valid_local") or k in keep}
        slim["record"] = rec
        slices.append(slim)
    return {"slices": slices, "seconds": seconds, "steps": steps, "jobs": B * world, "valid_total": valid, "status_errors": errors,
            "records": [s["record"] for s in slices], "key_material": emu[0], "secret_key_set": emu[1]}


def run_bench(args, eng, dev, rank, world, peak, roofline, cpu_baseline=None):
    """bench.py --config 5: 131 072 jobs per GPU by default (x 8 GPUs = the BASELINE batch).  cpu_baseline: the
    caller's oracle leg (bench.py owns every use of oracle/; this package never imports it)."""
    import torch
    t = 67 if args.t is None else args.t
    N = 200 if args.signers is None else args.signers
    B = 131072 if args.batch is None else args.batch

    harness = getattr(eng, "is_test_harness", False)   # bench.py --test-engine: CPU ranks on gloo (tests/hostsim_engine.py)
    cuda = dev.type == "cuda"

    def sync():
        eng.sync()
        if cuda:
            torch.cuda.synchronize()
        if parallel.joined(world):
            import torch.distributed as dist
            dist.barrier()
            if cuda:
                torch.cuda.synchronize()

    eng.set_timing(True)
    if hasattr(eng, "set_input_checks"):
        eng.set_input_checks(False)   # the shares and hash points are made on the device by the library itself: known members
    emu_world = getattr(args, "emulate_world", 0) or 0

    def check_master_signature(res):
        # size-independent property on every job of the slice: the combination equals the master key's own signature
        master_sk = res["secret_key_set"].poly[0]
        msk = torch.from_numpy(np.frombuffer(master_sk.to_bytes(32, "little"), dtype=np.uint8)[None].copy())
        msig, _ = eng.g2_mul(msk.to(dev) if cuda else msk.numpy(), res["hashes"])
        eng.sync()      # device-I/O calls return before their kernels have run: wait before torch copies the result
        msig = msig.cpu() if hasattr(msig, "cpu") else torch.from_numpy(np.asarray(msig))
        assert bool((msig[:, 0] == torch.from_numpy(res["sig"])).all().item()), "combine != master-key signature"

    if emu_world:
        # BASELINE config 5 at its stated batch on ONE GPU: the `emu_world` rank slices one after the other
        assert world == 1, "--emulate-world runs every slice on one rank"

        first = {}

        def on_slice(r):
            if r["rank"] == 0:      # what the cpu_baseline leg samples (bench.py)
                first.update(idx=r["idx"][:256].copy(), hashes=r["hashes"][:256], sig=r["sig"][:256].copy())
            assert r["status_errors"] == 0 and r["valid_local"] == B, "config 5 slice %d: %d status errors, %d of %d verified" % (
                r["rank"], r["status_errors"], r["valid_local"], B)
            check_master_signature(r)
        emu = run_emulated_world(eng, t, N, B, emu_world, device=dev if cuda else None, steps=args.steps, warmup=args.warmup, sync=sync,
                                 on_slice=on_slice)
        res = {"phase_ms": [m for s in emu["slices"] for m in s["phase_ms"]], "seconds": emu["seconds"] / emu_world,
               "valid_total": emu["valid_total"], "records": emu["records"], "key_material": emu["key_material"],
               "secret_key_set": emu["secret_key_set"], **first}
    else:
        res = run_pipeline(eng, t, N, B, rank, world, device=dev if cuda else None, steps=args.steps, warmup=args.warmup, sync=sync)
        assert res["status_errors"] == 0 and res["valid_local"] == B, "config 5: %d status errors, %d of %d verified" % (
            res["status_errors"], res["valid_local"], B)
        if rank == 0:
            check_master_signature(res)
    if rank != 0:
        return None
    ms = np.array(res["phase_ms"], dtype=np.float64).mean(axis=0)
    if harness:
        ms = np.maximum(ms, 1e-9)   # nothing is timed per kernel in the test harness
    step_s = res["seconds"] / args.steps            # one slice's step (emulated world: the mean over the slices)
    slices_per_step = emu_world or world            # slices one "whole job" step holds
    whole_step_s = step_s * (emu_world or 1)        # emulated: the slices run one after the other on this GPU
    legs = {
        "combine": roofline("k_lagrange_all + k_msm_tables + k_msm_ladder", "combine_g2_t67_msm", "combine_g2_t67", "combine_g2", t, B,
                            float(ms[1]), peak, traffic_key="config5_combine"),
        # (csrc/tc_launch.h: the per-message comb from 24 signers and 8192 messages on)
        "share_sign": roofline("k_comb_tables + k_comb_sign" if (t + 1 >= 24 and B >= 8192) else "k_g2_mul_gather",
                               "g2_sign_comb_68_signers" if (t + 1 >= 24 and B >= 8192) else "g2_mul_gather_68_signers", "g2_mul", "g2_mul", t,
                               (t + 1) * B, float(ms[0]), peak, traffic_key="config5_sign"),
        "pairing_check": roofline("k_miller_lines + k_miller_accumulate + k_final_exp", "verify_g2_prepared", "verify_g2", "verify_g2", t, B, float(ms[2]), peak,
                                  traffic_key="config5_verify"),
    }
    cpu = cpu_baseline(res, t) if (cpu_baseline and not args.no_cpu_baseline and world == 1 and not harness) else None
    return {
        "metric": "threshold signatures (sign t+1 shares + combine + verify)/sec", "value": round(B * slices_per_step / whole_step_s, 1),
        "unit": "threshold_signatures/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(whole_step_s * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i32 limbs (Fq = 14 x 28-bit signed, Montgomery R=2^392; 64-bit column accumulators)", "data": "synthetic",
        "config": {"workload": ("t=%d,N=%d,batch=%d = %d rank slices of %d jobs run one after the other on ONE GPU: shares signed on device, "
                                "combined, verified" % (t, N, B * emu_world, emu_world, B)) if emu_world else
                               "t=%d,N=%d,batch=%d per GPU (x%d GPUs): shares signed on device, combined, verified" % (t, N, B, world),
                   "t": t, "N": N, "batch_per_gpu": B * (emu_world or 1), "parallelism": "jobs sharded, dp%d" % world,
                   "emulated_world": emu_world or None},
        "combine_signatures_per_s": round(B * world / (ms[1] * 1e-3), 1),
        "share_signs_per_s": round((t + 1) * B * world / (ms[0] * 1e-3), 1),
        "pairing_verifies_per_s": round(B * world / (ms[2] * 1e-3), 1),
        "phase_kernel_ms": {"sign": round(float(ms[0]), 3), "combine": round(float(ms[1]), 3), "verify": round(float(ms[2]), 3)},
        "valid_total_all_ranks": res["valid_total"], "rank_records_start_jobs_valid_digest": res["records"],
        "verified_all": True, "roofline": legs["combine"], "secondary_rooflines": legs, "cpu_baseline": cpu,
    }
