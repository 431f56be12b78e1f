"""Host-side logic that needs no GPU: message packing, share ordering, sharding, workload
determinism, and the world_size-2 gloo path of the multi-GPU helpers."""
import os
import sys

import numpy as np
import pytest

from threshold_crypto_amd import api, parallel, workload
from threshold_crypto_amd.engine import pack_messages


def test_pack_messages():
    flat, off = pack_messages([b"ab", b"", b"cde"])
    assert off.tolist() == [0, 2, 2, 5] and bytes(flat[:5]) == b"abcde" and off.dtype == np.uint64
    flat, off = pack_messages([])
    assert off.tolist() == [0]


def test_share_ordering_matches_btreemap():
    d = {8: "c", 5: "a", 7: "b"}
    assert api._ordered(d) == [(5, "a"), (7, "b"), (8, "c")]
    assert api._ordered([(9, "x"), (1, "y")]) == [(9, "x"), (1, "y")]
    assert api.into_fr_plus_1(0) == 1 and api.into_fr_plus_1(2 ** 64 - 1) == 2 ** 64


def test_secret_key_set_matches_horner():
    s = api.SecretKeySet([5, 7, 11])
    assert s.threshold() == 2
    assert s.secret_key_share(0).fr == 5 + 7 + 11 and s.secret_key_share(2).fr == 5 + 7 * 3 + 11 * 9


@pytest.mark.parametrize("total,world", [(10, 1), (10, 3), (65536, 8), (7, 8), (0, 2)])
def test_shard_range_partitions(total, world):
    spans = [parallel.shard_range(total, world, r) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == total
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b == c and b - a >= d - c >= 0


def test_workload_is_deterministic_and_well_formed():
    a = workload.signer_subsets(50, 10, 3)
    b = workload.signer_subsets(50, 10, 3)
    assert (a == b).all() and a.shape == (50, 4)
    assert all(len(set(r)) == 4 and list(r) == sorted(r) and max(r) < 10 for r in a.tolist())
    assert (workload.signer_subsets(10, 10, 3, start=40) == a[40:50]).all()
    assert len({tuple(r) for r in a.tolist()}) > 20
    assert workload.messages(2, start=5) == [b"tc/msg" + (5).to_bytes(8, "little"), b"tc/msg" + (6).to_bytes(8, "little")]
    assert workload.key_set(3).poly == workload.key_set(3).poly and len(workload.key_set(3).poly) == 4


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    commit = torch.arange(4 * 96, dtype=torch.int64).to(torch.uint8) if rank == 0 else torch.zeros(4 * 96, dtype=torch.uint8)
    commit = parallel.broadcast_key_set(commit, world)
    lo, hi = parallel.shard_range(1001, world, rank)
    tot = parallel.total_count(hi - lo, world)
    q.put((rank, int(commit.to(torch.int64).sum()), lo, hi, tot))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_sharding():
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = int(torch.arange(4 * 96, dtype=torch.int64).to(torch.uint8).to(torch.int64).sum())
    assert [r[1] for r in res] == [want, want]           # both ranks hold rank 0's commitment
    assert res[0][2:4] == (0, 501) and res[1][2:4] == (501, 1001)
    assert res[0][4] == res[1][4] == 1001


# ---- BASELINE config 5 flow on two gloo ranks, the host-compiled device source standing in for the GPU -------
class HostSimEngine:
    """The engine methods threshold_crypto_amd/config5.py uses, executed by tests/hostsim (the SAME per-lane job
    bodies the kernels run, compiled by g++).  Test harness only."""

    def __init__(self):
        import ctypes
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        csrc = os.path.join(os.path.dirname(here), "threshold_crypto_amd", "csrc")
        src = os.path.join(here, "hostsim", "hostsim.cpp")
        lib = os.path.join(here, "hostsim", "libtc_hostsim.so")
        newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith(".h"))
        if not os.path.exists(lib) or os.path.getmtime(lib) < max(newest, os.path.getmtime(src)):
            subprocess.run(["g++", "-O2", "-std=c++17", "-DTC_TEST_HOOKS", "-shared", "-fPIC", "-I" + csrc, src, "-o", lib], check=True)
        self.L = ctypes.CDLL(lib)
        self.ct = ctypes
        self.L.hs_hash_g2.restype = None

    def last_kernel_ms(self):
        return 0.0

    def _buf(self, n):
        return self.ct.create_string_buffer(n)

    def g1_commitment(self, fr):
        out = np.zeros((fr.shape[0], 96), np.uint8)
        st = np.zeros(fr.shape[0], np.uint8)
        for i in range(fr.shape[0]):
            b = self._buf(96)
            st[i] = self.L.hs_g1_fixed_base_mul(bytes(fr[i]), b)
            out[i] = np.frombuffer(b.raw, np.uint8)
        return out, st

    def hash_g2(self, flat, off):
        B = off.shape[0] - 1
        out = np.zeros((B, 192), np.uint8)
        for j in range(B):
            m = bytes(flat[int(off[j]): int(off[j + 1])])
            b = self._buf(192)
            self.L.hs_hash_g2(m, self.ct.c_size_t(len(m)), b)
            out[j] = np.frombuffer(b.raw, np.uint8)
        return out

    def sign_shares_g2(self, sk_table, idx, hashes):
        B, n = idx.shape
        out = np.zeros((B, n, 192), np.uint8)
        st = np.zeros((B, n), np.uint8)
        for j in range(B):
            for k in range(n):
                b = self._buf(192)
                st[j, k] = self.L.hs_g2_mul(bytes(sk_table[int(idx[j, k])]), bytes(hashes[j]), b)
                out[j, k] = np.frombuffer(b.raw, np.uint8)
        return out, st

    def combine_g2(self, t, idx, shares):
        B, n = idx.shape
        out = np.zeros((B, 192), np.uint8)
        st = np.zeros(B, np.uint8)
        for j in range(B):
            b = self._buf(192)
            ids = (self.ct.c_uint64 * n)(*[int(v) for v in idx[j]])
            st[j] = self.L.hs_combine_g2(int(t), ids, shares[j].tobytes(), b)
            out[j] = np.frombuffer(b.raw, np.uint8)
        return out, st

    def verify_g2(self, pk, sig, hashes):
        g1 = api._G1_GEN
        return np.array([self.L.hs_pairing_check(bytes(pk), bytes(hashes[j]), g1, bytes(sig[j])) for j in range(sig.shape[0])], np.uint8)


def _config5_worker(rank, world, port, q):
    import torch.distributed as dist
    from threshold_crypto_amd import config5
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = config5.run_pipeline(HostSimEngine(), 8, 12, 3, rank, world, device=None, steps=1)
    q.put((rank, res["start"], res["jobs"], res["status_errors"], res["valid_local"], res["valid_total"], res["records"],
           res["key_material"].commit.tobytes(), res["key_material"].sk_table.tobytes(), res["sig"].tobytes(), res["idx"].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_config5_two_rank_gloo_pipeline_sharding_broadcast_gather():
    """BASELINE config 5 flow (threshold_crypto_amd/config5.py, the code bench.py --config 5 runs) on two gloo
    ranks: rank 0's key material reaches rank 1 in ONE broadcast, each rank derives its slice from GLOBAL job
    indices, signs the selected shares, combines (t + 1 = 9: the large-threshold two-stage path) and verifies
    them; the valid counts are all-reduced and one record per rank is gathered.  Kernels are the host build of
    the device source.  The union of both ranks' signatures equals a single-rank run over all six jobs."""
    import torch.multiprocessing as mp
    from threshold_crypto_amd import config5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_config5_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    r0, r1 = res
    assert (r0[1], r0[2], r1[1], r1[2]) == (0, 3, 3, 3)                     # contiguous slices of the global batch
    assert r0[3] == r1[3] == 0 and r0[4] == r1[4] == 3 and r0[5] == r1[5] == 6   # no errors; 3 + 3 verified; all-reduce
    assert r0[7] == r1[7] and r0[8] == r1[8]                                 # rank 1 holds rank 0's commitment + share table
    assert r0[6] == r1[6] and [rec[:3] for rec in r0[6]] == [[0, 3, 3], [3, 3, 3]]   # gathered records, rank order
    assert r0[6][0][3] == parallel.digest64(r0[9]) and r0[6][1][3] == parallel.digest64(r1[9])
    single = config5.run_pipeline(HostSimEngine(), 8, 12, 6, 0, 1, device=None, steps=1)
    assert single["sig"].tobytes() == r0[9] + r1[9] and single["idx"].tolist() == r0[10] + r1[10]
    # and the signatures are right: each equals the master key's signature of the job's hash point
    eng = HostSimEngine()
    msk = single["secret_key_set"].poly[0].to_bytes(32, "little")
    for j in range(6):
        b = eng._buf(192)
        assert eng.L.hs_g2_mul(msk, bytes(single["hashes"][j]), b) == 0 and b.raw == single["sig"][j].tobytes()


def test_committed_bench_lines_are_self_consistent():
    """The bench lines kept under profiles/ (what DESIGN.md quotes) obey the arithmetic of the contract: value =
    units / step time, frac = achieved / peak, achieved = executed multiply-adds x units / kernel time, every fraction a
    utilisation, the kernels named in the line present in the rocprofv3 summary captured with it."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    macs = json.load(open(os.path.join(root, "profiles", "executed_macs.json")))
    for tag in ("r02_i", "r02_j"):
        line = [l for l in open(os.path.join(root, "profiles", tag + "_bench.txt")) if l.startswith("{")][-1]
        d = json.loads(line)
        B = d["config"]["batch_per_gpu"]
        assert d["metric"] == "combine_signatures/sec" and d["n_gpus"] == 1 and d["vs_baseline"] is None
        assert abs(d["value"] - B / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3
        assert abs(d["sequential"]["value"] - B / (d["sequential"]["ms_per_step"] * 1e-3)) / d["sequential"]["value"] < 2e-3
        legs = [d["roofline"]] + list(d["secondary_rooflines"].values())
        for r in legs:
            assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
            want = r["executed_macs_per_unit"] * r["units_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12
            assert abs(r["achieved"] - want) / want < 2e-3
            assert r["hbm_frac"] < 0.01                       # nowhere near the HBM roof: the bound is the integer multiplier
        assert d["roofline"]["executed_macs_per_unit"] == macs["combine_g2_t3_fast"]
        assert d["secondary_rooflines"]["pairing_check"]["executed_macs_per_unit"] == macs["verify_g2"]
        assert 0 < d["roofline"]["frac_timed_region"] <= 1 and d["roofline"]["frac_slowest_class"] <= 1
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and d["value"] / c["value"] > 100
        summary = open(os.path.join(root, "profiles", tag + "_rocprofv3_summary.csv")).read()
        for k in ("k_combine_fast<tc::Fq2>", "k_pairing_check", "k_hash_g2", "k_g2_mul_shared", "SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
            assert k in summary, (tag, k)
    c5 = json.loads([l for l in open(os.path.join(root, "profiles", "r02_config5_1gpu_bench.txt")) if l.startswith("{")][-1])
    assert c5["config"]["t"] == 67 and c5["config"]["N"] == 200 and c5["verified_all"] is True
    assert abs(c5["value"] - c5["config"]["batch_per_gpu"] / (c5["ms_per_step"] * 1e-3)) / c5["value"] < 2e-3
    assert abs(sum(c5["phase_kernel_ms"].values()) - c5["ms_per_step"]) / c5["ms_per_step"] < 0.02   # the step IS its three kernels' time
