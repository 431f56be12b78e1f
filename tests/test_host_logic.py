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
from hostsim_engine import HostSimEngine  # noqa: E402  (tests/hostsim_engine.py: test harness)


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


R05_TAG = "r05_c"   # round 5's capture (then tools/capture_r05.sh)
R06_TAG = "r06_c"   # the capture (tools/capture_legs.sh) committed with the SHIPPED library: profiles/README.md, profile_constants.json


def _check_leg_lines(root, macs, tag, d, c5, constants):
    """rounds 5 and 6: every roofline object of the line carries `traffic` from a capture of ITS OWN leg
    (tools/profile_legs.py: the leg's timed launches only) with the ratio to the algorithmic bytes and the VALU cross-check,
    `frac_useful` beside `frac`; configs 3 and 4 and the wire leg carry a CPU baseline; the wire leg runs two decodes per lane pair.
    constants: the line belongs to the capture profile_constants.json was generated from (the newest tag only)."""
    import json
    useful = json.load(open(os.path.join(root, "profiles", "useful_macs.json")))
    prof = json.load(open(os.path.join(root, "profiles", "profile_constants.json")))
    B, t = d["config"]["batch_per_gpu"], d["config"]["t"]
    assert d["metric"] == "combine_signatures/sec" and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["config"]["overlapped"] is False
    assert abs(d["value"] - B / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3 and d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.001
    named = {"combine_g2_t3": d["roofline"], "general_path": d["general_path"]["roofline"], "wire": d["wire"]["roofline"]}
    named.update({k: d["secondary_rooflines"][k] for k in ("hash_g2", "g2_sign", "threshold_decrypt", "ciphertext_verify", "pairing_check")})
    for key, r in named.items():
        assert 0 < r["frac_useful"] <= r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3, key
        want = r["executed_macs_per_unit"] * r["units_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12
        assert abs(r["achieved"] - want) / want < 2e-3 and r["useful_macs_per_unit"] <= r["executed_macs_per_unit"], key
        assert r["traffic"] > 50 * r["algorithmic_bytes_per_launch"], key
        assert abs(r["traffic_over_algorithmic_bytes"] - r["traffic"] / r["algorithmic_bytes_per_launch"]) < 0.06, key
        assert 0.4 < r["executed_cross_check"]["implied_v_mad_share_of_valu"] < 0.85, key
        # the same launches under rocprofv3 last as long as the line's HIP events say (the first launches of a process run a
        # little slower, and the 3 ms G1 leg shows it most: 2.87-3.11 ms over its four profiled launches, 2.72-2.79 in the line)
        assert abs(r["profile_kernel_ms"] - r["kernel_ms"]) / r["kernel_ms"] < 0.12, key
        src = "profiles/%s_%s_rocprofv3_summary.csv" % (tag, {"combine_g2_t3": "combine", "pairing_check": "verify_g2"}.get(key, key))
        assert src in r["traffic_is"] and os.path.exists(os.path.join(root, src)), key
        if constants:
            assert prof[key]["source"] == src and r["traffic"] == prof[key]["traffic_bytes"], key
    assert d["roofline"]["executed_macs_per_unit"] == macs["combine_g2_t3_fast"] and d["roofline"]["useful_macs_per_unit"] == useful["combine_g2_t3_fast"]
    w = d["wire"]["roofline"]
    assert w["executed_macs_per_unit"] == (t + 1) * macs["g2_decompress_x2"] + macs["combine_g2_t3_fast"] and "k_decompress_take_g2_x2" in w["kernel"]
    assert w["useful_macs_per_unit"] == (t + 1) * useful["g2_decompress_x2"] + useful["combine_g2_t3_fast"]
    assert macs["g2_decompress_x2"] < 0.6 * macs["g2_decompress"] and useful["g2_decompress_x2"] > 0.99 * macs["g2_decompress_x2"]
    assert d["wire"]["ms_per_step"] < 14.0 and d["wire"]["value"] > 4.6e6        # 16.2 ms / 4.04 M/s in round 4
    # the CPU path timed beside BOTH halves of BASELINE's metric and the legs around them (Oracle B on all host threads: kind "port")
    for leg, unit in (("config3", "pairing_verifies/s"), ("config4", "threshold_decryptions/s"), ("wire", "combine_signatures/s")):
        c = d[leg]["cpu_baseline"]
        assert c["kind"] == "port" and c["unit"] == unit and c["cores"] >= 1 and "/root/reference/src/lib.rs" in c["reference"], leg
        assert "8192 jobs" in c["sample"] and c["value"] > c["single_thread_per_s"] and d[leg]["value"] / c["value"] > 100, leg
    assert d["cpu_baseline"]["kind"] == "port" and d["value"] / d["cpu_baseline"]["value"] > 100
    assert c5["config"]["t"] == 67 and c5["config"]["N"] == 200 and c5["verified_all"] is True and c5["config"]["batch_per_gpu"] == 131072
    for key, leg in (("config5_sign", "share_sign"), ("config5_combine", "combine"), ("config5_verify", "pairing_check")):
        r = c5["secondary_rooflines"][leg]
        assert "profiles/%s_config5_rocprofv3_summary.csv" % tag in r["traffic_is"], key
        if constants:
            assert r["traffic"] == prof[key]["traffic_bytes"] and prof[key]["source"] == "profiles/%s_config5_rocprofv3_summary.csv" % tag, key
        assert abs(r["profile_kernel_ms"] - r["kernel_ms"]) / r["kernel_ms"] < 0.05 and 0.4 < r["executed_cross_check"]["implied_v_mad_share_of_valu"] < 0.85, key


def _check_round5_lines(root, macs):
    import json
    d = json.loads([l for l in open(os.path.join(root, "profiles", R05_TAG + "_bench.txt")) if l.startswith("{")][-1])
    c5 = json.loads([l for l in open(os.path.join(root, "profiles", R05_TAG + "_config5_bench.txt")) if l.startswith("{")][-1])
    _check_leg_lines(root, macs, R05_TAG, d, c5, constants=False)


def _check_round6_lines(root, macs):
    """round 6 (BENCH_r05.json came back `parsed: null` on a 23.6 KB line): the committed bench output is what
    the driver sees -- ONE compact JSON object under 8 000 bytes that carries the contract's fields, a compact `roofline` and
    `cpu_baseline`, one object per secondary BASELINE configuration -- and the detail object beside it (bench_detail.json) obeys
    everything round 5's 23 KB line obeyed; the compact line is a projection of it."""
    import json
    import benchline
    for name in ("_bench", "_config5_bench"):
        text = open(os.path.join(root, "profiles", R06_TAG + name + ".txt")).read()
        detail = json.load(open(os.path.join(root, "profiles", R06_TAG + name + "_detail.json")))
        line, _ = benchline.parse(text)
        assert len(text.splitlines()[-1]) < 8000 and json.loads(text[-8000:].splitlines()[-1]) == line
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling"):
            assert line[k] == detail[k], (name, k)
        for k in ("kernel_ms", "frac", "frac_useful", "achieved", "peak", "traffic", "executed_macs_per_unit", "algorithmic_bytes_per_launch"):
            assert line["roofline"][k] == detail["roofline"][k], (name, k)
        assert line["roofline"]["bound"] == "valu_int32_mac" and line["roofline"]["unit"] == "TMAC/s" and line["ranks"]["world_size"] == 1
        if name == "_bench":
            d = detail
            assert line["config"]["t"] == 3 and line["config"]["N"] == 10 and line["config"]["batch_per_gpu"] == 65536 and "batch=65536" in line["config"]["workload"]
            for k in ("value", "unit", "cores", "kind"):
                assert line["cpu_baseline"][k] == detail["cpu_baseline"][k], k
            for leg in ("config3", "config4", "wire"):
                assert line[leg]["value"] == detail[leg]["value"] and line[leg]["frac"] == detail[leg]["roofline"]["frac"], leg
                assert line[leg]["cpu_baseline"]["value"] == detail[leg]["cpu_baseline"]["value"] and line[leg]["kernel_ms"] > 0, leg
        else:
            c5 = detail
            assert line["config"]["t"] == 67 and line["config"]["N"] == 200 and line["valid_total_all_ranks"] == 131072
    _check_leg_lines(root, macs, R06_TAG, d, c5, constants=True)
    # the form labels of the line are the library's own thresholds (tc_ctx_get_tuning), not constants of bench.py
    assert d["secondary_rooflines"]["hash_g2"]["kernel"] == "k_hash_g2" and "k_hash_g1_g2 +" in d["secondary_rooflines"]["ciphertext_verify"]["kernel"]


def _check_round4_lines(root, macs):
    """round 4: the pairing check is three kernels (prepared lines), the line has a `wire` leg, the profile constants bench.py
    embeds come from the capture committed with the line, and config 5 has run at its full 1 048 576 jobs on one GPU"""
    import json
    d = json.loads([l for l in open(os.path.join(root, "profiles", "r04_f_bench.txt")) if l.startswith("{")][-1])   # the final build (r04_a: mid-round)
    B = d["config"]["batch_per_gpu"]
    assert d["metric"] == "combine_signatures/sec" and d["n_gpus"] == 1 == d["ranks"]["world_size"] and d["vs_baseline"] is None
    assert d["config"]["overlapped"] is False and abs(d["value"] - B / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3
    assert d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.001 and d["streaming"]["value"] > d["value"]
    legs = [d["roofline"], d["general_path"]["roofline"], d["config3"]["roofline"], d["config4"]["roofline"], d["wire"]["roofline"]]
    legs += list(d["secondary_rooflines"].values())
    for r in legs:
        assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3 and r["hbm_frac"] < 0.01
        want = r["executed_macs_per_unit"] * r["units_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12
        assert abs(r["achieved"] - want) / want < 2e-3
        assert 36.0 < r["peak"] < 38.5                        # ONE denominator: the saturated v_mad_i64_i32 wall rate
    assert d["roofline"]["executed_macs_per_unit"] == macs["combine_g2_t3_fast"]
    assert d["config3"]["roofline"]["executed_macs_per_unit"] == macs["verify_g2_prepared"] == macs["verify_g2"]
    assert "k_miller_lines + k_miller_accumulate + k_final_exp" in d["config3"]["roofline"]["kernel"]
    t = d["config"]["t"]
    assert d["wire"]["roofline"]["executed_macs_per_unit"] == (t + 1) * macs["g2_decompress"] + macs["combine_g2_t3_fast"]
    assert abs(d["wire"]["value"] - B / (d["wire"]["ms_per_step"] * 1e-3)) / d["wire"]["value"] < 2e-3
    assert d["wire"]["value"] < d["value"]                    # the checked decode of t + 1 shares is most of the wire call
    assert d["config3"]["roofline"]["traffic"] < 30e9        # was 41.7 GB in round 3 (the constants file now belongs to round 5's capture)
    summary = open(os.path.join(root, "profiles", "r04_f_rocprofv3_summary.csv")).read()
    avg = {}
    for line in summary.splitlines():
        f = line.split(",")
        name = f[0].replace("void ", "")
        if len(f) >= 12 and name in ("tc::k_combine_fast<tc::Fq2>", "tc::k_miller_lines", "tc::k_miller_accumulate", "tc::k_final_exp") and name not in avg:
            avg[name] = float(f[3])
    pair_ms = avg["tc::k_miller_lines"] + avg["tc::k_miller_accumulate"] + avg["tc::k_final_exp"]
    assert abs(pair_ms - d["config3"]["kernel_ms"]) / pair_ms < 0.05
    assert abs(avg["tc::k_combine_fast<tc::Fq2>"] - d["roofline"]["kernel_ms"]) / d["roofline"]["kernel_ms"] < 0.08
    for k in ("k_g1_mul_arena", "k_hash_g1_g2", "k_msm_ladder", "SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
        assert k in summary, k
    # the line also says which clock its legs ran at (rocm-smi sampled during the sustained legs): the denominator's loop runs
    # at ~2.36 GHz, the kernels at 2.1-2.25 GHz and up to 1.25 kW; frac_at_kernel_clock takes that out and nothing else
    for b in (d, json.loads([l for l in open(os.path.join(root, "profiles", "r04_b_bench.txt")) if l.startswith("{")][-1])):
      for leg, sus in ((b["roofline"], b["sustained"]), (b["config3"]["roofline"], b["config3"]["sustained"])):
        clk = sus["clock"]
        assert 1.5 < clk["sclk_GHz"] < leg["peak_clock_GHz"] <= 2.45 and clk["samples"] >= 3 and 300 < clk["power_W"] < 1500
        assert abs(leg["frac_at_kernel_clock"] - leg["frac"] * leg["peak_clock_GHz"] / clk["sclk_GHz"]) < 2e-3
        assert leg["frac"] < leg["frac_at_kernel_clock"] <= 1
      assert abs(b["config3"]["sustained"]["value"] - b["config3"]["value"]) / b["config3"]["value"] < 0.05
    # ... and what the shipped product alone reaches at the kernels' occupancy: the practical ceiling of frac, measured live
    e = json.loads([l for l in open(os.path.join(root, "profiles", "r04_d_bench.txt")) if l.startswith("{")][-1])
    pc = e["product_ceiling"]
    assert 0.6 < pc["product_frac_1_wave"] < pc["product_frac_2_waves"] < 0.9
    assert e["roofline"]["frac"] < e["config3"]["roofline"]["frac"] < pc["product_frac_2_waves"]
    full = json.loads([l for l in open(os.path.join(root, "profiles", "r04_config5_full.txt")) if l.startswith("{")][-1])
    assert full["config"]["emulated_world"] == 8 and full["n_gpus"] == 1 and full["config"]["batch_per_gpu"] == 1048576
    assert full["valid_total_all_ranks"] == 1048576 and full["verified_all"] is True
    recs = full["rank_records_start_jobs_valid_digest"]
    assert [r[:3] for r in recs] == [[131072 * i, 131072, 131072] for i in range(8)] and len({r[3] for r in recs}) == 8
    assert abs(full["value"] - 1048576 / (full["ms_per_step"] * 1e-3)) / full["value"] < 2e-3
    one = json.loads([l for l in open(os.path.join(root, "profiles", "r04_config5_1gpu_bench.txt")) if l.startswith("{")][-1])
    assert one["config"]["t"] == 67 and one["config"]["N"] == 200 and one["verified_all"] is True and one["config"]["batch_per_gpu"] == 131072
    assert recs[0][3] == one["rank_records_start_jobs_valid_digest"][0][3]       # slice 0 of the emulation IS the 1-GPU run's slice
    assert abs(full["value"] - one["value"]) / one["value"] < 0.03               # eight slices one after the other cost eight slices


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
    # round 3: the headline is K steps on ONE stream (per-launch event times measured inside the timed region), the two-contexts
    # figure is the `streaming` object; a mechanical reader's check kernel_ms <= ms_per_step holds; every leg of configs 2-4
    # carries its own roofline
    d = json.loads([l for l in open(os.path.join(root, "profiles", "r03_c_bench.txt")) if l.startswith("{")][-1])
    B = d["config"]["batch_per_gpu"]
    assert d["metric"] == "combine_signatures/sec" and d["n_gpus"] == 1 == d["ranks"]["world_size"] and d["vs_baseline"] is None
    assert d["config"]["overlapped"] is False and d["config"]["steps_in_flight"] == 1
    assert abs(d["value"] - B / (d["ms_per_step"] * 1e-3)) / d["value"] < 2e-3
    assert d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.001
    assert d["streaming"]["overlapped"] is True and d["streaming"]["value"] > d["value"]
    assert d["sustained"]["seconds"] >= 1.0 and abs(d["sustained"]["value"] - d["value"]) / d["value"] < 0.05
    legs = [d["roofline"], d["general_path"]["roofline"], d["config3"]["roofline"], d["config4"]["roofline"]] + list(d["secondary_rooflines"].values())
    for r in legs:
        assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
        want = r["executed_macs_per_unit"] * r["units_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12
        assert abs(r["achieved"] - want) / want < 2e-3 and r["hbm_frac"] < 0.01
    assert d["roofline"]["executed_macs_per_unit"] == macs["combine_g2_t3_fast"]
    assert d["general_path"]["roofline"]["executed_macs_per_unit"] == macs["combine_g2_t3_general"]
    assert d["config3"]["roofline"]["executed_macs_per_unit"] == macs["verify_g2"] and d["config3"]["kernel_ms"] == d["config3"]["roofline"]["kernel_ms"]
    assert d["config4"]["roofline"]["executed_macs_per_unit"] == macs["verify_g2"] + macs["hash_g1_g2"] + macs["combine_g1_t3_fast"]
    assert d["cpu_baseline"]["kind"] == "port" and d["value"] / d["cpu_baseline"]["value"] > 100
    assert d["roofline"]["traffic"] > 50 * d["roofline"]["algorithmic_bytes_per_launch"] and "profiles/" in d["roofline"]["traffic_is"]
    summary = open(os.path.join(root, "profiles", "r03_c_rocprofv3_summary.csv")).read()
    for k in ("k_combine_fast<tc::Fq2>", "k_miller_loop", "k_final_exp", "k_hash_g2", "k_hash_g1_g2", "k_g2_mul_shared", "SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
        assert k in summary, k
    # frac reproducible from the profile within 5 %: executed multiply-adds / the profile's average duration
    avg = {}
    for line in summary.splitlines():
        f = line.split(",")
        if len(f) >= 12 and f[0].replace("void ", "") in ("tc::k_combine_fast<tc::Fq2>", "tc::k_miller_loop", "tc::k_final_exp") and f[0] not in avg:
            avg[f[0].replace("void ", "")] = float(f[3])
    pair_ms = avg["tc::k_miller_loop"] + avg["tc::k_final_exp"]
    assert abs(pair_ms - d["config3"]["kernel_ms"]) / pair_ms < 0.05
    assert abs(avg["tc::k_combine_fast<tc::Fq2>"] - d["roofline"]["kernel_ms"]) / d["roofline"]["kernel_ms"] < 0.08   # (+ the small grouping kernels)
    _check_round4_lines(root, macs)
    _check_round5_lines(root, macs)
    _check_round6_lines(root, macs)
    c5 = json.loads([l for l in open(os.path.join(root, "profiles", "r03_config5_1gpu_bench.txt")) if l.startswith("{")][-1])
    assert c5["config"]["t"] == 67 and c5["config"]["N"] == 200 and c5["verified_all"] is True and c5["ranks"]["world_size"] == 1
    c5 = json.loads([l for l in open(os.path.join(root, "profiles", "r02_config5_1gpu_bench.txt")) if l.startswith("{")][-1])
    assert c5["config"]["t"] == 67 and c5["config"]["N"] == 200 and c5["verified_all"] is True
    assert abs(c5["value"] - c5["config"]["batch_per_gpu"] / (c5["ms_per_step"] * 1e-3)) / c5["value"] < 2e-3
    assert abs(sum(c5["phase_kernel_ms"].values()) - c5["ms_per_step"]) / c5["ms_per_step"] < 0.02   # the step IS its three kernels' time


# ---- bench.py --gpus N starts N ranks itself ---------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_bench(*argv, timeout=900):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), capture_output=True, text=True, timeout=timeout, cwd=root, env=env)
    # stdout = the driver's compact line and nothing else (tests/benchline.py checks it the way the driver reads it); the tests
    # below look at the DETAIL object (stderr, tagged), of which the compact line is a projection
    import benchline
    out.compact, detail = benchline.parse(out.stdout, out.stderr)
    return out, ([detail] if detail is not None else [])


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` as the driver calls it (no torchrun around it, no WORLD_SIZE in the environment) must
    start two ranks, not run one and print n_gpus 1: the flow runs end to end on two gloo ranks with the host build of the
    device source standing in for the GPU (bench.py --test-engine: a test harness, marked in the line), and the line's
    n_gpus is the world size the ranks really joined."""
    out, lines = _run_bench("--gpus", "2", "--backend", "gloo", "--test-engine", "hostsim", "--batch", "3", "--t", "1", "--signers", "3",
                            "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1                                    # ONE JSON line, rank 0's
    d = lines[0]
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and d["ranks"]["devices"] == ["cpu:0", "cpu:1"]
    assert "self-spawn" in d["ranks"]["launched_by"] and d["ranks"]["backend"] == "gloo" and "test_harness" in d
    assert out.compact["n_gpus"] == 2 and out.compact["ranks"]["distinct_devices"] == 2 and "test_harness" in out.compact
    assert d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["batch_per_gpu"] == 3 and d["verified_all"] is True
    assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01        # whole-job rate: both ranks' jobs
    assert d["pairing_verify_valid_count_all_ranks"] == 2 * 2                              # 3 jobs per rank, job 0 of each corrupted


def test_bench_gpus_2_config5_on_two_ranks():
    out, lines = _run_bench("--gpus", "2", "--config", "5", "--backend", "gloo", "--test-engine", "hostsim", "--batch", "2", "--t", "8",
                            "--signers", "12", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert out.returncode == 0, out.stderr[-3000:]
    d = lines[0]
    assert len(lines) == 1 and d["n_gpus"] == 2 and d["config"]["t"] == 8 and d["valid_total_all_ranks"] == 4
    assert [r[:3] for r in d["rank_records_start_jobs_valid_digest"]] == [[0, 2, 2], [2, 2, 2]]


def test_bench_gpus_8_rank_logic_on_gloo():
    """The driver's largest shape, `bench.py --gpus 8`: eight ranks (gloo, host build) -- contiguous shards of the global job range, one
    record per rank gathered in rank order, counts summed over all eight, for config 2 and config 5."""
    out, lines = _run_bench("--gpus", "8", "--backend", "gloo", "--test-engine", "hostsim", "--batch", "2", "--t", "1", "--signers", "3",
                            "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert out.returncode == 0, out.stderr[-3000:]
    d = lines[0]
    assert len(lines) == 1 and d["n_gpus"] == 8 and d["ranks"]["devices"] == ["cpu:%d" % r for r in range(8)]
    assert d["verified_all"] is True and d["pairing_verify_valid_count_all_ranks"] == 8 and d["config"]["batch_per_gpu"] == 2
    out, lines = _run_bench("--gpus", "8", "--config", "5", "--backend", "gloo", "--test-engine", "hostsim", "--batch", "1", "--t", "8",
                            "--signers", "12", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert out.returncode == 0, out.stderr[-3000:]
    d = lines[0]
    assert len(lines) == 1 and d["n_gpus"] == 8 and d["valid_total_all_ranks"] == 8
    assert [r[:3] for r in d["rank_records_start_jobs_valid_digest"]] == [[r, 1, 1] for r in range(8)]
    # the eight-rank line stays well inside the driver's 8 KB (device list and full digests live in the detail object)
    assert out.compact["ranks"]["distinct_devices"] == 8 and len(out.stdout.splitlines()[-1]) < 3000
    assert [r[:3] for r in out.compact["rank_records"]] == [[r, 1, 1] for r in range(8)]
    assert all(str(full[3]).startswith(short[3]) for full, short in zip(d["rank_records_start_jobs_valid_digest"], out.compact["rank_records"]))


def test_config5_emulated_world_equals_the_eight_rank_run():
    """config5.run_emulated_world (every rank slice of an 8-rank job through ONE engine, one after the other -- how one GPU runs
    BASELINE config 5 at its stated 1 048 576 jobs) cuts the same slices and produces the same signatures as eight real ranks:
    its per-slice records equal the committed records of the 8-rank gloo run (tests/golden/config5_world8_reduced.json,
    tools/gen_config5_digests.py), and `bench.py --config 5 --emulate-world 8` reports them on one rank."""
    import json
    from threshold_crypto_amd import config5
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = json.load(open(os.path.join(root, "tests", "golden", "config5_world8_reduced.json")))
    emu = config5.run_emulated_world(HostSimEngine(), gold["t"], gold["N"], gold["batch_per_rank"], gold["world"])
    assert emu["records"] == gold["records"] and emu["status_errors"] == 0 and emu["valid_total"] == gold["world"] * gold["batch_per_rank"]
    out, lines = _run_bench("--config", "5", "--emulate-world", "8", "--backend", "gloo", "--test-engine", "hostsim", "--batch", "2", "--t", "8",
                            "--signers", "12", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert out.returncode == 0, out.stderr[-3000:]
    d = lines[0]
    assert d["n_gpus"] == 1 and d["config"]["emulated_world"] == 8 and d["config"]["batch_per_gpu"] == 16
    assert d["rank_records_start_jobs_valid_digest"] == gold["records"] and d["valid_total_all_ranks"] == 16
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01


def test_bench_under_a_launcher_joins_the_rendezvous_even_as_the_only_rank():
    """The driver's N>1 command shape (`python -m torch.distributed.run ... bench.py --gpus N`) with N = 1: bench.py joins the
    launcher's rendezvous and takes every collective (key-set broadcast, barriers, MAX of the timed region, the count
    all-reduce) through the process group -- the same run goes through RCCL on a one-GPU box (tests/test_gpu_api.py)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "1", "--backend", "gloo", "--test-engine", "hostsim", "--batch", "3", "--t", "1", "--signers", "3",
           "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    import benchline
    line, d = benchline.parse(out.stdout, out.stderr)
    assert line["ranks"] == {"world_size": 1, "backend": "gloo", "rccl_version": None, "distinct_devices": 1, "launched_by": line["ranks"]["launched_by"]}
    assert d["n_gpus"] == 1 and d["ranks"]["backend"] == "gloo" and d["ranks"]["devices"] == ["cpu:0"]
    assert "external launcher" in d["ranks"]["launched_by"] and d["verified_all"] is True


def test_bench_refuses_more_gpus_than_the_node_has():
    """--gpus N on a node with fewer GPUs fails loudly instead of reporting an N-GPU number (this container has none)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this node has two GPUs")
    out, lines = _run_bench("--gpus", "2", "--steps", "1", timeout=300)
    assert out.returncode != 0 and not lines and "refusing" in out.stderr
    out, lines = _run_bench("--gpus", "2", "--backend", "gloo", timeout=60)          # gloo without the test engine: no CPU path
    assert out.returncode != 0 and not lines


def test_commitment_evaluate_batch_never_re_enters_itself():
    """abscissae outside [1, 2^64 - 1] used to recurse forever; they are routed by value (IntoFr: modulo r):
    0 -> coefficient 0, 1 .. 2^64 - 1 -> the Horner kernel with idx = x - 1, everything else -> a linear combination."""
    from threshold_crypto_amd import poly

    class StubEngine:
        def public_key_shares(self, c, idx):
            self.idx = idx.tolist()
            return np.full((len(idx), 96), 1, np.uint8), np.zeros(len(idx), np.uint8)

        def lincomb_g1(self, sc, pts):
            self.scalars = [[int.from_bytes(bytes(r), "little") for r in row] for row in sc]
            return np.full((sc.shape[0], 96), 2, np.uint8), np.zeros(sc.shape[0], np.uint8)

    e = StubEngine()
    c = poly.Commitment([bytes([3]) * 96, bytes([4]) * 96, bytes([5]) * 96], _trusted=True)
    r = c.evaluate_batch([0, 1, 2 ** 64, -1, 2 ** 64 - 1], e)
    assert [x[0] for x in r] == [3, 1, 2, 2, 1]
    assert e.idx == [0, 2 ** 64 - 2]
    R = poly._R
    assert e.scalars == [[1, 2 ** 64, 2 ** 128 % R], [1, R - 1, 1]]


def test_share_maps_iterate_like_the_reference_btreemap():
    """PublicKeySet::combine_signatures / decrypt take `IntoIterator<Item = (T, &Share)>` and callers pass BTreeMaps: interpolate()
    uses the FIRST t + 1 samples in the map's order (/root/reference/src/lib.rs:727-730).  The Python mirror orders a dict the way
    the BTreeMap of ITS key type would: integers by their signed value (BTreeMap<i64, _>: negatives first), `Fr` keys by their
    canonical value (BTreeMap<Fr, _>: the images of negative integers sort LAST); a dict mixing the two has no counterpart in the
    reference and is refused; sequences of pairs keep the caller's order."""
    from threshold_crypto_amd import api
    assert [k for k, _ in api._ordered({3: "a", -2: "b", 0: "c", -(2 ** 40): "d"})] == [-(2 ** 40), -2, 0, 3]
    frs = {api.Fr(-2): "b", api.Fr(3): "a", api.Fr(0): "c"}
    assert [int(k) for k, _ in api._ordered(frs)] == [0, 3, api._R - 2]
    with pytest.raises(TypeError):
        api._ordered({api.Fr(1): "x", 2: "y"})
    assert api._ordered([(5, "x"), (-1, "y")]) == [(5, "x"), (-1, "y")]


def test_compact_bench_line_never_exceeds_the_limit():
    """bench.compact() (the line the driver parses) keeps the contract's fields, `roofline`, `cpu_baseline` and `ranks` whatever
    happens and drops its OPTIONAL groups -- per-rank records first -- before it would reach 8 000 bytes: a 512-rank record list,
    oversized extras and secondary legs on top of round 5's real 23 KB line still give a parseable line under the limit."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    d = json.loads([l for l in open(os.path.join(root, "profiles", "r05_c_bench.txt")) if l.startswith("{")][-1])
    base = bench.compact(d)
    assert len(base) < 4000 and json.loads(base)["secondary"]["hash_g2"]["frac"] == d["secondary_rooflines"]["hash_g2"]["frac"]
    big = dict(d)
    big["rank_records_start_jobs_valid_digest"] = [[131072 * r, 131072, 131072, "%064x" % r] for r in range(512)]
    big["extras"] = dict(d["extras"], **{"extra_number_%03d" % i: float(i) for i in range(300)})
    big["secondary_rooflines"] = dict(d["secondary_rooflines"], **{"leg_%03d" % i: d["secondary_rooflines"]["hash_g2"] for i in range(120)})
    big["ranks"] = {"world_size": 512, "backend": "nccl", "rccl_version": "2.26.6", "devices": ["AMD Instinct MI355X gpu%d uuid-%032x" % (r, r) for r in range(512)],
                    "launched_by": "torch.distributed.run (external launcher)"}
    text = bench.compact(big)
    line = json.loads(text)
    assert len(text) < bench.LINE_LIMIT
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "ranks"):
        assert k in line, k
    assert line["ranks"]["distinct_devices"] == 512 and "devices" not in line["ranks"]
    assert line["roofline"]["frac"] == d["roofline"]["frac"] and line["cpu_baseline"]["value"] == d["cpu_baseline"]["value"]
    assert "rank_records" not in line and "secondary" not in line          # the optional groups went first
    assert line["config3"]["value"] == d["config3"]["value"]               # ... the BASELINE configurations stayed
