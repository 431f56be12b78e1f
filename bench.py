#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|5] [--batch B] [--t T] [--signers N]

--gpus N with N > 1 STARTS N ranks: when the process was not itself started by torch.distributed.run (no
WORLD_SIZE in the environment) it re-executes itself under `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU over RCCL, after checking that the node has N GPUs; the
JSON line's `n_gpus` is the world size the ranks really joined (dist.get_world_size()), with the RCCL version and
every rank's device identity next to it.

--config 2 (default; BASELINE "t=3, N=10, batch=65 536 threshold signatures on 1xMI355X"): a "step" is one pass of
PublicKeySet::combine_signatures (src/lib.rs:608-615) over `batch` independent (message, share-set) jobs through
the C ABI (tc_combine_g2_batch) with every input already resident in HBM.  The K timed steps are queued on ONE
context (one HIP stream: launches do not overlap) and bracketed by barrier + synchronize: `value` / `ms_per_step`.
The per-launch kernel time of the roofline comes from HIP events recorded around every one of those K steps on
that stream, inside the timed region (`roofline.kernel_ms` <= `ms_per_step` by construction).  `streaming` is the same step with TWO contexts in flight (two streams:
consecutive launches overlap at their ends, "overlapped": true), `sustained` the one-context rate over >= 1 s,
`general_path` the same batch with share indices the small-index fast path does not take.  The batch is then
verified (config 3: PublicKey::verify_g2, src/lib.rs:108-110; every 16th signature replaced by its neighbour's, so
the expected ok-vector is known), re-signed, verified with hashing on the device, and run through the
threshold-decryption path (config 4: Ciphertext::verify + PublicKeySet::decrypt); each leg carries its own roofline
object and ANY failing leg fails the run.  Beside `frac` (executed multiply-adds / the live-measured v_mad_i64_i32 peak) the line says
what bounds it from above and what it was charged for: `product_ceiling` = what the shipped lane-pair product alone reaches at one /
two waves per SIMD (tools/ubench_product --ceiling), `sustained.clock` / `config3.sustained.clock` = shader clock and board power
sampled with rocm-smi while the leg loops, `roofline.frac_at_kernel_clock` = frac x peak clock / that clock, `wire` =
the bytes-in / bytes-out combination (tc_combine_signatures_wire_batch), and with --latency-table one call's latency at B = 1 ... 4 096
against one CPU core.

--config 5 (BASELINE "t=67, N=200, batch=1 048 576 mixed sign+combine+verify sharded across 8 GPUs"): one step =
sign the t+1 selected shares of every job ON the device, combine them, verify the result; 131 072 jobs per GPU (weak
scaling: --gpus 8 is the BASELINE batch), nothing larger than the key set crosses PCIe or xGMI.

One process per GPU; for N > 1 the batch is per-rank (weak scaling), the key-set parameters are broadcast from rank 0
over RCCL and the per-rank valid counts are all-reduced; there is no data-path collective (jobs are independent).
Rank 0 prints ONE compact JSON line to stdout -- the only thing this script writes there, < 8 000 bytes at any world size (what the
driver parses) -- after the full DETAIL object has gone to bench_detail.json and, tagged `bench_detail `, to stderr.  The oracle (oracle/) is used only for the cpu_baseline leg and to check the GPU
output of the timed batch bit-for-bit.

--backend gloo --test-engine hostsim (tests/test_host_logic.py): the same rank start-up, rendezvous, broadcast,
sharding and line bookkeeping on CPU ranks with tests/hostsim (the g++ build of the device source) standing in for
the GPU -- a TEST HARNESS for the N-rank path in the GPU-less container; its line is marked "test_harness" and is
not a measurement.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# frozen reference-algorithm work constants (Fq multiplications+squarings per unit), measured with
# Oracle B's counter (oracle/c/tc_oracle.c or_fq_mul_count; DESIGN.md "Work constants"); 1 Fq-mul = 300
# 32-bit multiply-adds (12x12 product + 12x12 reduction + 12 quotient digits, CIOS: SURVEY 8d)
W_FQMUL = {"combine_g2_t3": 31148, "g2_mul": 7771, "verify_g2": 41226, "hash_g2": 19604, "combine_g1_t3": 12800,
           "combine_g2_t67": 530000, "ciphertext_verify": 41226 + 19604}
MAC_PER_FQMUL = 300
# what the kernels execute per unit, in v_mad_i64_i32 lane-instructions (one 14x14 limb product or one
# Montgomery reduction = 196): counted by running the same per-lane job bodies in the host build
# (tests/count_ops.py).  Two lanes work on a G2 job; work inside Fq2 operations is split between them, Fq
# work outside (inversions, root exponentiations) is done by both and counted twice.
CPU_LEG_JOBS = 8192   # sample of the timed batch the secondary CPU legs (config 3, config 4, wire) run: 2-6 s each on 16 threads
EXECUTED_MACS = json.load(open(os.path.join(ROOT, "profiles", "executed_macs.json")))
# the same with the work both lanes of a pair repeat identically counted ONCE (tests/count_ops.py --json-useful): `frac_useful`
USEFUL_MACS = json.load(open(os.path.join(ROOT, "profiles", "useful_macs.json")))
# algorithmic bytes per unit (canonical uncompressed affine I/O)
ALG_BYTES = {"combine_g2": lambda t: (t + 1) * (192 + 8) + 192, "combine_g2_wire": lambda t: (t + 1) * (96 + 8) + 96, "verify_g2": lambda t: 385, "g2_mul": lambda t: 192,
             "hash_g2": lambda t: 207, "combine_g1": lambda t: (t + 1) * (96 + 8) + 96 + 32,
             "ciphertext_verify": lambda t: 96 + 32 + 192 + 1}
# Roofline peak: the issue rate of the multiplier's own instruction (v_mad_i64_i32) with every SIMD full,
# measured LIVE by tools/ubench_chain --peak when that binary is present (sustained ~40 ms launches, clock
# read from s_memtime / s_memrealtime); otherwise the recorded value.
PEAK_RECORDED = {"tmacs": 37.5, "clock_ghz": 2.1, "source": "profiles/r02_ubench_chain.txt (v_mad_i64_i32 chains, >= 2 waves/SIMD)"}
HBM_PEAK_GBPS = 8000.0
# L2<->fabric bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, separate PMC passes, FETCH doubled per the
# gfx950 note of MI355X_MICROARCH.md) and SQ_INSTS_VALU per launch at batch 65 536, from the profile named
PROFILE = json.load(open(os.path.join(ROOT, "profiles", "profile_constants.json")))


class ClockSampler:
    """Shader clock and board power while a leg runs (`rocm-smi`, one sample per ~0.2 s on a side thread; rank 0 only).  The
    roofline denominator is measured in a multiply-add-only loop at 2.36 GHz; the kernels themselves run at 2.1-2.25 GHz
    and up to 1.25 kW (tools/clock_probe.py), so a leg's `frac` also charges it for the clock it ran at:
    frac_at_kernel_clock = frac x peak clock / sampled clock says what is left once that is taken out."""

    def __init__(self, enabled):
        import shutil
        import threading
        self.rows, self.stop = [], threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True) if (enabled and shutil.which("rocm-smi")) else None

    @staticmethod
    def sample():
        import re
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            card = d[sorted(k for k in d if k.startswith("card"))[int(os.environ.get("LOCAL_RANK", "0"))]]
            sclk = next((v for k, v in card.items() if "sclk clock speed" in k.lower()), None)
            pw = next((v for k, v in card.items() if "power (w)" in k.lower()), None)
            m = re.search(r"(\d+)\s*[Mm][Hh]z", str(sclk))
            return (int(m.group(1)) if m else None, float(pw) if pw not in (None, "N/A") else None)
        except Exception:
            return (None, None)

    def _run(self):
        while not self.stop.is_set():
            self.rows.append(self.sample())
            self.stop.wait(0.15)

    def __enter__(self):
        if self.thread:
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        if self.thread:
            self.thread.join(timeout=10)

    def result(self):
        clk = sorted(c for c, _ in self.rows[2:] if c)           # the first samples may still see the idle clock
        pw = sorted(p for _, p in self.rows[2:] if p)
        if len(clk) < 3:
            return None
        return {"sclk_GHz": round(clk[len(clk) // 2] / 1e3, 3), "sclk_GHz_min_max": [round(clk[0] / 1e3, 3), round(clk[-1] / 1e3, 3)],
                "power_W": pw[len(pw) // 2] if pw else None, "samples": len(clk), "source": "rocm-smi --showclocks --showpower, sampled while the leg ran"}


def measure_peak():
    """The roofline denominator: the wall rate of v_mad_i64_i32 -- the multiplier's own instruction, issued the way the field
    multiplier issues it -- with every SIMD saturated, from ~40 ms launches (the clock the chip sustains).  Two loop shapes are
    measured live, tools/ubench_chain --peak (168 multiply-adds per iteration over 14 x 14 operand limbs) and tools/ubench_issue
    --peak (64 per iteration, with the sustained clock and the cycles one wave-instruction occupies a SIMD); the LARGER rate is
    the peak (a larger denominator can only lower `frac`)."""
    best = None
    try:
        out = subprocess.run([os.path.join(ROOT, "tools", "ubench_issue"), "--peak"], capture_output=True, text=True, timeout=120).stdout
        d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
        best = {"tmacs": d["T_lane_ops_per_s"], "clock_ghz": d["clock_GHz"], "cycles_per_wave_instr_per_simd": d["cycles_per_wave_instr_per_simd_wall"],
                "issue_interval_per_wave_cycles": d["issue_interval_per_wave_cycles_median"],
                "source": "measured live: tools/ubench_issue --peak (v_mad_i64_i32, chained accumulators, 4 waves/SIMD, %.1f ms launch)" % d["ms"]}
    except Exception:
        pass
    try:
        out = subprocess.run([os.path.join(ROOT, "tools", "ubench_chain"), "--peak"], capture_output=True, text=True, timeout=120).stdout
        for line in out.splitlines():
            d = json.loads(line)
            if d.get("op") == "v_mad_i64_i32_chained" and d.get("waves_per_simd") == 4 and (best is None or d["T_mad_per_s"] > best["tmacs"]):
                extra = {k: v for k, v in (best or {}).items() if k in ("clock_ghz", "cycles_per_wave_instr_per_simd", "issue_interval_per_wave_cycles")}
                if extra.get("clock_ghz"):   # cycles per wave-instruction at THIS rate and the measured clock
                    extra["cycles_per_wave_instr_per_simd"] = round(1024 * extra["clock_ghz"] * 1e9 / (d["T_mad_per_s"] * 1e12 / 64), 3)
                best = dict({"tmacs": d["T_mad_per_s"], "clock_ghz": None,
                             "source": "measured live: tools/ubench_chain --peak (v_mad_i64_i32, 14 x 14 operand limbs, two chained accumulators as in "
                                       "the field multiplier, 4 waves/SIMD, %.1f ms launch); clock and issue interval from tools/ubench_issue --peak" % d["ms"]},
                            **extra)
    except Exception:
        pass
    best = best or dict(PEAK_RECORDED)
    # what the SHIPPED lane-pair product reaches when a SIMD runs nothing else, at the occupancies the kernels run at: the
    # practical ceiling of `frac` for an Fq2 kernel; reported beside every roofline, never used as its denominator
    try:
        out = subprocess.run([os.path.join(ROOT, "tools", "ubench_product"), "--ceiling"], capture_output=True, text=True, timeout=120).stdout
        best["product_ceiling"] = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    except Exception:
        best["product_ceiling"] = None
    return best


def roofline(kernel, unit_key, ref_key, alg_key, t, units, kernel_ms, peak, traffic_key=None, extra=None, executed=None, useful=None):
    """frac = the kernel's OWN multiply-add count / time / peak (utilisation of the integer multiplier);
    algorithmic_speedup = reference-algorithm work / executed work (what the smarter algorithm buys)."""
    if not kernel_ms or kernel_ms < 1e-6:
        return None     # (test harness: nothing was timed)
    sec = kernel_ms * 1e-3
    per_unit = executed if executed is not None else EXECUTED_MACS[unit_key]
    useful_per_unit = useful if useful is not None else (USEFUL_MACS[unit_key] if executed is None else None)
    executed_total = per_unit * units
    ref = W_FQMUL[ref_key] * MAC_PER_FQMUL * units if ref_key else None
    ach = executed_total / sec / 1e12
    alg_bytes = ALG_BYTES[alg_key](t) * units
    prof = PROFILE.get(traffic_key or "", {})
    r = {"bound": "valu_int32_mac", "kernel": kernel, "kernel_ms": round(kernel_ms, 3), "units_per_launch": units,
         "achieved": round(ach, 3), "peak": peak["tmacs"], "unit": "TMAC/s", "frac": round(ach / peak["tmacs"], 4),
         "peak_clock_GHz": peak["clock_ghz"], "peak_source": peak["source"],
         "executed_macs_per_unit": per_unit,
         "achieved_is": "v_mad the kernel executes per unit (tests/count_ops.py) x units / HIP-event kernel time",
         "useful_macs_per_unit": useful_per_unit,
         "frac_useful": round(useful_per_unit * units / sec / 1e12 / peak["tmacs"], 4) if useful_per_unit else None,
         "frac_useful_is": "frac with the Fq-only work both lanes of a pair execute identically counted once (tests/count_ops.py --json-useful)",
         "algorithmic_bytes_per_launch": alg_bytes,
         "hbm_achieved_GBps": round(alg_bytes / sec / 1e9, 3), "hbm_peak_GBps": HBM_PEAK_GBPS,
         "hbm_frac": round(alg_bytes / sec / 1e9 / HBM_PEAK_GBPS, 6),
         "traffic": prof.get("traffic_bytes") if units == prof.get("units") else None,
         "traffic_is": ("L2<->fabric bytes per launch (2 x FETCH_SIZE + WRITE_SIZE), %s" % prof.get("source")) if prof else None}
    if r["traffic"]:
        r["traffic_over_algorithmic_bytes"] = round(r["traffic"] / alg_bytes, 1)
        r["traffic_GBps"] = round(r["traffic"] / sec / 1e9, 1)
        r["profile_kernel_ms"] = prof.get("profile_kernel_ms")   # the same launches under rocprofv3 --kernel-trace --stats
    if ref:
        r["reference_work_TMACs"] = round(ref / sec / 1e12, 3)
        r["algorithmic_speedup"] = round(ref / executed_total, 3)
    if prof.get("sq_insts_valu") and units == prof.get("units"):
        lane_instr = prof["sq_insts_valu"] * 64
        r["executed_cross_check"] = {"sq_insts_valu_wave_instr_per_launch": prof["sq_insts_valu"],
                                     "executed_v_mad_lane_instr_per_launch": executed_total,
                                     "implied_v_mad_share_of_valu": round(executed_total / lane_instr, 3),
                                     "static_v_mad_share_of_the_multiplier_bodies": PROFILE.get("static_mad_share")}
    if extra:
        r.update(extra)
    return r


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 20 (config 2), 6 (config 5)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=[2, 5])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--t", type=int, default=None)
    ap.add_argument("--signers", type=int, default=None)
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="--config 5 on ONE GPU: run the N rank slices of an N-rank job one after the other (8 x 131 072 = the BASELINE batch)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="contexts (one HIP stream each) of the `streaming` leg of config 2; 1 = skip that leg")
    ap.add_argument("--sustain-seconds", type=float, default=3.0, help="length of the `sustained` leg of config 2 (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary legs (general path, configs 3 and 4, PCIe-inclusive rate)")
    ap.add_argument("--profile-run", action="store_true",
                    help="for rocprofv3 captures of the whole line (r04's captures; since r05 one leg at a time: tools/capture_legs.sh, tools/profile_legs.py): skip the legs that launch the headline kernels on OTHER work "
                         "(general-path and checked-input combines), so that a kernel's per-launch averages describe one kind of launch")
    ap.add_argument("--latency-table", action="store_true",
                    help="instead of the bench line: call latency (host buffers in, results back) of sign / combine / verify / decrypt at "
                         "B = 1 ... 4096 next to ONE core of the CPU port -- the small-request table of INTEGRATION.md")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: CPU ranks, only with --test-engine")
    ap.add_argument("--test-engine", default=None, choices=[None, "hostsim"],
                    help="TEST HARNESS: tests/hostsim (g++ build of the device source) instead of the GPU; needs --backend gloo")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 20 if args.config == 2 else 6
    if (args.backend == "gloo") != (args.test_engine is not None):
        ap.error("--backend gloo and --test-engine go together (there is no CPU compute path in the product)")
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.emulate_world and (args.config != 5 or args.gpus != 1):
        ap.error("--emulate-world belongs to --config 5 on one GPU")
    return args


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """`bench.py --gpus N` started as a plain process: start N ranks (one per GPU) under torch.distributed.run and hand
    their output through.  Fails loudly when the node has fewer than N GPUs."""
    if args.test_engine is None:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but this node has %d visible GPU(s); refusing to report n_gpus=%d\n" % (args.gpus, have, args.gpus))
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["TC_BENCH_LAUNCHED_BY"] = "bench.py --gpus %d (self-spawn)" % args.gpus
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


ORACLE_BUILD = "gcc -O3 -march=x86-64-v3 (the library committed with the repository, oracle/Makefile)"


def oracle_for_this_host():
    """BASELINE.md 3 promises the CPU baseline `-O3 -march=native`.  The oracle library that travels with the repository is built
    where there is no GPU, for x86-64-v3, so the CPU legs rebuild the SAME source (oracle/c/tc_oracle.c) for THIS box's cores into a
    temporary directory and load that (TC_ORACLE_LIB, oracle/c_oracle.py); without gcc the committed build is timed and says so."""
    global ORACLE_BUILD
    import shutil
    import tempfile
    gcc = shutil.which("gcc")
    if not gcc or os.environ.get("TC_ORACLE_LIB"):
        return
    # a directory of this process's own (mkdtemp: mode 0700, unpredictable name -- ADVICE r05: a fixed path under /tmp can be raced by
    # a concurrent run and pre-created by another user); removed when the process exits
    import atexit
    build_dir = tempfile.mkdtemp(prefix="tc_oracle_native_")
    atexit.register(shutil.rmtree, build_dir, True)
    out = os.path.join(build_dir, "libtc_oracle_native.so")
    src = os.path.join(ROOT, "oracle", "c", "tc_oracle.c")
    try:
        subprocess.run([gcc, "-O3", "-march=native", "-fPIC", "-std=gnu11", "-fvisibility=hidden", "-shared", "-o", out, src, "-lpthread"],
                       check=True, timeout=300, capture_output=True)
    except (subprocess.SubprocessError, OSError):
        return
    os.environ["TC_ORACLE_LIB"] = out
    ORACLE_BUILD = "gcc -O3 -march=native, built on this box for its cores (BASELINE.md 3)"


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    # (the CPU legs run on rank 0 of a one-GPU run only: N ranks would rebuild and time N oracles against each other on one host)
    if not args.no_cpu_baseline and not args.test_engine and args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        oracle_for_this_host()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher and the flag disagree" % (args.gpus, world))
    harness = args.test_engine is not None
    if harness:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
            raise SystemExit("bench.py: rank %d needs GPU %d, this node shows %d (no CPU path exists)" % (rank, local_rank, torch.cuda.device_count() if torch.cuda.is_available() else 0))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    ranks_info = {"world_size": 1, "backend": None, "launched_by": os.environ.get("TC_BENCH_LAUNCHED_BY", "direct")}
    # started by a launcher (RANK/WORLD_SIZE in the environment): join its rendezvous -- also as the only rank, so that
    # `torch.distributed.run --nproc-per-node 1 bench.py` takes every collective of the N-rank run through RCCL
    if world > 1 or "WORLD_SIZE" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if harness:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        # every rank's device identity, gathered: N ranks on N DISTINCT GPUs, or the run is not an N-GPU run
        ident = "cpu:%d" % rank if harness else _device_identity(torch, local_rank)
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        if not harness and len(set(idents)) != world:
            raise SystemExit("bench.py: %d ranks but only %d distinct GPUs (%s)" % (world, len(set(idents)), idents))
        rccl = None
        if not harness:
            try:
                rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                rccl = "unknown"
        ranks_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": rccl, "devices": idents,
                      "launched_by": os.environ.get("TC_BENCH_LAUNCHED_BY", "torch.distributed.run (external launcher)")}

    if harness:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from hostsim_engine import HostSimEngine
        eng = HostSimEngine()
        peak = dict(PEAK_RECORDED)
    else:
        from threshold_crypto_amd.engine import Engine
        eng = Engine(local_rank)
        # every operand of the timed legs is an output of this library's own kernels, resident in HBM (hash points, share
        # signatures, ciphertexts made by the workload generator): known group members, so the per-operand membership
        # tests a context runs by default are switched off -- the reference's combine_signatures / verify do not repeat
        # from_bytes either.  `extras.combine_with_input_checks_per_s` is the same step with them on.
        eng.set_input_checks(False)
        peak = measure_peak() if rank == 0 else dict(PEAK_RECORDED)
    if args.latency_table:
        latency_table(eng)
        return None
    if args.config == 5:
        from threshold_crypto_amd import config5
        result = config5.run_bench(args, eng, dev, rank, world, peak, roofline, cpu_baseline=cpu_baseline_config5)
    else:
        result = run_config2(args, eng, dev, rank, world, peak)
    if rank == 0:
        result["n_gpus"] = ranks_info["world_size"]
        result["ranks"] = ranks_info
        if harness:
            result["test_harness"] = eng.version() + ": rank start-up / sharding / collectives exercised, NOT a measurement"
        emit(result)
    if ranks_info["backend"] is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return result


# ---- the line the driver parses ------------------------------------------------------------------------------------
# BENCH_r05.json: a 23.6 KB line came back `parsed: null` (the driver keeps 8 KB of stdout).  The LAST stdout line is therefore a
# COMPACT object (numbers and short names only, < LINE_LIMIT bytes whatever the world size); everything explanatory (the `*_is`
# prose, cross-checks, clocks, peak provenance, every secondary roofline in full) is the DETAIL object: written to
# bench_detail.json beside this script and printed FIRST, on stderr, behind the tag below: stdout carries the compact line and
# nothing else.
DETAIL_TAG = "bench_detail "
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")
LINE_LIMIT = 8000
ROOFLINE_KEEP = ("bound", "kernel", "kernel_ms", "units_per_launch", "achieved", "peak", "unit", "frac", "frac_useful", "executed_macs_per_unit",
                 "useful_macs_per_unit", "algorithmic_bytes_per_launch", "traffic", "profile_kernel_ms")
CPU_KEEP = ("value", "unit", "cores", "kind", "single_thread_per_s")


def _short(s, n=96):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def compact_roofline(r, extra=()):
    if not r:
        return None
    out = {k: r[k] for k in ROOFLINE_KEEP + tuple(extra) if k in r}
    out["kernel"] = _short(out.get("kernel", ""))
    return out


def compact_cpu(c):
    if not c:
        return None
    out = {k: c[k] for k in CPU_KEEP if k in c}
    if "sample" in c:
        out["sample"] = _short(c["sample"].split(" (")[0], 72)
    return out


def compact_leg(leg, kernel_ms=None):
    """one number each for a secondary BASELINE configuration: rate, kernel time, roofline fractions, the CPU rate beside it"""
    if not leg:
        return None
    r = leg.get("roofline") or {}
    out = {"value": leg.get("value"), "unit": leg.get("unit"), "ms_per_step": leg.get("ms_per_step"),
           "kernel_ms": kernel_ms if kernel_ms is not None else r.get("kernel_ms"),
           "frac": r.get("frac"), "frac_useful": r.get("frac_useful"), "traffic": r.get("traffic"),
           "cpu_baseline": compact_cpu(leg.get("cpu_baseline"))}
    return out


def compact(result):
    """The driver's line: exactly the contract's fields + a compact `roofline` and `cpu_baseline` + one object per secondary
    BASELINE configuration; never more than LINE_LIMIT bytes (the optional groups are dropped, last first, if a future field
    pushes it over)."""
    d = result
    cfg = d.get("config") or {}
    line = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")}
    line["dtype"] = _short(d.get("dtype", "").split(" (")[0] + " (Fq = 14 x 28-bit limbs, 64-bit column sums)", 64)
    line["data"] = d.get("data")
    line["config"] = {k: (_short(v, 160) if isinstance(v, str) else v) for k, v in cfg.items()
                      if k in ("workload", "t", "N", "batch_per_gpu", "parallelism", "fast_path", "steps_in_flight", "overlapped", "emulated_world")}
    if cfg.get("input_checks"):
        line["config"]["input_checks"] = cfg["input_checks"].split(":")[0]
    line["roofline"] = compact_roofline(d.get("roofline"), extra=("frac_timed_region", "frac_slowest_class", "frac_streaming", "frac_at_kernel_clock"))
    line["cpu_baseline"] = compact_cpu(d.get("cpu_baseline"))
    ranks = d.get("ranks") or {}
    devs = ranks.get("devices") or []
    line["ranks"] = {"world_size": ranks.get("world_size"), "backend": ranks.get("backend"), "rccl_version": ranks.get("rccl_version"),
                     "distinct_devices": len(set(devs)) if devs else None, "launched_by": _short(ranks.get("launched_by", ""), 48)}
    for k in ("verified_all", "test_harness", "valid_total_all_ranks"):
        if k in d:
            line[k] = d[k]
    optional = []       # dropped from the END first if the line would not fit
    if d.get("config3"):
        line["config3"] = compact_leg(d["config3"], d["config3"].get("kernel_ms"))
        line["config4"] = compact_leg(d.get("config4"), sum((d.get("config4") or {}).get("kernel_ms", {}).values()) or None) if d.get("config4") else None
        line["wire"] = compact_leg(d.get("wire"))
        optional += ["wire", "config4", "config3"]
    for k in ("streaming", "sustained", "general_path"):
        if d.get(k):
            line[k] = {"value": d[k].get("value"), "ms_per_step": d[k].get("ms_per_step")}
            optional.insert(0, k)
    if d.get("extras"):
        line["extras"] = {k: v for k, v in d["extras"].items() if isinstance(v, (int, float))}
        optional.insert(0, "extras")
    if d.get("secondary_rooflines"):
        line["secondary"] = {k: {"kernel_ms": r.get("kernel_ms"), "frac": r.get("frac"), "frac_useful": r.get("frac_useful"), "traffic": r.get("traffic")}
                             for k, r in d["secondary_rooflines"].items() if r}
        optional.insert(0, "secondary")
    for k in ("phase_kernel_ms", "combine_signatures_per_s", "share_signs_per_s", "pairing_verifies_per_s", "share_sign_kernel_ms"):
        if k in d:
            line[k] = d[k]
    recs = d.get("rank_records_start_jobs_valid_digest")
    if recs:          # [start, jobs, valid, sha3 digest of the rank's signatures]: 16 hex digits of the digest are enough to compare runs
        line["rank_records"] = [[r[0], r[1], r[2], str(r[3])[:16]] for r in recs]
        optional.insert(0, "rank_records")
    line["detail"] = "bench_detail.json (also the stderr line tagged '%s', written before this one)" % DETAIL_TAG.strip()
    text = json.dumps(line, separators=(",", ":"))
    while len(text) >= LINE_LIMIT - 500 and optional:
        line.pop(optional.pop(0), None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:
        raise AssertionError("bench line is %d bytes (limit %d)" % (len(text), LINE_LIMIT))
    return text


def emit(result):
    """detail -> bench_detail.json + a tagged line on STDERR, written first; then the compact line, the ONLY thing this script
    writes to stdout -- whichever end of stdout a harness keeps, and however little of it, the line survives whole.  (A caller that
    merges the two streams sees the tagged detail line EARLIER and the compact line last.)"""
    detail = json.dumps(result)
    try:
        with open(DETAIL_FILE, "w") as f:
            f.write(detail + "\n")
    except OSError:
        pass
    sys.stderr.write(DETAIL_TAG + detail + "\n")
    sys.stderr.flush()
    sys.stdout.write(compact(result) + "\n")
    sys.stdout.flush()


def _device_identity(torch, i):
    p = torch.cuda.get_device_properties(i)
    tag = getattr(p, "uuid", None)
    if tag is None:
        tag = "%s/%s/%s" % (getattr(p, "pci_domain_id", "?"), getattr(p, "pci_bus_id", "?"), getattr(p, "pci_device_id", "?"))
    return "%s gpu%d %s" % (p.name, i, tag)


def run_config2(args, eng, dev, rank, world, peak):
    import numpy as np
    import torch
    from threshold_crypto_amd.workload import ThresholdSigWorkload, ThresholdEncWorkload
    from threshold_crypto_amd.parallel import broadcast_key_set, shard_range, total_count, max_over_ranks, joined
    if joined(world):
        import torch.distributed as dist
    harness = getattr(eng, "is_test_harness", False)
    cuda = dev.type == "cuda"

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
        if cuda:
            torch.cuda.synchronize()
        if joined(world):
            dist.barrier()
            if cuda:
                torch.cuda.synchronize()

    def timed(fn, steps):
        """barrier + synchronize, `steps` calls of fn queued without host waits, barrier + synchronize; MAX over ranks"""
        sync()
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = fn()
        sync()
        return max_over_ranks(time.perf_counter() - t0, world, dev if cuda else None), out

    # ---- headline: combine_signatures, K steps on ONE context (one stream: no overlap between launches) -------------
    # The context runs on a torch-owned HIP stream so that events can bracket every step INSIDE the timed region without
    # a host wait: per-launch durations of exactly the launches that were timed (their sum cannot exceed the wall time).
    eng.set_timing(False)
    stream = None
    if cuda:
        stream = torch.cuda.Stream(device=dev)
        eng.set_stream(stream.cuda_stream)
    for _ in range(max(1, args.warmup)):
        sig, st = eng.combine_g2(t, d_idx, d_shares)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if cuda else []
    step_no = [0]

    def headline_step():
        i = step_no[0]
        step_no[0] += 1
        if cuda:
            ev[i][0].record(stream)
        o = eng.combine_g2(t, d_idx, d_shares)
        if cuda:
            ev[i][1].record(stream)
        return o

    dt, (sig, st) = timed(headline_step, args.steps)
    ms_per_step = dt / args.steps * 1e3
    value = B * world / (dt / args.steps)
    assert int(st.to(torch.int32).sum().item()) == 0, "combine reported per-job errors"
    kernel_ms = [a.elapsed_time(b) for a, b in ev] if cuda else [0.0]
    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
    assert avg_kernel_ms <= ms_per_step * 1.001 or not cuda, "per-launch event time exceeds the step time on one stream"

    # ---- streaming: the same step with `--in-flight` contexts (one HIP stream each): launches overlap at their ends --
    streaming = None
    engines = [eng]
    if not harness and args.in_flight > 1:
        from threshold_crypto_amd.engine import Engine
        engines = [eng] + [Engine(dev.index if dev.index is not None else 0) for _ in range(args.in_flight - 1)]
        for e in engines:
            e.set_input_checks(False)
            e.set_timing(False)
            e.combine_g2(t, d_idx, d_shares)
        for e in engines:
            e.sync()
        outs = []
        counter = [0]

        def step_rr():
            o = engines[counter[0] % len(engines)].combine_g2(t, d_idx, d_shares)
            counter[0] += 1
            outs.append(o)
            if len(outs) > 2 * len(engines):
                outs.pop(0)
            return o

        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_rr()
        for e in engines:
            e.sync()
        sync()
        sdt = max_over_ranks(time.perf_counter() - t0, world, dev)
        for o_sig, o_st in outs:
            assert bool((o_sig == sig).all().item()) and int(o_st.to(torch.int32).sum().item()) == 0, "steps in flight disagree with the one-stream step"
        del outs
        streaming = {"value": round(B * world / (sdt / args.steps), 1), "ms_per_step": round(sdt / args.steps * 1e3, 3), "steps": args.steps,
                     "contexts": len(engines), "overlapped": True,
                     "is": "the K steps alternate between %d contexts (one HIP stream each) and are not synchronised with the host "
                           "inside the timed region: the tail of one launch overlaps the head of the next (DESIGN.md 5.2); each "
                           "step is a full pass over the batch" % len(engines)}
    # ---- sustained: the one-context step repeated for >= --sustain-seconds ------------------------------------------
    sustained = None
    if not harness and args.sustain_seconds > 0:
        eng.set_timing(False)
        n = max(args.steps, int(args.sustain_seconds / (ms_per_step * 1e-3)) + 1)
        with ClockSampler(cuda and rank == 0) as cs:
            sdt, _ = timed(lambda: eng.combine_g2(t, d_idx, d_shares), n)
        sustained = {"value": round(B * world / (sdt / n), 1), "ms_per_step": round(sdt / n * 1e3, 3), "steps": n, "seconds": round(sdt, 3),
                     "is": "the headline step queued %d times on the one context (>= %.1f s of GPU time)" % (n, args.sustain_seconds),
                     "clock": cs.result()}
    eng.set_timing(True)

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
    # the same call repeated for ~1.5 s on the one context: the rate it sustains and the clock / power it runs at
    verify_sustained = None
    if not harness and args.sustain_seconds > 0:
        eng.set_timing(False)
        nv = max(4, int(min(args.sustain_seconds, 1.5) / max(verify_dt, 1e-4)) + 1)
        with ClockSampler(cuda and rank == 0) as cs:
            vdt, _ = timed(lambda: eng.verify_g2(master_pk, bad, d_hashes), nv)
        verify_sustained = {"value": round(B * world / (vdt / nv), 1), "ms_per_step": round(vdt / nv * 1e3, 3), "steps": nv, "clock": cs.result()}
        eng.set_timing(True)
    # the same check with calls in flight: 8 steps alternating between the contexts
    verify_if_dt = None
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
    n_valid = total_count(int(ok.to(torch.int64).sum().item()), world, dev if cuda else None)   # per-rank valid counts, summed over RCCL
    assert n_valid == int(expect_ok.sum().item()) * world
    _sh, _st = eng.g2_mul(d_sk, d_hashes)      # (untimed: the first call allocates the 50 MB result and the context's table arena)
    sync()
    del _sh, _st
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
    general = config4 = None
    # the CPU legs run on rank 0 of a one-GPU run only (N ranks would time N oracles against each other on one host)
    with_cpu = not (args.no_cpu_baseline or world > 1 or harness)
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
    if not args.no_extras and not harness:
        # ---- the same batch through the GENERAL path: share indices the small-index fast path does not take ---------
        # (signers OFFSET + i: abscissae above 65 535; Lagrange coefficients from k_lagrange, two-stage combination)
        OFFSET = 1 << 20
    if not args.no_extras and not harness and not args.profile_run:
        wg = ThresholdSigWorkload(eng, t, N, B, start=start, index_offset=OFFSET, hashes=wl.hashes)
        g_idx = torch.from_numpy(wg.idx.view(np.int64)).to(dev)
        g_shares = torch.from_numpy(wg.shares).to(dev)
        gsig, gst = eng.combine_g2(t, g_idx, g_shares)
        sync()
        g0 = time.perf_counter()
        gsig, gst = eng.combine_g2(t, g_idx, g_shares)
        general_kernel_ms = eng.last_kernel_ms()
        sync()
        g_dt = max_over_ranks(time.perf_counter() - g0, world, dev)
        assert int(gst.to(torch.int32).sum().item()) == 0 and bool((gsig == sig).all().item()), "general-path combination differs from the fast path's"
        general = {"value": round(B * world / g_dt, 1), "ms_per_step": round(g_dt * 1e3, 3),
                   "is": "the same messages and key set with signer indices %d + i (outside the small-index fast path): "
                         "k_lagrange + k_msm_tables + k_msm_ladder; result compared with the fast path's" % OFFSET,
                   "roofline": roofline("k_lagrange + k_msm_tables + k_msm_ladder", "combine_g2_t3_general", "combine_g2_t3", "combine_g2", t, B,
                                        general_kernel_ms, peak, traffic_key="general_path") if t == 3 else None}
        del wg, g_idx, g_shares, gsig
    if not args.no_extras and not harness:
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
        # which form ran: the context's own thresholds (tc_ctx_get_tuning; csrc/tc_launch.h kDuoMinHash / kDuoMinDecode unless the
        # environment said otherwise when the context was created)
        tuning = eng.tuning()
        hash_x2 = B >= tuning["duo_min_hash"]
        legs["hash_g2"] = roofline("k_hash_g2_x2" if hash_x2 else "k_hash_g2", "hash_g2_x2" if hash_x2 else "hash_g2", "hash_g2", "hash_g2", t, B,
                                   hash_kernel_ms, peak, traffic_key="hash_g2")
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
        cv_kernel_ms = eng.last_kernel_ms()
        sync()
        e1 = time.perf_counter()
        out, dst = eng.decrypt(t, didx, dsh, dv, doff)
        dec_kernel_ms = eng.last_kernel_ms()
        sync()
        e2 = time.perf_counter()
        extras["ciphertext_verify_kernel_ms"] = round(cv_kernel_ms, 3)
        extras["ciphertext_verifies_per_s"] = round(B * world / (e1 - e0), 1)
        extras["threshold_decrypts_per_s"] = round(B * world / (e2 - e1), 1)
        extras["threshold_decryptions_incl_ciphertext_verify_per_s"] = round(B * world / (e2 - e0), 1)
        assert int(okc.to(torch.int32).sum().item()) == B and int(dst.to(torch.int32).sum().item()) == 0
        assert bool((out.cpu() == torch.from_numpy(we.plain_flat)[: out.numel()]).all().item()), "threshold decryption returned wrong plaintext"
        legs["threshold_decrypt"] = roofline("k_combine_fast_g1_arena + k_xor_with_hash", "combine_g1_t3_fast", "combine_g1_t3",
                                             "combine_g1", t, B, dec_kernel_ms, peak, traffic_key="threshold_decrypt")
        # (the hashes take two messages per lane pair from 131 072 messages on: csrc/tc_launch.h kDuoMinHash)
        hkey = "hash_g1_g2_x2" if hash_x2 else "hash_g1_g2"
        cv_macs = EXECUTED_MACS["verify_g2"] + EXECUTED_MACS[hkey]
        cv_useful = USEFUL_MACS["verify_g2"] + USEFUL_MACS[hkey]
        legs["ciphertext_verify"] = roofline(("k_hash_g1_g2_x2" if hash_x2 else "k_hash_g1_g2") + " + k_miller_lines + k_miller_accumulate + k_final_exp", None, "ciphertext_verify", "ciphertext_verify", t, B,
                                             cv_kernel_ms, peak, traffic_key="ciphertext_verify", executed=cv_macs, useful=cv_useful)
        config4 = {"value": round(B * world / (e2 - e0), 1), "unit": "threshold_decryptions/s", "ms_per_step": round((e2 - e0) * 1e3, 3),
                   "is": "BASELINE config 4: Ciphertext::verify (hash_g1_g2 + pairing check) then PublicKeySet::decrypt (G1 combine + "
                         "keystream) over the batch; every plaintext compared with the workload's",
                   "kernel_ms": {"ciphertext_verify": round(cv_kernel_ms, 3), "decrypt": round(dec_kernel_ms, 3)},
                   "roofline": roofline(("k_hash_g1_g2_x2" if hash_x2 else "k_hash_g1_g2") + " + k_miller_lines + k_miller_accumulate + k_final_exp + k_combine_fast_g1_arena + k_xor_with_hash", None, None,
                                        "ciphertext_verify", t, B, cv_kernel_ms + dec_kernel_ms, peak,
                                        executed=cv_macs + EXECUTED_MACS["combine_g1_t3_fast"], useful=cv_useful + USEFUL_MACS["combine_g1_t3_fast"])}
        _cv, _td = legs["ciphertext_verify"], legs["threshold_decrypt"]
        if config4["roofline"] and _cv.get("traffic") and _td.get("traffic"):   # the step is the two legs one after the other
            config4["roofline"]["traffic"] = _cv["traffic"] + _td["traffic"]
            config4["roofline"]["traffic_is"] = "ciphertext_verify.traffic + threshold_decrypt.traffic (secondary_rooflines)"
            config4["roofline"]["traffic_over_algorithmic_bytes"] = round(config4["roofline"]["traffic"] / config4["roofline"]["algorithmic_bytes_per_launch"], 1)
        if with_cpu:
            import c_oracle as _co
            plain_gpu = out.cpu().numpy()

            def _c4(k, T):
                okv = _co.ciphertext_verify_batch(we.u[:k], we.v[:32 * k], 32, we.w[:k], T)
                pt, rc = _co.threshold_decrypt_batch(t, we.idx[:k], we.shares[:k], we.v[:32 * k], 32, T)
                return okv, pt, rc
            config4["cpu_baseline"] = cpu_leg(
                "threshold_decryptions/s", "config 4", "/root/reference/src/lib.rs:508-512 (Ciphertext::verify), :618-626 (PublicKeySet::decrypt)",
                min(B, CPU_LEG_JOBS), lambda k: _c4(k, 1), _c4,
                lambda g, k: int((g[0] != 1).sum()) + int(g[2].any()) + int((g[1].reshape(k, 32) != plain_gpu[:32 * k].reshape(k, 32)).any(axis=1).sum()))
    if not args.no_extras and not harness and not args.profile_run:
        # ---- the headline step with the context's default membership tests on every share -------------------------
        # (r06: the tests run on the context's second stream beside the combination, tc_api.hip Call::run_checks; best of three calls)
        eng.set_input_checks(True)
        csig, cst = eng.combine_g2(t, d_idx, d_shares)
        checked_dt = 1e9
        for _ in range(3):
            sync()
            c0 = time.perf_counter()
            csig, cst = eng.combine_g2(t, d_idx, d_shares)
            sync()
            checked_dt = min(checked_dt, time.perf_counter() - c0)
        extras["combine_with_input_checks_per_s"] = round(B * world / checked_dt, 1)
        eng.set_input_checks(False)
        assert bool((csig == sig).all().item()) and int(cst.to(torch.int32).sum().item()) == 0
    wire = None
    if not args.no_extras and not harness and not args.profile_run and cuda and t >= 1:
        # ---- the wire-level combine: 96-byte shares in, 96-byte signature out (checked decode + combine + to_bytes) ----------
        comp, cst2 = eng.g2_compress(d_shares.reshape(B * (t + 1), 192))
        d_wire = comp.reshape(B, t + 1, 96).contiguous()
        eng.set_timing(True)
        wsig, wst = eng.combine_signatures_wire(t, d_idx, d_wire)
        wire_ms = []
        w0 = None
        for _ in range(3):
            sync()
            w0 = time.perf_counter()
            wsig, wst = eng.combine_signatures_wire(t, d_idx, d_wire)
            sync()
            wire_ms.append((time.perf_counter() - w0, eng.last_kernel_ms()))
        eng.set_timing(False)
        want, _ = eng.g2_compress(sig)
        sync()
        assert int(wst.to(torch.int32).sum().item()) == 0 and bool((wsig == want).all().item()), "wire-level combine differs from compress(combine)"
        wall, kern = min(wire_ms)
        # (more than 32 768 decodes per call run two points per lane pair: csrc/tc_launch.h kDuoMinDecode, tc_duo.h)
        decode_x2 = B * (t + 1) >= eng.tuning()["duo_min_decode"]
        dkey = "g2_decompress_x2" if decode_x2 else "g2_decompress"
        wire_macs = (t + 1) * EXECUTED_MACS[dkey] + EXECUTED_MACS["combine_g2_t3_fast"]
        wire_useful = (t + 1) * USEFUL_MACS[dkey] + USEFUL_MACS["combine_g2_t3_fast"]
        wire = {"value": round(B * world / wall, 1), "unit": "combine_signatures/s", "ms_per_step": round(wall * 1e3, 3),
                "is": "tc_combine_signatures_wire_batch on the BASELINE batch: %d compressed shares per job through the checked decode of from_bytes "
                      "(square root + membership test each), combined, returned as Signature::to_bytes; result compared with "
                      "compress(combine) of the timed batch" % (t + 1),
                "algorithmic_bytes_per_job": (t + 1) * (96 + 8) + 96,
                "roofline": roofline(("k_decompress_take_g2_x2" if decode_x2 else "k_decompress_take<Fq2>") + " + k_combine_fast<Fq2> + k_compress<Fq2>", None, None, "combine_g2_wire", t, B, kern, peak, traffic_key="wire",
                                     executed=wire_macs, useful=wire_useful) if t == 3 else None}
        extras["wire_combine_per_s"] = wire["value"]
        if with_cpu:
            import c_oracle as _co
            wire_np, wsig_np = d_wire.cpu().numpy(), wsig.cpu().numpy()

            def _wire(k, T):
                return _co.combine_signatures_wire_batch(t, wl.idx[:k], wire_np[:k], T)
            wire["cpu_baseline"] = cpu_leg(
                "combine_signatures/s", "wire-level combine", "/root/reference/src/lib.rs:246-252 (from_bytes), :608-615 (combine_signatures), :255-259 (to_bytes)",
                min(B, CPU_LEG_JOBS), lambda k: _wire(k, 1), _wire,
                lambda g, k: int(g[1].any()) + int((g[0] != wsig_np[:k]).any(axis=1).sum()))
    if not args.no_extras and not harness and not args.profile_run:
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
    fast = 1 <= t <= 3
    unit_key = "combine_g2_t3_fast" if t == 3 else None
    head = None
    if unit_key and not harness:
        head = roofline("k_lagrange + 3 grouping kernels + k_combine_fast<Fq2> + two-stage kernels (idle)", unit_key, "combine_g2_t3",
                        "combine_g2", t, B, avg_kernel_ms, peak, traffic_key="combine_g2_t3",
                        extra={"frac_slowest_class": round(EXECUTED_MACS["combine_g2_t3_fast_general_denominator"] * B
                                                           / (avg_kernel_ms * 1e-3) / 1e12 / peak["tmacs"], 4),
                               "frac_timed_region": round(EXECUTED_MACS[unit_key] * B / (ms_per_step * 1e-3) / 1e12 / peak["tmacs"], 4),
                               "frac_timed_region_is": "executed multiply-adds of the K timed steps / their wall time / peak (one context); "
                                                       "with steps in flight: frac_streaming",
                               "frac_streaming": round(EXECUTED_MACS[unit_key] * B / (streaming["ms_per_step"] * 1e-3) / 1e12 / peak["tmacs"], 4) if streaming else None,
                               "frac_is": "average job over the 4-of-10 subsets; the 65 536-job batch is ONE resident round "
                                          "of 2048 waves and lasts as long as its slowest denominator class, whose waves run "
                                          "at frac_slowest_class (DESIGN.md 5.2)"})
    if not harness:
        legs["pairing_check"] = roofline("k_miller_lines + k_miller_accumulate + k_final_exp (above 16 384 checks; k_pairing_quad below)", "verify_g2_prepared", "verify_g2", "verify_g2", t, B, verify_kernel_ms, peak,
                                         traffic_key="pairing_check")
        legs["g2_sign"] = roofline("k_g2_mul_shared", "g2_mul_4_scalars_per_point", "g2_mul", "g2_mul", t, (t + 1) * B, sign_kernel_ms, peak, traffic_key="g2_sign")
    # the CPU leg runs on rank 0 of a one-GPU run only (N ranks would time N oracles against each other on one host)
    cpu = None if (args.no_cpu_baseline or world > 1 or harness) else cpu_baseline(wl, sig.cpu().numpy(), t, args.cpu_seconds)
    config3 = {"value": round(B * world / verify_dt, 1), "unit": "pairing_verifies/s", "ms_per_step": round(verify_dt * 1e3, 3),
               "kernel_ms": round(verify_kernel_ms, 3),
               "is": "BASELINE config 3: PublicKey::verify_g2 over the batch, one call, host waits; every 16th signature replaced, ok-vector checked",
               "streaming": ({"value": round(B * world / verify_if_dt, 1), "ms_per_step": round(verify_if_dt * 1e3, 3), "overlapped": True}
                             if verify_if_dt else None),
               "sustained": verify_sustained,
               "roofline": legs.get("pairing_check")}
    if with_cpu:
        import c_oracle as _co
        bad_np, hashes_np, ok_np, pk_np = bad.cpu().numpy(), d_hashes.cpu().numpy(), ok.cpu().numpy(), bytes(master_pk.cpu().numpy())
        config3["cpu_baseline"] = cpu_leg(
            "pairing_verifies/s", "config 3", "/root/reference/src/lib.rs:108-117 (PublicKey::verify_g2: two pairings)", min(B, CPU_LEG_JOBS),
            lambda k: _co.verify_g2_batch(pk_np, bad_np[:k], hashes_np[:k], 1), lambda k, T: _co.verify_g2_batch(pk_np, bad_np[:k], hashes_np[:k], T),
            lambda g, k: int((g.astype(np.uint8) != ok_np[:k]).sum()))
    # what is left of each leg's frac once the clock it ran at is taken out (the denominator's loop runs ~5-10 % faster)
    for leg, clock in ((head, sustained and sustained.get("clock")), (config3["roofline"], verify_sustained and verify_sustained.get("clock"))):
        if leg and clock and clock.get("sclk_GHz") and leg.get("peak_clock_GHz"):
            leg["kernel_clock"] = clock
            leg["frac_at_kernel_clock"] = round(leg["frac"] * leg["peak_clock_GHz"] / clock["sclk_GHz"], 4)
    return {
        "metric": "combine_signatures/sec", "value": round(value, 1), "unit": "combine_signatures/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i32 limbs (Fq = 14 x 28-bit signed, Montgomery R=2^392; 64-bit column accumulators; one Fq2 coefficient per lane of a lane pair)",
        "data": "synthetic",
        "config": {"workload": "t=%d,N=%d,batch=%d threshold signatures (combine_signatures, G2), per GPU" % (t, N, B),
                   "t": t, "N": N, "batch_per_gpu": B, "parallelism": "jobs sharded, dp%d" % world,
                   "fast_path": fast, "steps_in_flight": 1, "overlapped": False,
                   "input_checks": "off: every operand is an output of the library's own kernels (known members); "
                                   "extras.combine_with_input_checks_per_s has them on",
                   "timed_region_is": "K steps queued on ONE context (one HIP stream: launches do not overlap), barrier + "
                                      "synchronize on both sides, MAX over ranks",
                   "comparable_with": "round 2's BENCH value (17.6 M) was the two-contexts figure: compare it with streaming.value; "
                                      "round 2's sequential.value (13.3 M) is what `value` is now"},
        "streaming": streaming,
        "sustained": sustained,
        "general_path": general,
        "config3": config3,
        "config4": config4,
        "wire": wire,
        "pairing_verifies_per_s": round(B * world / min(verify_dt, verify_if_dt or verify_dt), 1),
        "pairing_verifies_sequential_per_s": round(B * world / verify_dt, 1),
        "pairing_verify_kernel_ms": round(verify_kernel_ms, 3),
        "pairing_verify_valid_count_all_ranks": n_valid,
        "share_signs_per_s": round((t + 1) * B * world / sign_dt, 1),
        "share_sign_kernel_ms": round(sign_kernel_ms, 3),
        "verified_all": True,
        "extras": extras,
        "roofline": head,
        "product_ceiling": peak.get("product_ceiling"),
        "secondary_rooflines": legs,
        "cpu_baseline": cpu,
    }


def latency_table(eng):
    """Small requests.  The reference's API is single-item (SecretKeyShare::sign src/lib.rs:447, PublicKeySet::combine_signatures
    :608, PublicKey::verify :115, PublicKeySet::decrypt :618); a GPU call costs the latency of ONE wave's job however few jobs it
    carries.  Per entry and batch size B: wall time of one call with HOST buffers (staging copies included; median of 7 after a
    warm-up), the kernel time inside it, the resulting jobs per second -- next to ONE host core of Oracle B on the same jobs (the
    cpu_baseline leg: `kind: port`), and the smallest B from which the call beats that core.  One JSON line per entry."""
    import numpy as np
    from threshold_crypto_amd.engine import pack_messages
    from threshold_crypto_amd.workload import ThresholdSigWorkload, ThresholdEncWorkload
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    c_oracle.load()
    t, N, BMAX = 3, 10, 4096
    sizes = [1, 8, 64, 1024, 4096]
    eng.set_timing(True)
    eng.set_input_checks(True)          # the context's default: what a caller that hands over received shares gets
    wl = ThresholdSigWorkload(eng, t, N, BMAX)
    we = ThresholdEncWorkload(eng, t, N, BMAX)
    sig, st = eng.combine_g2(t, wl.idx, wl.shares)
    sk = wl.shares_sk[0]._bytes()
    fr = np.frombuffer(sk, dtype=np.uint8)[None].copy()

    def med(fn, reps=7):
        fn()
        wall, kern = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            wall.append(time.perf_counter() - t0)
            kern.append(eng.last_kernel_ms())
        return sorted(wall)[reps // 2] * 1e3, sorted(kern)[reps // 2]

    def cpu_time(fn, n):
        t0 = time.perf_counter()
        for j in range(n):
            fn(j)
        return (time.perf_counter() - t0) / n * 1e3

    def msgs(B):
        return pack_messages(wl.msgs[:B])

    entries = {
        "sign (SecretKeyShare::sign: hash_g2 + [sk] H)": (
            lambda B: (lambda m=msgs(B): eng.sign(fr, m[0], m[1])),
            lambda j: c_oracle.sign(sk, wl.msgs[j])),
        "combine_signatures (t = 3, membership tests on the 4 shares: the context's default)": (
            lambda B: (lambda: eng.combine_g2(t, wl.idx[:B].copy(), wl.shares[:B].copy())),
            lambda j: c_oracle.combine_g2(t, [int(i) for i in wl.idx[j]], [bytes(x) for x in wl.shares[j]])),
        "verify (PublicKey::verify: hash_g2 + pairing check, membership test on the signature)": (
            lambda B: (lambda m=msgs(B): eng.verify_sig(wl.master_pk, sig[:B].copy(), m[0], m[1])),
            lambda j: c_oracle.verify(bytes(wl.master_pk), bytes(sig[j]), wl.msgs[j])),
        "decrypt (Ciphertext::verify + PublicKeySet::decrypt, t = 3)": (
            lambda B: (lambda: (eng.ciphertext_verify(we.u[:B].copy(), we.v[:32 * B].copy(), we.off[:B + 1].copy(), we.w[:B].copy()),
                                eng.decrypt(t, we.idx[:B].copy(), we.shares[:B].copy(), we.v[:32 * B].copy(), we.off[:B + 1].copy()))),
            lambda j: (c_oracle.ciphertext_verify(bytes(we.u[j]), bytes(we.v[32 * j:32 * j + 32]), bytes(we.w[j])),
                       c_oracle.threshold_decrypt(t, [int(i) for i in we.idx[j]], [bytes(x) for x in we.shares[j]], bytes(we.v[32 * j:32 * j + 32])))),
    }
    # what one PublicKey::verify at B = 1 ... 64 is made of (could a check spread over a whole wave reach one CPU
    # core's 3.8 ms?): the three device phases alone, kernel time.  hash_g2 is a chain of Fq2 operations on ONE lane pair (two square-
    # root exponentiations, the cofactor ladder): no spreading of the PAIRING over more lanes shortens it.
    for B in (1, 8, 64):
        m = msgs(B)
        ph = {"hash_g2": med(lambda: eng.hash_g2(m[0], m[1]))[1],
              "signature_membership_test": med(lambda: eng.g2_subgroup_check(sig[:B].copy()))[1]}
        eng.set_input_checks(False)
        ph["pairing_check_four_lanes_per_check"] = med(lambda: eng.verify_g2(wl.master_pk, sig[:B].copy(), wl.hashes[:B].copy()))[1]
        eng.set_input_checks(True)
        print(json.dumps({"entry": "verify phases (kernel ms, each phase alone)", "B": B, **{k: round(v, 3) for k, v in ph.items()},
                          "sum_ms": round(sum(ph.values()), 3)}), flush=True)
    for name, (gpu, cpu) in entries.items():
        cpu_ms = cpu_time(cpu, 6)
        rows, crossover = [], None
        for B in sizes:
            wall_ms, kern_ms = med(gpu(B))
            rows.append({"B": B, "call_ms": round(wall_ms, 3), "kernel_ms": round(kern_ms, 3), "jobs_per_s": round(B / (wall_ms * 1e-3), 1)})
            if crossover is None and wall_ms < B * cpu_ms:
                crossover = B
        # refine the crossover between the last losing and the first winning size
        if crossover and crossover > 1:
            lo = sizes[sizes.index(crossover) - 1]
            for B in range(lo + 1, crossover + 1):
                if med(gpu(B), 3)[0] < B * cpu_ms:
                    crossover = B
                    break
        print(json.dumps({"entry": name, "one_cpu_core_ms_per_job": round(cpu_ms, 3), "cpu_is": "Oracle B (oracle/c/tc_oracle.c), one host core",
                          "gpu_rows": rows, "gpu_call_beats_one_core_from_B": crossover}), flush=True)


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
            "build": ORACLE_BUILD,
            "sample": "first %d jobs of the timed batch on %d pthreads (oracle/c/tc_oracle.c); "
                      "every sampled job compared bit-exact with the GPU output" % (n, threads),
            "single_thread_per_s": round(1.0 / per, 2)}


def cpu_leg(unit, what, ref, n, single, threaded, check):
    """Oracle B beside a secondary leg of the bench line (BASELINE's metric names the pairing rate too, and
    north_star asks for the CPU path timed beside each): `single(k)` runs the first k jobs on one thread, `threaded(k, T)` on T
    pthreads and returns what `check` compares with the GPU's output of the timed batch.  `ref` = the reference lines the leg
    restates.  kind = port (the reference is Rust with un-vendored crates: no oracle/_ref in this image)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    c_oracle.load()
    threads = c_oracle.host_threads()
    n1 = 4
    t0 = time.perf_counter()
    single(n1)
    per = max((time.perf_counter() - t0) / n1, 1e-5)
    t0 = time.perf_counter()
    got = threaded(n, threads)
    dt = time.perf_counter() - t0
    bad = check(got, n)
    if bad:
        raise AssertionError("%s: GPU output differs from the CPU oracle on %d of %d sampled jobs" % (what, bad, n))
    return {"value": round(n / dt, 2), "unit": unit, "cores": threads, "kind": "port", "reference": ref,
            "single_thread_per_s": round(1.0 / per, 2), "thread_scaling": round((n / dt) * per, 2), "build": ORACLE_BUILD,
            "sample": "first %d jobs of the timed batch on %d pthreads (oracle/c/tc_oracle.c); every sampled job compared with the GPU output"
                      % (n, threads)}


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
