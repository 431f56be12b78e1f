"""Does the chip hold its clock under the real kernels?  bench.py's roofline denominator is measured in a multiply-add-only loop
(2.35 GHz); if the pairing / combine kernels run at a lower shader clock (power management), `frac` charges the kernels for it.
Samples `rocm-smi` (sclk, power) every ~0.2 s while (a) the peak microbenchmark loops, (b) verify_g2 loops on a resident batch,
(c) combine_signatures loops.  python tools/clock_probe.py [seconds per leg]"""
import json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True).stdout
    try:
        d = json.loads(out)
        card = d[sorted(d)[0]]
        sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
        pw = next((v for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), None)
        m = re.search(r"(\d+)\s*[Mm][Hh]z", str(sclk))
        return (int(m.group(1)) if m else None, float(pw) if pw not in (None, "N/A") else None)
    except Exception:
        return (None, None)


def watch(stop, rows):
    while not stop.is_set():
        rows.append(sample())
        time.sleep(0.15)


def leg(name, work):
    stop, rows = threading.Event(), []
    th = threading.Thread(target=watch, args=(stop, rows))
    th.start()
    t0 = time.time()
    info = work(t0 + SECS)
    stop.set()
    th.join()
    clk = sorted(c for c, _ in rows[2:] if c)
    pw = sorted(p for _, p in rows[2:] if p)
    print(json.dumps({"leg": name, "samples": len(clk), "sclk_MHz_min_median_max": [clk[0], clk[len(clk) // 2], clk[-1]] if clk else None,
                      "power_W_min_median_max": [pw[0], pw[len(pw) // 2], pw[-1]] if pw else None, "info": info}), flush=True)


def peak_loop(t_end):
    last = None
    while time.time() < t_end:
        out = subprocess.run([os.path.join(ROOT, "tools", "ubench_issue"), "--peak"], capture_output=True, text=True).stdout.strip().splitlines()
        if out:
            last = json.loads(out[-1])
    return {"in_kernel_clock_GHz": last and last.get("clock_GHz"), "T_lane_ops_per_s": last and last.get("T_lane_ops_per_s")}


def main():
    print(json.dumps({"idle": sample()}), flush=True)
    leg("peak microbenchmark (v_mad_i64_i32 only)", peak_loop)
    import numpy as np
    import torch
    from threshold_crypto_amd.engine import Engine
    from threshold_crypto_amd.workload import ThresholdSigWorkload
    e = Engine(0)
    e.set_input_checks(False)
    B = 65536
    wl = ThresholdSigWorkload(e, 3, 10, B)
    sig, st = e.combine_g2(3, wl.idx, wl.shares)
    d = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (wl.master_pk, sig, wl.hashes, wl.idx, wl.shares)]

    def verify_loop(t_end):
        n, t0 = 0, time.time()
        while time.time() < t_end:
            for _ in range(8):
                ok = e.verify_g2(d[0], d[1], d[2])
            e.sync()
            n += 8
        return {"calls": n, "ms_per_call": round((time.time() - t0) / n * 1e3, 3)}

    def combine_loop(t_end):
        n, t0 = 0, time.time()
        while time.time() < t_end:
            for _ in range(16):
                out = e.combine_g2(3, d[3], d[4])
            e.sync()
            n += 16
        return {"calls": n, "ms_per_call": round((time.time() - t0) / n * 1e3, 3)}

    leg("verify_g2, 65 536 checks per call, device-resident, back to back", verify_loop)
    leg("combine_signatures, 65 536 jobs per call, device-resident, back to back", combine_loop)
    leg("peak microbenchmark again", peak_loop)


main()
