#!/usr/bin/env python3
"""tests/golden/config5_world8_reduced.json: the per-rank records (start, jobs, valid, digest of the rank's signatures) of
BASELINE config 5's flow on EIGHT gloo ranks at a reduced size (t = 8, N = 12, 2 jobs per rank), kernels = the host build
of the device source (tests/hostsim).  The GPU test replays the same eight slices through HIP on one GPU
(config5.run_emulated_world) and must reproduce every digest.

    python tools/gen_config5_digests.py            (→ file)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = {"t": 8, "N": 12, "batch_per_rank": 2, "world": 8}


def main():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(SHAPE["world"]), "--config", "5", "--backend", "gloo", "--test-engine",
           "hostsim", "--batch", str(SHAPE["batch_per_rank"]), "--t", str(SHAPE["t"]), "--signers", str(SHAPE["N"]), "--steps", "1",
           "--warmup", "0", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")][0]
    assert line["n_gpus"] == SHAPE["world"] and line["valid_total_all_ranks"] == SHAPE["world"] * SHAPE["batch_per_rank"]
    doc = dict(SHAPE, source="bench.py --gpus 8 --config 5 --backend gloo --test-engine hostsim (8 gloo ranks, host build of the device source)",
               records=line["rank_records_start_jobs_valid_digest"])
    path = os.path.join(ROOT, "tests", "golden", "config5_world8_reduced.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
