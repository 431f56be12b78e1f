"""Builds libtc_amd.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m threshold_crypto_amd.build [--force] [--jobs N]

One translation unit per kernel family, compiled in parallel; objects are cached under
threshold_crypto_amd/_build/ keyed by a hash of the sources, so rebuilding is incremental.
hipcc cross-compiles for gfx950 without a GPU.
"""
import argparse
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
# experiments: TC_BUILD_FLAGS="-DTC_INLINE_JAC" TC_BUILD_SUFFIX="_inl" builds libtc_amd_inl.so
EXTRA_FLAGS = os.environ.get("TC_BUILD_FLAGS", "").split()
SUFFIX = os.environ.get("TC_BUILD_SUFFIX", "")
LIB = os.path.join(HERE, "libtc_amd%s.so" % SUFFIX)
UNITS = ["tc_api", "tc_group", "k_mul", "k_combine", "k_pairing", "k_hash", "k_check", "k_dkg", "k_msm", "k_comb"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-fno-gpu-rdc", "-Wno-unused-result"] + EXTRA_FLAGS


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libtc_amd.so)")


def _headers_digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith(".h"):
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    with open(os.path.join(HERE, "..", "include", "tc_amd.h"), "rb") as f:
        h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h


def _unit_key(unit, base):
    h = base.copy()
    with open(os.path.join(CSRC, unit + ".hip"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:16]


def _compile(unit, key, timeout):
    obj = os.path.join(BUILD, "%s-%s.o" % (unit, key))
    if os.path.exists(obj):
        return unit, obj, 0.0, True
    t0 = time.time()
    cmd = [_hipcc()] + FLAGS + ["-c", os.path.join(CSRC, unit + ".hip"), "-o", obj + ".tmp"]
    subprocess.run(cmd, check=True, timeout=timeout)
    os.replace(obj + ".tmp", obj)
    return unit, obj, time.time() - t0, False


def build(force=False, jobs=None, timeout=1500, verbose=True):
    os.makedirs(BUILD, exist_ok=True)
    base = _headers_digest()
    keys = {u: _unit_key(u, base) for u in UNITS}
    stamp = os.path.join(BUILD, "lib%s.stamp" % SUFFIX)
    want = " ".join(keys[u] for u in UNITS)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == want:
        return LIB
    objs = {}
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or len(UNITS)) as ex:
        futs = [ex.submit(_compile, u, keys[u], timeout) for u in UNITS]
        for f in concurrent.futures.as_completed(futs):
            unit, obj, dt, cached = f.result()
            objs[unit] = obj
            if verbose:
                print("[tc build] %-10s %s" % (unit, "cached" if cached else "%.1fs" % dt), flush=True)
    cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB + ".tmp"] + [objs[u] for u in UNITS] + ["-ldl", "-lpthread"]
    subprocess.run(cmd, check=True, timeout=600)
    os.replace(LIB + ".tmp", LIB)
    with open(stamp, "w") as f:
        f.write(want)
    # drop stale objects (default build only; experiment builds share the directory and simply
    # recompile): the directory travels to the GPU box with every gpurun snapshot
    if not SUFFIX:
        keep = {os.path.basename(o) for o in objs.values()}
        for name in os.listdir(BUILD):
            if name.endswith(".o") and name not in keep:
                os.remove(os.path.join(BUILD, name))
    if verbose:
        print("[tc build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--timeout", type=int, default=1500)
    a = ap.parse_args()
    try:
        build(force=a.force, jobs=a.jobs, timeout=a.timeout)
    except (subprocess.CalledProcessError, subprocess.TimeoutExpired, RuntimeError) as e:
        print("[tc build] FAILED:", e, file=sys.stderr)
        sys.exit(1)
