"""The reference's own tests (src/lib.rs:793-1008, doc-test :583-607, examples/threshold_sig.rs)
replayed through the host-side API mirror (threshold_crypto_amd/api.py) on the GPU, with the
oracle as the independent checker.  Names and flow follow the Rust tests line by line."""
import os
import random

import numpy as np
import pytest

import tc_oracle as o
from threshold_crypto_amd import api
from threshold_crypto_amd.api import (Ciphertext, DecryptionShare, NotEnoughShares, PublicKeySet, SecretKey, SecretKeySet,
                                      SignatureShare)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _engine(engine):
    api.set_default_engine(engine)
    yield


@pytest.fixture(scope="module")
def rnd():
    return random.Random(4242)


def random_sk_set(t, rnd):
    return SecretKeySet([rnd.randrange(o.R) for _ in range(t + 1)])


def test_doc_test_combine_signatures(rnd):
    """src/lib.rs:583-607."""
    sk_set = random_sk_set(3, rnd)
    sk_shares = [sk_set.secret_key_share(i) for i in range(6)]
    pk_set = sk_set.public_keys()
    msg = b"Happy birthday! If this is signed, at least four people remembered!"
    sig_shares = {i: sk_shares[i].sign(msg) for i in range(4)}
    for i, sig_share in sig_shares.items():
        assert pk_set.public_key_share(i).verify(sig_share, msg)
    sig = pk_set.combine_signatures(sig_shares)
    assert pk_set.public_key().verify(sig, msg)
    # oracle: the combination is the master key's signature
    assert sig.raw == o.g2_uncompressed(o.sign(sk_set.poly[0], msg))


def test_threshold_sig(rnd):
    """test_threshold_sig, src/lib.rs:822-873."""
    sk_set = random_sk_set(3, rnd)
    pk_set = sk_set.public_keys()
    pk_master = pk_set.public_key()
    assert pk_master.raw != pk_set.public_key_share(0).raw and pk_master.raw != pk_set.public_key_share(2).raw
    assert pk_set.threshold() == 3
    msg = b"Totally real news"
    sigs = {i: sk_set.secret_key_share(i).sign(msg) for i in (5, 8, 7, 10)}
    for i, sig in sigs.items():
        assert pk_set.public_key_share(i).verify(sig, msg)
    sig = pk_set.combine_signatures(sigs)
    assert pk_set.public_key().verify(sig, msg)
    sigs2 = {i: sk_set.secret_key_share(i).sign(msg) for i in (42, 43, 44, 45)}
    sig2 = pk_set.combine_signatures(sigs2)
    assert sig == sig2
    with pytest.raises(NotEnoughShares):
        pk_set.combine_signatures({i: sigs[i] for i in (5, 8, 7)})
    # more than t+1 shares: the first t+1 in BTreeMap order are used (take(t+1), src/lib.rs:728)
    more = dict(sigs)
    more[11] = sk_set.secret_key_share(11).sign(msg)
    assert pk_set.combine_signatures(more) == sig


def test_simple_sig(rnd):
    """test_simple_sig, src/lib.rs:810-820."""
    sk0, sk1 = SecretKey(rnd.randrange(o.R)), SecretKey(rnd.randrange(o.R))
    pk0, pk1 = sk0.public_key(), sk1.public_key()
    msg0, msg1 = b"Real news", b"Fake news"
    assert pk0.verify(sk0.sign(msg0), msg0)
    assert not pk1.verify(sk0.sign(msg0), msg0)
    assert not pk0.verify(sk0.sign(msg0), msg1)
    assert pk0.raw == o.g1_uncompressed(o.public_key(sk0.fr))


def test_simple_enc_and_threshold_enc(rnd):
    """test_simple_enc (src/lib.rs:875-897) and test_threshold_enc (:907-939); ciphertexts are built
    by the oracle's encrypt (encryption is a 'next' row), decrypted on the GPU."""
    sk_bob = SecretKey(rnd.randrange(o.R))
    pk_bob = o.public_key(sk_bob.fr)
    msg = b"Muffins in the canteen today! Don't tell Eve!"
    u, v, w = o.encrypt_with_r(pk_bob, rnd.randrange(1, o.R), msg)
    ct = Ciphertext(o.g1_uncompressed(u), v, o.g2_uncompressed(w))
    assert ct.verify()
    assert sk_bob.decrypt(ct) == msg
    assert SecretKey(rnd.randrange(o.R)).decrypt(ct) != msg
    fake = Ciphertext(ct.u, bytes([ct.v[0] ^ 1]) + ct.v[1:], ct.w)
    assert not fake.verify() and sk_bob.decrypt(fake) is None
    # threshold
    sk_set = random_sk_set(3, rnd)
    pk_set = sk_set.public_keys()
    u, v, w = o.encrypt_with_r(o.g1_from_uncompressed(pk_set.public_key().raw, check=False), rnd.randrange(1, o.R), b"Totally real news")
    ct = Ciphertext(o.g1_uncompressed(u), v, o.g2_uncompressed(w))
    assert ct.verify()
    shares = {}
    for i in (8, 4, 7, 9):
        sh = sk_set.secret_key_share(i).decrypt_share(ct)
        assert isinstance(sh, DecryptionShare)
        assert pk_set.public_key_share(i).verify_decryption_share(sh, ct)
        shares[i] = sh
    assert pk_set.decrypt(shares, ct) == b"Totally real news"
    bad = dict(shares)
    bad[8] = shares[4]
    assert not pk_set.public_key_share(8).verify_decryption_share(bad[8], ct)
    with pytest.raises(NotEnoughShares):
        pk_set.decrypt({i: shares[i] for i in (8, 4, 7)}, ct)
    fake = Ciphertext(ct.u, bytes([ct.v[0] ^ 1]) + ct.v[1:], ct.w)
    assert sk_set.secret_key_share(2).decrypt_share(fake) is None


def test_hash_g2_properties_and_oracle(rnd):
    """test_hash_g2 (src/lib.rs:941-952): deterministic, message-sensitive; plus oracle equality."""
    msg = bytes(rnd.randrange(256) for _ in range(1000))
    msg_end0 = msg + b"\x00"
    msg_end1 = msg + b"\x01"
    h = api.hash_g2_batch([msg, msg, msg_end0, msg_end1])
    assert h[0] == h[1] and h[0] != h[2] and h[2] != h[3]
    assert h[0] == o.g2_uncompressed(o.hash_g2(msg))


def test_from_to_bytes_sizes(rnd):
    """test_from_to_bytes / test_size (src/lib.rs:984-993,1049-1053): 48 / 96 byte encodings."""
    sk = SecretKey(rnd.randrange(o.R))
    sig = sk.sign(b"Please sign here: ______")
    pk = sk.public_key()
    assert len(pk.to_bytes()) == api.PK_SIZE == 48 and len(sig.to_bytes()) == api.SIG_SIZE == 96
    assert pk.to_bytes() == o.g1_compressed(o.public_key(sk.fr))
    assert o.g2_from_compressed(sig.to_bytes()) == o.sign(sk.fr, b"Please sign here: ______")
    assert sig.parity() == o.signature_parity(o.sign(sk.fr, b"Please sign here: ______"))


def test_share_validation_loop_of_threshold_sig_example(rnd):
    """examples/threshold_sig.rs:115-131: validate every share under its public key share, drop
    the invalid ones, combine the rest."""
    t, n = 2, 6
    sk_set = random_sk_set(t, rnd)
    pk_set = sk_set.public_keys()
    msg = b"a block"
    shares = {i: sk_set.secret_key_share(i).sign(msg) for i in range(n)}
    shares[3] = SignatureShare(sk_set.secret_key_share(4).sign(msg).raw)  # node 3 lies
    pk_shares = pk_set.public_key_shares(list(range(n)))
    ok = api.PublicKeyShare.verify_batch_shares(pk_shares, [shares[i] for i in range(n)], [msg] * n)
    assert ok.tolist() == [True, True, True, False, True, True]
    good = {i: shares[i] for i in range(n) if ok[i]}
    assert pk_set.public_key().verify(pk_set.combine_signatures(good), msg)
    assert [p.raw for p in pk_shares] == [o.g1_uncompressed(o.public_key(sk_set.secret_key_share(i).fr)) for i in range(n)]


def test_opt_in_rlc_paths_through_the_api_mirror(rnd):
    """The two round-3 random-linear-combination paths from the reference's vocabulary: PublicKey.verify_batch(rlc=True)
    (many signatures under one key) and PublicKeyShare.verify_decryption_shares_rlc (the loop of
    examples/threshold_enc.rs); same booleans as the per-item methods, a liar caught in both."""
    t, n = 2, 5
    sk_set = random_sk_set(t, rnd)
    pk_set = sk_set.public_keys()
    sk = SecretKey(rnd.randrange(o.R))
    pk = sk.public_key()
    msgs = [b"block %d" % i for i in range(70)]
    sigs = [sk.sign(m) for m in msgs]
    sigs[41] = sk.sign(b"something else")
    want = pk.verify_batch(sigs, msgs)
    assert want.tolist() == [i != 41 for i in range(70)]
    assert (pk.verify_batch(sigs, msgs, rlc=True) == want).all()
    pk_master = o.g1_from_uncompressed(pk_set.public_key().raw, check=False)
    cts = []
    for j in range(3):
        u, v, w = o.encrypt_with_r(pk_master, rnd.randrange(1, o.R), b"ciphertext %d" % j)
        cts.append(Ciphertext(o.g1_uncompressed(u), v, o.g2_uncompressed(w)))
    pk_shares = pk_set.public_key_shares(list(range(n)))
    shares = [[sk_set.secret_key_share(i).decrypt_share(ct) for i in range(n)] for ct in cts]
    shares[1][2] = shares[1][3]                                   # node 2 hands in somebody else's share
    ok = api.PublicKeyShare.verify_decryption_shares_rlc(pk_shares, shares, cts)
    for j in range(3):
        for i in range(n):
            assert ok[j, i] == pk_shares[i].verify_decryption_share(shares[j][i], cts[j]) == (not (j == 1 and i == 2))


def test_larger_threshold_general_path(engine, rnd):
    """t = 9 (three Straus chunks, general Lagrange path) and indices beyond the fast path's range."""
    t = 9
    sk_set = random_sk_set(t, rnd)
    pk_set = sk_set.public_keys()
    h = api.hash_g2(b"big committee")
    ids = sorted(rnd.sample(range(200), t + 1))
    shares = {i: sk_set.secret_key_share(i).sign_g2(h) for i in ids}
    sig = pk_set.combine_signatures(shares)
    assert sig.raw == o.g2_uncompressed(o.E2.mul(o.g2_from_uncompressed(h, check=False), sk_set.poly[0]))
    big = {i + 10 ** 12: SignatureShare(o.g2_uncompressed(o.E2.mul(o.g2_from_uncompressed(h, check=False), o.poly_evaluate(sk_set.poly, (i + 10 ** 12 + 1) % o.R))))
           for i in range(t + 1)}
    assert pk_set.combine_signatures(big) == sig


def test_full_batch_properties(engine, rnd):
    """Size-independent properties on a 4 096-job batch (the same checks bench.py applies at
    65 536): combine == master-key signature, verifies, corrupted jobs are caught."""
    from threshold_crypto_amd.workload import ThresholdSigWorkload
    t, N, B = 3, 10, 4096
    wl = ThresholdSigWorkload(engine, t, N, B)
    sig, st = engine.combine_g2(t, wl.idx, wl.shares)
    assert not st.any()
    msig, _ = engine.g2_mul(wl.master_sk_fr[None].copy(), wl.hashes)
    assert (msig[:, 0] == sig).all()
    bad = sig.copy()
    bad[::16] = np.roll(sig, -1, axis=0)[::16]  # every 16th job gets its neighbour's signature
    ok = engine.verify_g2(wl.master_pk, bad, wl.hashes)
    want = np.ones(B, dtype=np.uint8)
    want[::16] = 0
    assert (ok == want).all()
    assert engine.verify_sig(wl.master_pk, sig, wl.msg_flat, wl.msg_off).all()
    # a second, disjoint subset of signers gives the same signatures (uniqueness)
    idx2 = (9 - wl.idx[:, ::-1]).astype(np.uint64)  # complement-mirror keeps rows ascending
    fr = np.stack([np.frombuffer(s._bytes(), dtype=np.uint8) for s in wl.shares_sk])
    allsh, _ = engine.g2_mul(fr, np.ascontiguousarray(wl.hashes[:512]))
    rows = np.arange(512)[:, None]
    sig2, st2 = engine.combine_g2(t, np.ascontiguousarray(idx2[:512]), np.ascontiguousarray(allsh[rows, idx2[:512].astype(np.int64)]))
    assert not st2.any() and (sig2 == sig[:512]).all()


def test_encryption_workload_matches_oracle_and_decrypts(engine):
    """The threshold-encryption workload (encrypt_with_rng composed from batch entry points,
    src/lib.rs:128-137) equals the oracle's ciphertexts; ciphertexts verify; threshold decryption
    returns the plaintexts."""
    from threshold_crypto_amd.workload import ThresholdEncWorkload, _sha3_scalars, SEED
    t, N, B = 3, 10, 70
    we = ThresholdEncWorkload(engine, t, N, B)
    pk = o.g1_from_uncompressed(bytes(we.master_pk), check=False)
    rs = _sha3_scalars(b"tc/enc", B, SEED)
    for j in (0, 1, 69):
        r = int.from_bytes(bytes(rs[j]), "little")
        u, v, w = o.encrypt_with_r(pk, r, we.plain[j])
        assert bytes(we.u[j]) == o.g1_uncompressed(u) and bytes(we.w[j]) == o.g2_uncompressed(w)
        assert bytes(we.v[32 * j: 32 * j + 32]) == v
    assert engine.ciphertext_verify(we.u, we.v, we.off, we.w).all()
    out, st = engine.decrypt(t, we.idx, we.shares, we.v, we.off)
    assert not st.any() and bytes(out[: 32 * B]) == b"".join(we.plain)


def test_encrypt_and_public_key_shares_entry_points(engine, rnd):
    """tc_encrypt_batch (encrypt_with_rng, src/lib.rs:128-137) and tc_public_key_share_batch
    (public_key_share, :570-573): oracle equality, decryptability, and the share-validation loop of
    examples/threshold_sig.rs:115-131 in two launches."""
    sk_set = random_sk_set(3, rnd)
    pk_set = sk_set.public_keys()
    pk = pk_set.public_key()
    msgs = [bytes(rnd.randrange(256) for _ in range(n)) for n in (0, 1, 31, 64, 65, 200)] * 11
    rs = [rnd.randrange(1, o.R) for _ in msgs]
    cts = pk.encrypt_with_r_batch(rs, msgs)
    opk = o.g1_from_uncompressed(pk.raw, check=False)
    for j in (0, 3, 5, 65):
        u, v, w = o.encrypt_with_r(opk, rs[j], msgs[j])
        assert (cts[j].u, cts[j].v, cts[j].w) == (o.g1_uncompressed(u), v, o.g2_uncompressed(w))
    assert Ciphertext.verify_batch(cts).all()
    assert SecretKey(sk_set.poly[0]).decrypt(cts[4]) == msgs[4]
    n = 7
    pks = pk_set.public_key_shares(list(range(n)))
    assert [p.raw for p in pks] == [o.g1_uncompressed(o.public_key(sk_set.secret_key_share(i).fr)) for i in range(n)]
    msg = b"validate me"
    shares = [sk_set.secret_key_share(i).sign(msg) for i in range(n)]
    assert api.PublicKeyShare.verify_batch_shares(pks, shares, [msg] * n).all()


def test_bench_line_contract():
    """bench.py prints ONE compact JSON line (< 8 000 bytes, the only JSON object on stdout) with the contract's fields: the headline is K steps on ONE context (launches do not
    overlap, so the per-launch kernel time fits inside the step time), the two-contexts figure is the `streaming` object
    marked overlapped, the roofline fraction is a utilisation (<= 1)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "1", "--batch", "8192",
                          "--no-extras", "--cpu-seconds", "1", "--sustain-seconds", "0.2"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    # what the driver keeps is the LAST line of the last 8 000 bytes of stdout -- one compact JSON object
    # (tests/benchline.py reads it that way); the explanatory detail object is in bench_detail.json and, tagged, on stderr
    sys.path.insert(0, os.path.join(root, "tests"))
    import benchline
    line, d = benchline.parse(out.stdout, out.stderr)
    assert len(out.stdout.splitlines()[-1]) < 8000 and json.loads(out.stdout[-8000:].splitlines()[-1]) == line
    assert json.load(open(os.path.join(root, "bench_detail.json"))) == d
    for k in ("bound", "kernel", "kernel_ms", "frac", "frac_useful", "executed_macs_per_unit", "algorithmic_bytes_per_launch", "traffic", "peak", "unit", "achieved"):
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == d["roofline"]["frac"] and 0 < line["roofline"]["frac"] <= 1
    for k in ("value", "unit", "cores", "kind"):
        assert line["cpu_baseline"][k] == d["cpu_baseline"][k], k
    assert line["config"]["t"] == 3 and line["config"]["N"] == 10 and line["config"]["batch_per_gpu"] == 8192 and line["ranks"]["world_size"] == 1
    assert line["config3"]["value"] == d["config3"]["value"] and line["config3"]["kernel_ms"] == d["config3"]["kernel_ms"]
    assert line["config3"]["cpu_baseline"]["value"] == d["config3"]["cpu_baseline"]["value"] and 0 < line["config3"]["frac"] <= 1
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "streaming", "sustained", "config3", "ranks"):
        assert k in d, k
    assert d["metric"] == "combine_signatures/sec" and d["n_gpus"] == 1 and d["steps"] == 6 and d["higher_is_better"] is True
    assert d["config"]["steps_in_flight"] == 1 and d["config"]["overlapped"] is False and "workload" in d["config"]
    assert abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    r = d["roofline"]
    assert r["bound"] == "valu_int32_mac" and 0 < r["frac"] <= 1 and 0 < r["frac_timed_region"] <= 1 and r["peak"] > 30
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02          # one stream: a launch fits inside its step
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    s = d["streaming"]
    assert s["overlapped"] is True and s["contexts"] == 2 and s["value"] > 0
    assert d["sustained"]["seconds"] >= 0.15 and d["config3"]["kernel_ms"] > 0
    # r05: the CPU path beside the pairing half of BASELINE's metric too, rebuilt for this box's cores; duplicated lane work counted once
    c3 = d["config3"]["cpu_baseline"]
    assert c3["kind"] == "port" and c3["unit"] == "pairing_verifies/s" and c3["value"] > c3["single_thread_per_s"] > 0
    assert "march=native" in c["build"] and "march=native" in c3["build"]
    assert 0 < r["frac_useful"] <= r["frac"] and 0 < d["config3"]["roofline"]["frac_useful"] <= d["config3"]["roofline"]["frac"]


@pytest.mark.parametrize("config", [2, 5])
def test_bench_under_the_launcher_runs_its_collectives_over_rccl(config):
    """The driver's multi-GPU command shape with the one GPU this box has: `python -m torch.distributed.run --nproc-per-node 1
    bench.py --gpus 1`.  bench.py joins the launcher's rendezvous with backend nccl (= RCCL) bound to its device, so the
    key-set broadcast, the barriers around the timed region, the MAX of the region, the valid-count all-reduce and the
    record all-gather all run through RCCL on the GPU -- the code path of the N-rank run, which this box cannot start."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    extra = ["--batch", "8192", "--no-extras", "--sustain-seconds", "0.2"] if config == 2 else ["--config", "5", "--batch", "2048"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    sys.path.insert(0, os.path.join(root, "tests"))
    import benchline
    line, d = benchline.parse(out.stdout, out.stderr)
    assert line["ranks"]["backend"] == "nccl" and line["ranks"]["distinct_devices"] == 1 and line["ranks"]["rccl_version"] == d["ranks"]["rccl_version"]
    assert d["n_gpus"] == 1 and d["ranks"]["backend"] == "nccl" and d["ranks"]["world_size"] == 1 and len(d["ranks"]["devices"]) == 1
    assert d["ranks"]["rccl_version"] not in (None, "unknown") and "external launcher" in d["ranks"]["launched_by"]
    assert d["value"] > 0 and d["scaling"] == "weak"
    if config == 5:
        assert d["valid_total_all_ranks"] == 2048 and d["rank_records_start_jobs_valid_digest"][0][:3] == [0, 2048, 2048]
    else:
        assert d["verified_all"] is True


def test_null_operands_are_an_error_not_a_crash():
    """With a LIVE context: every batch entry point called with all sizes = 1 and every data pointer NULL answers
    TC_ERR_INVALID_ARG -- not a HIP error, not a segmentation fault (child process); with all sizes = 0 it is a no-op or an
    argument error.  The context stays usable afterwards."""
    import subprocess
    import sys
    code = r"""
import ctypes, sys
sys.path.insert(0, %r)
from threshold_crypto_amd import _native
lib = _native.load()
ctx = ctypes.c_void_p()
assert lib.tc_ctx_create(ctypes.byref(ctx), 0) == 0
bad = []
for name, args in sorted(_native.PROTOTYPES.items()):
    ones = [1 if t is ctypes.c_size_t else None for t in args]
    rc = getattr(lib, name)(ctx, *ones)
    if rc != _native.TC_ERR_INVALID_ARG:
        bad.append((name, "sizes 1", rc))
    zeros = [0 if t is ctypes.c_size_t else None for t in args]
    rc = getattr(lib, name)(ctx, *zeros)
    if rc not in (_native.TC_OK, _native.TC_ERR_INVALID_ARG):
        bad.append((name, "sizes 0", rc))
assert lib.tc_sync(ctx) == 0
lib.tc_ctx_destroy(ctx)
print("BAD", bad)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stderr[-2000:])
    assert "BAD []" in out.stdout, out.stdout[-2000:]


def test_device_operands_passed_as_temporaries_stay_alive_until_sync(engine):
    """Device-I/O calls return before their kernels have run (INTEGRATION.md "Device-I/O calls are asynchronous").  The Python mirror
    keeps the operand tensors of every call since the last sync() alive, so the natural `eng.verify_sig(pk.to(dev), ...)` with
    temporaries is not a use-after-free: torch would hand a dead tensor's memory to the next `.to(dev)` while the kernels still read it
    (round 6: a test of this repository did exactly that and hung the GPU on garbage message offsets).  Twenty calls with fresh
    temporaries each, interleaved with allocations of the same sizes, then one sync: every result equals the host-buffer call's."""
    import torch
    from threshold_crypto_amd.engine import pack_messages
    from threshold_crypto_amd.workload import ThresholdSigWorkload
    dev = torch.device("cuda", 0)
    t, N, B = 3, 10, 300
    wl = ThresholdSigWorkload(engine, t, N, B)
    sig, st = engine.combine_g2(t, wl.idx, wl.shares)
    blob, off = pack_messages(wl.msgs)
    bad = sig.copy()
    bad[5] = sig[6]
    want_ok = engine.verify_sig(wl.master_pk, bad, blob, off)
    want_sig = sig
    assert want_ok.tolist() == [1] * 5 + [0] + [1] * (B - 6)
    to = lambda a: torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a.copy()).to(dev)
    results = []
    for i in range(20):
        ok = engine.verify_sig(to(wl.master_pk), to(bad), to(blob), to(off))
        s2, st2 = engine.combine_g2(t, to(wl.idx), to(wl.shares))
        junk = [to(np.full_like(x, 0xAB)) for x in (bad, blob, wl.shares)]      # what torch would place into freed operand memory
        del junk
        results.append((ok, s2, st2))
    engine.sync()
    assert not engine._keep
    for ok, s2, st2 in results:
        assert (ok.cpu().numpy() == want_ok).all() and (s2.cpu().numpy() == want_sig).all() and not st2.cpu().numpy().any()
