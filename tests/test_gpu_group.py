"""tc_group_* (the C ABI's multi-GPU surface) on the one GPU this box has: a 1-rank group runs its key-set
broadcast and its valid-count all-reduce on RCCL (ncclCommInitAll over one device); a group with device 0 listed
twice exercises the two-worker sharding (no RCCL there -- RCCL refuses duplicate devices -- documented test
configuration).  Results must equal direct single-context calls byte for byte."""
import numpy as np
import pytest

from threshold_crypto_amd.engine import Group, pack_messages
from threshold_crypto_amd.workload import ThresholdSigWorkload, key_set, messages
from threshold_crypto_amd.config5 import signer_subsets_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wl(engine):
    return ThresholdSigWorkload(engine, 3, 10, 300)


def _device_count():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


# [0, 1]: two PHYSICAL devices -- RCCL's broadcast and all-reduce cross xGMI.  Runs by itself wherever a second GPU is
# visible (row (e) has never met RCCL across two devices); skipped on the one-GPU boxes of this pool.
@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 1]])
def test_group_matches_single_context(engine, wl, devices):
    if len(set(devices)) > _device_count():
        pytest.skip("needs %d GPUs, this node shows %d" % (len(set(devices)), _device_count()))
    g = Group(devices)
    try:
        assert g.size() == len(devices) and g.uses_rccl() == (len(set(devices)) == len(devices))
        spans = [g.shard(wl.B, r) for r in range(g.size())]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == wl.B and all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
        commit = np.stack([np.frombuffer(c, dtype=np.uint8) for c in wl.sks.public_keys(engine).commit])
        g.set_keyset(commit)
        for r in range(g.size()):
            assert (g.get_keyset(r) == commit).all()            # every rank's HBM holds rank 0's commitment
        sig, st = g.combine_signatures(wl.idx, wl.shares)
        want, wst = engine.combine_g2(3, wl.idx, wl.shares)
        assert not st.any() and (sig == want).all()
        bad = sig.copy()
        bad[7] = sig[8]
        bad[299] = sig[0]
        ok, nvalid = g.verify_g2(bad, wl.hashes)
        assert nvalid == wl.B - 2 and ok.sum() == wl.B - 2 and ok[7] == 0 and ok[299] == 0
    finally:
        g.close()


def test_group_config5_pipeline(engine):
    """tc_group_sign_combine_verify: BASELINE config 5's flow (t = 8 here: the large-threshold path) from one call."""
    t, N, B = 8, 12, 70
    sks = key_set(t)
    commit = np.stack([np.frombuffer(c, dtype=np.uint8) for c in sks.public_keys(engine).commit])
    sk_table = np.stack([np.frombuffer(sks.secret_key_share(i)._bytes(), dtype=np.uint8) for i in range(N)])
    idx = signer_subsets_np(B, N, t)
    flat, off = pack_messages(messages(B))
    g = Group([0, 0])
    try:
        g.set_keyset(commit)
        up0, down0 = g.transfer_bytes()
        sig, ok, nvalid = g.sign_combine_verify(sk_table, idx, flat, off)
        up1, down1 = g.transfer_bytes()
        # a second call reuses the persistent workers and the per-rank device buffers
        sig2, ok2, nvalid2 = g.sign_combine_verify(sk_table, idx, flat, off)
        with pytest.raises(ValueError):
            bad_off = off.copy()
            bad_off[3] = off[2] - 1
            g.sign_combine_verify(sk_table, idx, flat, bad_off)
    finally:
        g.close()
    assert ok.all() and nvalid == B and (sig2 == sig).all() and ok2.all() and nvalid2 == B
    # hash points and share signatures (B x n x 192 B = %d KB here) stay in the ranks' HBM: what crosses PCIe is
    # the messages, offsets, signer indices and the secret share table going up, signatures + ok + status coming back
    share_bytes = B * (t + 1) * 192
    ranks = 2
    up_budget = flat.nbytes + (B + ranks) * 8 + idx.nbytes + ranks * sk_table.nbytes + 64 * ranks
    down_budget = B * (192 + 1 + 1) + 64 * ranks
    assert up1 - up0 <= up_budget and down1 - down0 <= down_budget, (up1 - up0, down1 - down0)
    assert (up1 - up0) + (down1 - down0) < share_bytes / 4 and (up1 - up0) + (down1 - down0) < (1 << 20)
    h = engine.hash_g2(flat, off)
    msig, st = engine.g2_mul(np.frombuffer(sks.poly[0].to_bytes(32, "little"), dtype=np.uint8)[None].copy(), h)
    assert (msig[:, 0] == sig).all()


def test_group_config5_pipeline_on_two_devices(engine):
    """The same one-call config-5 flow across two PHYSICAL GPUs (key set over RCCL/xGMI, counts all-reduced): equal to the [0, 0]
    group's and the single context's results.  Skipped unless a second GPU is visible."""
    if _device_count() < 2:
        pytest.skip("needs 2 GPUs, this node shows %d" % _device_count())
    t, N, B = 8, 12, 70
    sks = key_set(t)
    commit = np.stack([np.frombuffer(c, dtype=np.uint8) for c in sks.public_keys(engine).commit])
    sk_table = np.stack([np.frombuffer(sks.secret_key_share(i)._bytes(), dtype=np.uint8) for i in range(N)])
    idx = signer_subsets_np(B, N, t)
    flat, off = pack_messages(messages(B))
    got = {}
    for devices in ([0, 0], [0, 1]):
        g = Group(devices)
        try:
            assert g.uses_rccl() == (devices == [0, 1])
            g.set_keyset(commit)
            assert all((g.get_keyset(r) == commit).all() for r in range(2))
            got[tuple(devices)] = g.sign_combine_verify(sk_table, idx, flat, off)
        finally:
            g.close()
    (s0, ok0, n0), (s1, ok1, n1) = got[(0, 0)], got[(0, 1)]
    assert (s0 == s1).all() and ok0.all() and ok1.all() and n0 == n1 == B


def test_bench_gpus_2_on_two_devices():
    """`bench.py --gpus 2 --config 5` as the driver starts it, on two PHYSICAL GPUs when the node has them: two ranks joined over
    RCCL (backend nccl), two distinct devices, per-rank records equal to the emulated two-rank run on ONE GPU (same slices, same
    signatures), and a line that stays inside the driver's 8 KB.  Skipped on a one-GPU box."""
    if _device_count() < 2:
        pytest.skip("needs 2 GPUs, this node shows %d" % _device_count())
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    import benchline
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    common = ["--config", "5", "--batch", "4096", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + common, capture_output=True, text=True, timeout=1500, cwd=root, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    line, d = benchline.parse(two.stdout, two.stderr)
    assert line["n_gpus"] == 2 and line["ranks"] == dict(line["ranks"], world_size=2, backend="nccl", distinct_devices=2)
    assert d["ranks"]["world_size"] == 2 and len(set(d["ranks"]["devices"])) == 2 and d["valid_total_all_ranks"] == 2 * 4096
    emu = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--emulate-world", "2"] + common, capture_output=True, text=True,
                         timeout=1500, cwd=root, env=env)
    assert emu.returncode == 0, emu.stderr[-3000:]
    _, e = benchline.parse(emu.stdout, emu.stderr)
    assert d["rank_records_start_jobs_valid_digest"] == e["rank_records_start_jobs_valid_digest"]
    assert [r[:3] for r in d["rank_records_start_jobs_valid_digest"]] == [[0, 4096, 4096], [4096, 4096, 4096]]
