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
