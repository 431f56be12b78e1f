"""Multi-GPU sharding of independent jobs: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI on the GPUs, "gloo" in the CPU tests), contiguous job ranges, ONE broadcast of the key-set parameters per
key set and no data-path collective -- every (message, share-set) job is independent.  Results
stay on their rank; what travels back is bookkeeping: the valid counts (all-reduce of one int64) and one small
record per rank (all-gather) so that rank 0 can report the whole job.

The C ABI has the same thing for a single-process host (one thread per GPU): tc_group_* in include/tc_amd.h.
"""
import hashlib


def shard_range(total, world, rank):
    """Contiguous split of [0, total) into `world` ranges; returns (start, stop) of `rank`.
    The first total % world ranks get one extra job."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def joined(world):
    """True when this process takes part in collectives: more than one rank, or one rank that a launcher started
    and that joined the rendezvous (bench.py does, so `torch.distributed.run --nproc-per-node 1` drives every
    collective below through RCCL on a one-GPU box)."""
    if world > 1:
        return True
    import sys
    td = sys.modules.get("torch.distributed")
    return bool(td is not None and td.is_available() and td.is_initialized())


def broadcast_key_set(tensor, world, src=0):
    """Rank `src` holds key-set material (the PublicKeySet commitment, (t+1) x 96 B; for on-device signing also
    the N x 32 B table of secret key shares); every rank gets a copy in place.  (t+1)*96 B at t=67 is 6.5 KB, the
    share table at N=200 6.4 KB: latency-bound, one call per key set."""
    if not joined(world):
        return tensor
    import torch.distributed as dist
    dist.broadcast(tensor, src=src)
    return tensor


def total_count(local_count, world, device=None):
    """Sum of per-rank counts (1 x int64 all-reduce)."""
    if not joined(world):
        return int(local_count)
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def max_over_ranks(value, world, device=None):
    """Max of a per-rank float (the bench contract's MAX over ranks of the timed region)."""
    if not joined(world):
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def digest64(array_bytes):
    """64-bit digest of a rank's result bytes (what rank 0 gathers instead of the results themselves)."""
    return int.from_bytes(hashlib.sha3_256(array_bytes).digest()[:8], "little") & 0x7FFFFFFFFFFFFFFF


def gather_records(record, world, device=None):
    """All-gather of one small int64 record per rank (e.g. [start, count, valid, digest]); returns a list of
    lists, rank order."""
    rec = [int(x) for x in record]
    if not joined(world):
        return [rec]
    import torch
    import torch.distributed as dist
    mine = torch.tensor(rec, dtype=torch.int64, device=device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [[int(v) for v in o.tolist()] for o in out]
