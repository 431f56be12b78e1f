"""Multi-GPU sharding of independent jobs: one process per GPU, contiguous job ranges, ONE
broadcast of the key-set parameters (RCCL over xGMI on GPUs, gloo in the CPU tests) and no
data-path collective -- every (message, share-set) job is independent.
"""


def shard_range(total, world, rank):
    """Contiguous split of [0, total) into `world` ranges; returns (start, stop) of `rank`.
    The first total % world ranks get one extra job."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_key_set(commit_tensor, world, src=0):
    """Rank `src` holds the PublicKeySet commitment ((t+1) x 96 B, uint8 tensor); every rank gets
    a copy.  (t+1)*96 B at t=67 is 6.5 KB: latency-bound, one call per key set."""
    if world <= 1:
        return commit_tensor
    import torch.distributed as dist
    dist.broadcast(commit_tensor, src=src)
    return commit_tensor


def total_count(local_count, world, device=None):
    """Sum of per-rank valid counts (optional bookkeeping collective: 1 x int64 all-reduce)."""
    if world <= 1:
        return int(local_count)
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
