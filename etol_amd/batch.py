"""Instance sharding of a Monte-Carlo batch across ranks, and the one collective of the path.

A batch of independent VGP instances shards embarrassingly: instance s belongs to rank
floor(s * world / n) (contiguous blocks); nothing is exchanged while the
instances are evaluated or solved.  After the solve, ONE gather brings the trajectories
[X | U] of every instance to rank 0: torch.distributed, backend "nccl" (= RCCL over xGMI) for
device tensors, "gloo" for the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(n_instances, world, rank):
    """[first, last) of the contiguous block of instances owned by `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    first = (n_instances * rank) // world
    last = (n_instances * (rank + 1)) // world
    return first, last


def owner_of(instance, n_instances, world):
    """Rank that owns `instance` (inverse of shard_range)."""
    for r in range(world):
        first, last = shard_range(n_instances, world, r)
        if first <= instance < last:
            return r
    raise ValueError("instance out of range")


def gather_trajectories(X, U, n_instances, dst=0):
    """X [b][ns][M], U [b][nc][M] of this rank's block -> on `dst`: (X_all, U_all) for all
    n_instances in instance order; None elsewhere.  Blocks may differ in size by one."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return X, U
    b, ns, M = X.shape
    nc = U.shape[1]
    first, last = shard_range(n_instances, world, rank)
    assert b == last - first, "local block does not match shard_range"
    bmax = max(shard_range(n_instances, world, r)[1] - shard_range(n_instances, world, r)[0] for r in range(world))
    buf = torch.zeros((bmax, ns + nc, M), dtype=X.dtype, device=X.device)
    buf[:b, :ns] = X
    buf[:b, ns:] = U
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    Xs, Us = [], []
    for r in range(world):
        f, l = shard_range(n_instances, world, r)
        Xs.append(out[r][: l - f, :ns])
        Us.append(out[r][: l - f, ns:])
    return torch.cat(Xs), torch.cat(Us)
