"""Multi-GPU plumbing: one process per GPU, environments sharded by index, ONE exchange per
PPO/GAIL iteration - an all-gather of (count, sum, sum of squares) of the local advantages,
24 bytes per rank, over RCCL/xGMI (backend "nccl") or gloo on CPU.  Nothing else
communicates: rollouts, scans and rewards are rank-local."""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def shard_range(n_total, rank=None, world=None):
    """Contiguous env-index range [lo, hi) of this rank (near-equal shards)."""
    if rank is None:
        rank = dist.get_rank() if is_dist() else 0
    if world is None:
        world = dist.get_world_size() if is_dist() else 1
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_stats(stats3):
    """stats3: [3] float64 tensor (count, sum, sumsq) of the local shard, on the device the
    process group communicates from.  Returns the sum over ranks (same on every rank, ranks
    added in rank order so the result is bit-identical everywhere)."""
    if not is_dist() or dist.get_world_size() == 1:
        return stats3
    world = dist.get_world_size()
    gathered = [torch.empty_like(stats3) for _ in range(world)]
    dist.all_gather(gathered, stats3.contiguous())
    total = gathered[0].clone()
    for g in gathered[1:]:
        total += g
    return total


def mean_std_from_stats(stats3, ddof):
    """(mean, std) the way oly_adv_normalize derives them."""
    cnt = stats3[0]
    mean = stats3[1] / cnt
    var = torch.clamp((stats3[2] - cnt * mean * mean) / (cnt - ddof), min=0.0)
    return mean, torch.sqrt(var)
