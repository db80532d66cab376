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


def gather_stats(stats3):
    """stats3: [3] float64 device tensor (count, sum, sumsq) of the local shard.  Returns the
    [world,3] tensor of every rank's triple in rank order (the same on every rank): ONE all-gather
    of 24 bytes per rank, RCCL over xGMI with backend "nccl".  The consumer (oly_adv_normalize_parts)
    adds the triples by a balanced pairwise tree in rank order on the device, so no host arithmetic
    sits between the collective and the normalisation."""
    if not is_dist():
        return stats3.reshape(1, 3)
    world = dist.get_world_size()          # a one-rank group still goes through the collective
    if dist.get_backend() == "gloo" and stats3.is_cuda:          # rehearsal backend: bounce through the host
        host = stats3.detach().cpu().contiguous()
        out = torch.empty(world * 3, dtype=torch.float64)
        dist.all_gather_into_tensor(out, host)
        return out.view(world, 3).to(stats3.device)
    out = torch.empty(world * 3, dtype=torch.float64, device=stats3.device)
    dist.all_gather_into_tensor(out, stats3.contiguous())
    return out.view(world, 3)


def tree_sum(parts):
    """Balanced pairwise sum over dim 0 in index order (what oly_adv_normalize_parts does)."""
    a = [parts[i] for i in range(parts.shape[0])]
    while len(a) > 1:
        if len(a) % 2:
            a.append(torch.zeros_like(a[0]))
        a = [a[i] + a[i + 1] for i in range(0, len(a), 2)]
    return a[0]


def global_stats(stats3):
    """Sum of every rank's (count, sum, sumsq) triple, identical on every rank."""
    return tree_sum(gather_stats(stats3))


def mean_std_from_stats(stats3, ddof):
    """(mean, std) the way oly_adv_normalize derives them."""
    cnt = stats3[0]
    mean = stats3[1] / cnt
    var = torch.clamp((stats3[2] - cnt * mean * mean) / (cnt - ddof), min=0.0)
    return mean, torch.sqrt(var)


def _via_host(t):
    """gloo is the rehearsal backend (CPU collectives): device tensors bounce through the host."""
    return dist.get_backend() == "gloo" and t.is_cuda


def broadcast_parameters(modules, src=0):
    """One learner replicated on every rank: copy rank `src`'s parameters and buffers to all ranks
    (the reference has ONE policy / critic that every ray worker receives, rl/algos/ppo.py:200-207)."""
    if not is_dist():
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            if _via_host(t.data):
                h = t.data.cpu()
                dist.broadcast(h, src=src)
                t.data.copy_(h)
            else:
                dist.broadcast(t.data, src=src)


def allreduce_sum(values):
    """Element-wise sum over ranks of a short list of python numbers (float64), identical on every rank."""
    if not is_dist():
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    if dist.get_backend() != "gloo":
        t = t.cuda()
    dist.all_reduce(t)
    return t.cpu().tolist()


def allreduce_min(value):
    if not is_dist():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64)
    if dist.get_backend() != "gloo":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def allreduce_gradients(params, weight=None, total_weight=None):
    """Weighted mean of the gradients over ranks in ONE fused all-reduce (policy + critic ~ 160 k floats =
    0.6 MB: latency-bound on xGMI, so a single flat buffer).  weight = this rank's minibatch rows, total_weight =
    their sum over ranks: sum_r (rows_r / rows) grad_r is the gradient of the global minibatch the reference's
    single learner would see, whatever the shard sizes.  Without weights: the plain mean (equal shards).
    Every rank ends with bit-identical gradients (the collective's result is the same everywhere)."""
    if not is_dist():
        return
    for p in params:           # a parameter no loss term reached on THIS rank still takes part (zeros): the flat
        if p.grad is None:     # buffers of all ranks must have the same layout
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    if weight is not None:
        flat *= float(weight)
    if _via_host(flat):
        h = flat.cpu()
        dist.all_reduce(h)
        flat.copy_(h)
    else:
        dist.all_reduce(flat)
    flat /= float(total_weight) if weight is not None else dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def allreduce_flat(tensors, weight=None, total_weight=None):
    """allreduce_gradients for gradients that already live in flat buffers (ppo.KernelUpdate): the same weighted mean, one
    all-reduce per buffer, results written in place."""
    if not is_dist():
        return
    for flat in tensors:
        if weight is not None:
            flat *= float(weight)
        if _via_host(flat):
            h = flat.cpu()
            dist.all_reduce(h)
            flat.copy_(h)
        else:
            dist.all_reduce(flat)
        flat /= float(total_weight) if weight is not None else dist.get_world_size()
