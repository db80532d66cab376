"""Worker of test_adv_stats_allgather_two_ranks_gloo: CPU/gloo rehearsal of the N>1 path.
The statistics kernel itself needs a GPU; here the local (count,sum,sumsq) triple is formed
with torch on the CPU and everything downstream (shard_range, all-gather, combine,
normalise) is the product's host logic."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
from olympic_hip import dist as odist  # noqa: E402


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    T, N = 50, 1001
    adv = np.random.default_rng(0).normal(0.2, 1.7, (T, N)).astype(np.float32)
    lo, hi = odist.shard_range(N)
    mine = torch.as_tensor(adv[:, lo:hi]).double()
    stats = torch.tensor([mine.numel(), mine.sum().item(), (mine * mine).sum().item()], dtype=torch.float64)
    parts = odist.gather_stats(stats)                         # [world,3], the product's one collective
    total = odist.global_stats(stats)
    assert torch.equal(total, odist.tree_sum(parts))
    res = {"total": total.numpy(), "parts": parts.numpy()}
    for ddof, eps, key in ((1, 1e-5, "ppo"), (0, 1e-8, "gail")):
        mean, std = odist.mean_std_from_stats(total, ddof)
        res[key] = ((mine - mean) / (std + eps)).float().numpy()
    # one learner replicated: parameters broadcast from rank 0, gradients averaged in one all-reduce
    torch.manual_seed(100 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 2))
    odist.broadcast_parameters([net])
    res["w_after_broadcast"] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    for i, p in enumerate(net.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    odist.allreduce_gradients(list(net.parameters()))
    res["g_after_allreduce"] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy()
    # unequal minibatches: rank r holds rows_r = 3 + 2 r rows of one data set; the row-weighted all-reduce of the ranks'
    # mean-loss gradients is the gradient of the mean loss over the CONCATENATED rows (what one learner would see),
    # through both entry points (module .grad lists and ppo.KernelUpdate's flat buffers)
    rows = [3 + 2 * r for r in range(world)]
    lo = sum(rows[:rank])
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(sum(rows), 5, generator=g), torch.randn(sum(rows), 2, generator=g)
    net.zero_grad()
    torch.nn.functional.mse_loss(net(X[lo:lo + rows[rank]]), Y[lo:lo + rows[rank]]).backward()
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
    odist.allreduce_gradients(list(net.parameters()), weight=rows[rank], total_weight=sum(rows))
    odist.allreduce_flat([flat], weight=rows[rank], total_weight=sum(rows))
    res["g_weighted"] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy()
    res["g_weighted_flat"] = flat.numpy()
    net.zero_grad()
    torch.nn.functional.mse_loss(net(X), Y).backward()
    res["g_concatenated"] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy()
    np.savez(os.path.join(out, f"rank{rank}.npz"), **res)
    if rank == 0:
        np.save(os.path.join(out, "full.npy"), adv.astype(np.float64))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
