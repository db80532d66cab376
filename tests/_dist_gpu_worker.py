"""Worker of test_config5_two_ranks_share_one_gpu: two ranks on ONE GPU (gloo rendezvous, both on
cuda:0) run the real config-5 tail on their env shard: oly_return_scan_stats -> all-gather of the
device triple -> oly_adv_normalize_parts.  Rank 0 also runs the whole block in one piece."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
from olympic_hip import _abi, dist as odist  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402
from olympic_hip.rollout import PPORollout, RolloutBuffer  # noqa: E402


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    eng = Engine(0)
    T, N = 400, 8192                                   # 2 x 4096 environments
    rng = np.random.default_rng(77)                    # the same block on every rank
    rew = rng.uniform(-0.3, 1.0, (T, N))
    val, nval = (rng.normal(0, 1, (T, N)).astype(np.float32) for _ in range(2))
    last = rng.uniform(size=(T, N)) < 1 / 300
    flags = (last * _abi.FLAG_LAST + (last & (rng.uniform(size=(T, N)) < 0.5)) * _abi.FLAG_ABSORBING).astype(np.uint8)

    def block(lo, hi):
        buf = RolloutBuffer(T, hi - lo, 1, 1, eng.device, reward_dtype=torch.float64)
        for name, a in (("rewards", rew), ("values", val), ("next_values", nval), ("flags", flags)):
            getattr(buf, name).copy_(torch.as_tensor(np.ascontiguousarray(a[:, lo:hi])))
        return buf
    lo, hi = odist.shard_range(N)
    post = PPORollout(eng, gamma=0.99, lam=0.95, eps=1e-5)
    buf = block(lo, hi)
    ret, adv = post.finish(buf, normalize=True)        # scan + fused stats -> all-gather -> normalise
    parts = odist.gather_stats(post._stats)
    res = dict(lo=lo, hi=hi, ret=ret.cpu().numpy(), adv_norm=adv.cpu().numpy(), parts=parts.cpu().numpy(),
               local=post._stats.cpu().numpy())
    dist.barrier()
    if rank == 0:
        # one process holding all environments: the world-1 code path of the same objects
        whole = block(0, N)
        st = torch.zeros(3, dtype=torch.float64, device=eng.device)
        r1, a1 = eng.return_scan(_abi.SCAN_RETURN, 0.99, 0.95, whole.rewards, whole.values, whole.next_values,
                                 whole.flags, stats3=st)
        res["single_ret"], res["single_adv"] = r1.cpu().numpy(), a1.cpu().numpy()
        res["single_stats"] = st.cpu().numpy()
        eng.adv_normalize(a1, st, 1, 1e-5)
        res["single_adv_norm"] = a1.cpu().numpy()
    np.savez(os.path.join(out, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
