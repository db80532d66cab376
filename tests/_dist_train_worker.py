"""Worker of test_ppo_train_two_ranks_share_one_gpu: PPO.train under a 2-rank process group (gloo rendezvous,
both ranks on cuda:0), each rank with its OWN environments (different seeds, optionally different counts) and its
own differently initialised networks.  Writes this rank's final parameters and bookkeeping to <out>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from olympic_hip import specs  # noqa: E402
from olympic_hip.a3 import ReplayA3Physics, VecA3Env  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402
from olympic_hip.ppo import PPO, MLPCritic, MLPGaussianActor  # noqa: E402
from olympic_hip.synthetic import A3_FLOOR_BODY, A3_GEOM_BODYID, A3_LFOOT_BODY, A3_RFOOT_BODY, a3_synthetic_blocks  # noqa: E402


def main():
    out, n0, n1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    N = (n0, n1)[rank]

    def env_fn():
        blocks = {k: torch.as_tensor(v).cuda() for k, v in a3_synthetic_blocks(N, 9, seed=40 + rank, p_bad=0.02).items()}
        env = VecA3Env(specs.A3Spec(mass=41.5), N, Engine(0), ReplayA3Physics(blocks), A3_GEOM_BODYID, A3_FLOOR_BODY,
                       A3_RFOOT_BODY, A3_LFOOT_BODY, rs=np.random.RandomState(rank))
        env.device = env.eng.device
        return env
    args = dict(gamma=0.99, lam=0.95, lr=1e-3, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=256, epochs=2,
                max_traj_len=8, use_gae=False, num_procs=N, max_grad_norm=0.05, mirror_coeff=0.0, eval_freq=1)
    ppo = PPO(args, out)                                   # BOTH ranks point at the same directory
    torch.manual_seed(100 + rank)                          # different initial weights: rank 0's must win
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    first = pi.means.weight.detach().clone()
    ppo.highest_reward = -1e9
    hist = ppo.train(env_fn, pi, vf, n_itr=2, verbose=False)
    flat = torch.cat([p.detach().reshape(-1) for p in list(pi.parameters()) + list(vf.parameters())]).cpu().numpy()
    np.savez(os.path.join(out, f"rank{rank}.npz"), params=flat, first=first.cpu().numpy().reshape(-1),
             highest=ppo.highest_reward, eval_returns=np.array([h["eval_return"] for h in hist]),
             ep_returns=np.array([h["ep_return"] for h in hist]), losses=np.array([h["losses"] for h in hist]),
             total_steps=ppo.total_steps)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
