#!/usr/bin/env python3
"""PPO minibatch update step (rl/algos/ppo.py:232-282,352-400): the reference's op-by-op loss
(`PPO.update_policy`, ~30 elementwise torch kernels + autograd twins) against the fused
oly_ppo_loss / oly_mirror_loss path (`PPO.update_policy_fused`), same networks, same optimiser.
Wall clock per update (forward + backward + clip + Adam) with a device sync on both sides.
Prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from olympic_hip.engine import Engine  # noqa: E402
from olympic_hip.ppo import PPO, GraphedUpdate, KernelUpdate, MLPCritic, MLPGaussianActor  # noqa: E402
from olympic_hip.wrappers import SymmetricEnv, _signed_perm  # noqa: E402

MIRROR_OBS = [0.1, -1, 2, -3, -4, 5, -6, -13, -14, 15, 16, 17, 18, -7, -8, 9, 10, 11, 12, -25, -26, 27, 28, 29, 30, -19, -20,
              21, 22, 23, 24, 31, 32, 33, -34, 35, -36, 37, -38, 39, -40]
MIRROR_ACT = [-6, -7, 8, 9, 10, 11, -0.1, -1, 2, 3, 4, 5]


def main():
    eng = Engine(0)
    dev = eng.device
    torch.manual_seed(0)
    ppo = PPO.__new__(PPO)
    ppo.clip, ppo.vf_coeff = 0.2, 0.5

    class Dummy:
        base_obs_len = 41
    sym = SymmetricEnv(Dummy, mirrored_obs=MIRROR_OBS, mirrored_act=MIRROR_ACT, clock_inds=[31, 32])
    a_src, a_sgn = _signed_perm(MIRROR_ACT)
    a_src = torch.as_tensor(a_src.astype(np.int32), device=dev)
    a_sgn = torch.as_tensor(a_sgn, device=dev)
    out = {}
    for B in (64, 4096, 65536):
        obs = torch.randn(B, 41, device=dev)
        obs[:, 31:33] = torch.tanh(obs[:, 31:33])              # clock entries are sin/cos
        act = torch.randn(B, 12, device=dev) * 0.3
        ret, adv = torch.randn(B, 1, device=dev), torch.randn(B, 1, device=dev)
        row = {}
        for mirror in (False, True):
            for fused in (False, True, "graph", "kernel"):
                ppo.policy = MLPGaussianActor(41, 12).to(dev)
                ppo.old_policy = MLPGaussianActor(41, 12).to(dev)
                ppo.critic = MLPCritic(41).to(dev)
                opt_a = torch.optim.Adam(ppo.policy.parameters(), lr=1e-4, eps=1e-5, capturable=fused == "graph")
                opt_c = torch.optim.Adam(ppo.critic.parameters(), lr=1e-4, eps=1e-5, capturable=fused == "graph")
                if fused == "graph":
                    ppo.actor_optimizer, ppo.critic_optimizer = opt_a, opt_c
                    ppo.mirror_coeff, ppo.ent_coeff, ppo.grad_clip = 0.4, 0.0, 0.05
                    gu = GraphedUpdate(ppo, eng, B, 41, 12, sym.mirror_clock_observation if mirror else None,
                                       a_src if mirror else None, a_sgn if mirror else None)
                    idx = torch.arange(B, device=dev)
                if fused == "kernel":
                    ku = KernelUpdate(eng, ppo.policy, ppo.critic, ppo.old_policy, 0.2, 0.5, 0.4,
                                      sym.mirror_clock_observation if mirror else None, a_src if mirror else None,
                                      a_sgn if mirror else None, lr=1e-4, eps=1e-5, max_grad_norm=0.05)
                    ku.begin(obs)
                    idx32 = torch.arange(B, device=dev, dtype=torch.int32)
                    ret1, adv1 = ret.reshape(-1).contiguous(), adv.reshape(-1).contiguous()

                def step():
                    if fused == "kernel":
                        ku.step(obs, act, ret1, adv1, idx32)      # K14 gradients + clip + Adam: three launches
                        return
                    if fused == "graph":
                        gu(obs, act, ret, adv, idx)
                        return
                    if fused:
                        o = ppo.update_policy_fused(eng, obs, act, ret, adv, sym.mirror_clock_observation if mirror else None,
                                                    a_src if mirror else None, a_sgn if mirror else None)
                    else:
                        o = ppo.update_policy(obs, act, ret, adv, 1, sym.mirror_clock_observation if mirror else None,
                                              sym.mirror_action if mirror else None)
                    opt_a.zero_grad()
                    opt_c.zero_grad()
                    (o[0] + 0.4 * o[4] + 0.0 * o[1] + o[2]).sum().backward()
                    torch.nn.utils.clip_grad_norm_(ppo.policy.parameters(), 0.05)
                    opt_a.step()
                    torch.nn.utils.clip_grad_norm_(ppo.critic.parameters(), 0.05)
                    opt_c.step()
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 100
                for _ in range(reps):
                    step()
                torch.cuda.synchronize()
                row[("mirror_" if mirror else "") + {False: "torch", True: "fused", "graph": "fused_graph", "kernel": "kernel"}[fused] + "_ms"] = (time.perf_counter() - t0) / reps * 1e3
        out[f"minibatch_{B}"] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
