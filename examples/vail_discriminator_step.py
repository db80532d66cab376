#!/usr/bin/env python3
"""One VAIL iteration's reward side on UnitreeH1 (reference: imitation_lib/imitation/gail_TRPO.py
fit(): discriminator fit on policy vs demonstration states, make_discrim_reward, compute_gae,
advantage normalisation) with every step after the physics on the GPU.  The policy update itself
(mushroom's TRPO) is not part of this repository.

    python examples/vail_discriminator_step.py --num_envs 4096 --steps 100
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "olympics-mujoco_amd"))
import torch  # noqa: E402

from olympic_hip.envs import LocoEnvBase  # noqa: E402
from olympic_hip.gail import DiscriminatorReward, DiscriminatorTrainer, GAILAdvantage, VariationalDiscriminator, VDBLoss  # noqa: E402
from olympic_hip.ppo import MLPCritic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    torch.manual_seed(0)
    env = LocoEnvBase.make("UnitreeH1.walk.real", num_envs=args.num_envs, seed=0)
    vec, eng = env.vec, env.vec.eng
    gen = torch.Generator(device="cuda").manual_seed(0)
    demo = env.create_dataset()["states"]                       # expert states (trajectory preprocessing on the host)
    n_obs = vec.spec.n_obs
    disc = DiscriminatorReward(eng, VariationalDiscriminator(n_obs).cuda(), state_mask=vec.get_kinematic_obs_mask())
    trainer = DiscriminatorTrainer(disc, demo, VDBLoss(info_constraint=0.1, lr_beta=1e-5), lr=5e-5)
    critic = MLPCritic(n_obs).cuda()
    T, N = args.steps, args.num_envs
    x = torch.empty((T + 1, N, n_obs), dtype=torch.float32, device="cuda")
    r_env = torch.empty((T, N), dtype=torch.float32, device="cuda")
    absorbing = torch.empty((T, N), dtype=torch.bool, device="cuda")
    last = torch.empty((T, N), dtype=torch.bool, device="cuda")
    x[0] = vec.reset().to(torch.float32)
    for t in range(T):                                          # random policy on the kinematic stand-in
        o, r, a, info = vec.step(torch.empty((N, vec.spec.n_act), device="cuda").uniform_(-1, 1, generator=gen))
        x[t + 1], r_env[t], absorbing[t], last[t] = o.to(torch.float32), r, a, info["last"]
    last[-1] = True
    loss = trainer.fit(x[:-1].reshape(T * N, n_obs)[:8192].contiguous(), gen)
    with torch.no_grad():
        r, v_target, adv = GAILAdvantage(eng, disc, critic, gamma=0.99, lam=0.97)(x[:-1].contiguous(), x[1:].contiguous(),
                                                                                   r_env, absorbing, last)
    print(f"discriminator loss {loss[0]:.4f}; discriminator reward mean {float(r.mean()):.4f}; "
          f"normalised advantage mean {float(adv.mean()):+.2e} std {float(adv.std(unbiased=False)):.4f}; "
          f"absorbing fraction {float(absorbing.float().mean()):.4f}")


if __name__ == "__main__":
    main()
