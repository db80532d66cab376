#!/usr/bin/env python3
"""Config 3 end to end on one GPU: N StickFigureA3 envs x T steps per PPO iteration with a
synthetic physics readback resident on the device (no MuJoCo in the image), policy 41->256->256->12
and critic in PyTorch-ROCm, K5/K3/K2 per step, K6/K7 per iteration, K9 + graph for the updates.
Prints the reference's own instrumentation (sampling time, optimiser time, fps = total_steps /
elapsed, rl/algos/ppo.py:322-332,437-438) as one JSON object."""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from olympic_hip import _abi, specs  # noqa: E402
from olympic_hip.a3 import ReplayA3Physics, VecA3Env  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402
from olympic_hip.ppo import PPO, MLPCritic, MLPGaussianActor  # noqa: E402


def synthetic_blocks(N, K, gen):
    rnd = lambda *s: torch.empty(s, dtype=torch.float64, device="cuda").normal_(0, 1, generator=gen)
    step = torch.tensor([0.3, 0.15, 0.0], dtype=torch.float64, device="cuda")
    base = torch.zeros((N, 3), dtype=torch.float64, device="cuda") + step
    b = dict(qpos=rnd(K, N, 25), qvel=rnd(K, N, 24), act_len=rnd(K, N, 12), act_vel=rnd(K, N, 12),
             lf_pos=base + 0.1 * rnd(K, N, 3), rf_pos=base + 0.3 * rnd(K, N, 3), lf_vel=0.2 * rnd(K, N, 3),
             rf_vel=0.2 * rnd(K, N, 3), root_pos=base + torch.tensor([0, 0, 0.8], device="cuda") + 0.05 * rnd(K, N, 3),
             root_quat=rnd(K, N, 4), head_pos=base + torch.tensor([0, 0, 1.2], device="cuda") + 0.05 * rnd(K, N, 3),
             ncon=torch.randint(0, 5, (K, N), device="cuda", generator=gen, dtype=torch.int32),
             geom1=torch.zeros((K, N, 16), dtype=torch.int32, device="cuda"),
             geom2=torch.randint(8, 13, (K, N, 16), device="cuda", generator=gen, dtype=torch.int32),
             force6=100 * rnd(K, N, 16, 6), cpos_z=0.01 * rnd(K, N, 16))
    b["root_quat"] = b["root_quat"] / b["root_quat"].norm(dim=-1, keepdim=True)
    return {k: v.contiguous() for k, v in b.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--T", type=int, default=400)
    ap.add_argument("--itr", type=int, default=3)
    ap.add_argument("--minibatch", type=int, default=65536)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--mode", default="kernel", choices=["torch_losses", "fused", "fused_graph", "kernel"],
                    help="kernel: K14 gradients (one launch per minibatch) + torch clip / Adam")
    ap.add_argument("--rollout", default="host", choices=["host", "device"],
                    help="host: one env.step per python iteration (K5, K3, K2 launches); device: one fused K10 launch "
                         "per vec step, replayed from a HIP graph, resets on the device")
    ap.add_argument("--tuned-gemms", action="store_true", help="PPO.tuned_gemms: torch TunableOp for the update GEMMs")
    args = ap.parse_args()
    N, T = args.N, args.T
    gen = torch.Generator(device="cuda").manual_seed(1)
    sp = specs.A3Spec(mass=41.5)
    blocks = synthetic_blocks(N, 32, gen)          # 32 distinct readback steps, replayed cyclically
    gb = np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32)
    seq = torch.zeros((N, 20, 4), dtype=torch.float64, device="cuda")
    seq[:, :, 0] = 0.3 * torch.arange(20, device="cuda")
    seq[:, :, 1] = 0.15 * (1 - 2 * (torch.arange(20, device="cuda") % 2))

    class Env(VecA3Env):
        def __init__(self):
            super().__init__(sp, N, Engine(0), ReplayA3Physics(blocks), gb, 0, 7, 10)
            self.device = self.eng.device
            self.state["sequence"].copy_(seq)
            self.state["seq_len"].fill_(20)
            self.state["mode"].fill_(_abi.MODE_FORWARD)
            self.state["t2"].fill_(1)

        def reset(self, env_mask=None):
            if env_mask is None:
                self.state["phase"].zero_()
            else:
                self.state["phase"][env_mask] = 0
            return torch.zeros((N, 41), device="cuda")
    hp = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=args.minibatch,
              epochs=args.epochs, max_traj_len=T, use_gae=False, num_procs=N, max_grad_norm=0.05, mirror_coeff=0.0,
              eval_freq=10 ** 9)
    ppo = PPO(hp, tempfile.mkdtemp(prefix="oly_ppo_"))
    ppo.fused_loss, ppo.use_graph = args.mode != "torch_losses", args.mode == "fused_graph"
    ppo.update_kernel = args.mode == "kernel"
    ppo.use_graph_rollout = args.mode == "fused_graph"
    ppo.use_device_rollout = args.rollout == "device"
    ppo.tuned_gemms = args.tuned_gemms
    torch.manual_seed(0)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    hist = ppo.train(Env, pi, vf, n_itr=args.itr, verbose=False)
    last = hist[-1]
    steady = hist[1:] or hist
    out = dict(config=dict(N=N, T=T, minibatch=args.minibatch, epochs=args.epochs, mode=args.mode, physics="synthetic replay", tuned_gemms=args.tuned_gemms),
               sample_s=float(np.mean([h["sample_s"] for h in steady])),
               optim_s=float(np.mean([h["optim_s"] for h in steady])),
               env_steps_per_iteration=N * T,
               env_steps_per_s_sampling=N * T / float(np.mean([h["sample_s"] for h in steady])),
               env_steps_per_s_iteration=N * T / float(np.mean([h["sample_s"] + h["optim_s"] for h in steady])),
               fps_reference_definition=last["fps"], losses_finite=bool(np.isfinite(last["losses"]).all()))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
