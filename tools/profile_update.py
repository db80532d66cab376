#!/usr/bin/env python3
"""Kernel breakdown of one fused PPO minibatch update at minibatch 65536 (config 3's update phase):
run under `rocprofv3 --kernel-trace --stats`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import torch  # noqa: E402

from olympic_hip.engine import Engine  # noqa: E402
from olympic_hip.ppo import PPO, GraphedUpdate, MLPCritic, MLPGaussianActor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
if len(sys.argv) > 2:                 # "cublas" (= rocBLAS) or "cublaslt" (= hipBLASLt, torch's default here)
    torch.backends.cuda.preferred_blas_library(sys.argv[2])
eng = Engine(0)
dev = eng.device
torch.manual_seed(0)
ppo = PPO.__new__(PPO)
ppo.clip, ppo.vf_coeff = 0.2, 0.5
ppo.policy, ppo.old_policy, ppo.critic = MLPGaussianActor(41, 12).to(dev), MLPGaussianActor(41, 12).to(dev), MLPCritic(41).to(dev)
ppo.actor_optimizer = torch.optim.Adam(ppo.policy.parameters(), lr=1e-4, eps=1e-5, capturable=True)
ppo.critic_optimizer = torch.optim.Adam(ppo.critic.parameters(), lr=1e-4, eps=1e-5, capturable=True)
ppo.mirror_coeff, ppo.ent_coeff, ppo.grad_clip = 0.0, 0.0, 0.05
n = 4 * B
obs, act = torch.randn(n, 41, device=dev), torch.randn(n, 12, device=dev) * 0.3
ret, adv = torch.randn(n, 1, device=dev), torch.randn(n, 1, device=dev)
gu = GraphedUpdate(ppo, eng, B, 41, 12)
perm = torch.randperm(n, device=dev)
for i in range(4):
    gu(obs, act, ret, adv, perm[(i % 4) * B:(i % 4 + 1) * B])
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 20
for i in range(R):
    gu(obs, act, ret, adv, perm[(i % 4) * B:(i % 4 + 1) * B])
torch.cuda.synchronize()
print("blas:", torch.backends.cuda.preferred_blas_library(), "ms per update:", 1e3 * (time.perf_counter() - t0) / R)
