#!/usr/bin/env python3
"""K11 at the rollout batch: run under `rocprofv3 --kernel-trace --stats` for the kernel's own duration
(the Python launch path is slower than the kernel at N = 4096)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import torch  # noqa: E402

from olympic_hip.engine import Engine  # noqa: E402
from olympic_hip.mlp import FusedMLPForward  # noqa: E402
from olympic_hip.ppo import MLPCritic, MLPGaussianActor  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = Engine(0)
torch.manual_seed(0)
pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
fw = FusedMLPForward(eng, pi, vf)
x = torch.randn(N, 41, device="cuda")
for _ in range(200):
    fw(x)
torch.cuda.synchronize()
