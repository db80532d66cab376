#!/usr/bin/env python3
"""Kernel-level cost of the policy / critic forward at the rollout batch (N = 4096, fp32):
run under `rocprofv3 --kernel-trace --stats` to see what the GEMM and ReLU launches take."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import torch  # noqa: E402

from olympic_hip.ppo import MLPCritic, MLPGaussianActor  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
x = torch.randn(N, 41, device="cuda")
with torch.no_grad():
    for _ in range(20):
        pi(x), vf(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        pi(x), vf(x)
    torch.cuda.synchronize()
    print("eager us per (actor+critic):", 1e6 * (time.perf_counter() - t0) / 300)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        mu, v = pi(x), vf(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        g.replay()
    torch.cuda.synchronize()
    print("graph us per (actor+critic):", 1e6 * (time.perf_counter() - t0) / 300)
