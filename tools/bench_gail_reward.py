#!/usr/bin/env python3
"""Config 4: the VAIL discriminator reward + GAE(0.97) + biased-std normalisation pipeline on
UnitreeH1-shaped observations (mask + running standardisation on the device, encoder / decoder
GEMMs in PyTorch-ROCm, reparameterisation and reward epilogue as HIP kernels, K6, K7).
Wall clock per call with a device sync on both sides.  Prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from olympic_hip.engine import Engine  # noqa: E402
from olympic_hip.gail import DiscriminatorReward, GAILAdvantage, VariationalDiscriminator  # noqa: E402
from olympic_hip.ppo import MLPCritic  # noqa: E402


def main():
    eng = Engine(0)
    torch.manual_seed(0)
    net = VariationalDiscriminator(32).cuda()
    disc = DiscriminatorReward(eng, net, state_mask=np.arange(32))
    critic = MLPCritic(32).cuda()
    out = {}
    for T, N in ((1, 4096), (400, 4096)):
        x = torch.randn((T, N, 32), device="cuda")
        xn = torch.randn((T, N, 32), device="cuda")
        r_env = torch.zeros((T, N), device="cuda")
        ab = torch.rand((T, N), device="cuda") < 0.003
        last = ab | (torch.rand((T, N), device="cuda") < 0.003)
        eps = torch.randn((T * N, 128), device="cuda")
        flat = x.reshape(T * N, 32).contiguous()

        def timeit(fn, reps):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        adv = GAILAdvantage(eng, disc, critic, gamma=0.99, lam=0.97)
        reps = 200 if T == 1 else 10
        out[f"[{T},{N}]"] = dict(discriminator_reward_ms=timeit(lambda: disc(flat, eps), reps),
                                 reward_gae_normalise_ms=timeit(lambda: adv(x, xn, r_env, ab, last, eps), reps),
                                 samples=T * N)
        out[f"[{T},{N}]"]["samples_per_s_full_pipeline"] = T * N / out[f"[{T},{N}]"]["reward_gae_normalise_ms"] * 1e3
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
