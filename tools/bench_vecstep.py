#!/usr/bin/env python3
"""Config 3 sampling, per vec step: the device-resident rollout as (a) one K11 + one K10 launch per step, eager and
replayed from HIP graphs, and (b) ONE persistent launch per rollout (K13).  The persistent launch is also timed alone
with HIP events (kernel time of the T steps, without the reset launch / bootstrap pass / pool refill around it).
Prints one JSON object; run under `rocprofv3 --kernel-trace --stats` for the kernel durations."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from olympic_hip import specs  # noqa: E402
from olympic_hip.a3 import ReplayA3Physics, VecA3Env  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402
from olympic_hip.ppo import MLPCritic, MLPGaussianActor  # noqa: E402
from olympic_hip.synthetic import A3_FLOOR_BODY, A3_GEOM_BODYID, A3_LFOOT_BODY, A3_RFOOT_BODY, a3_synthetic_blocks  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--T", type=int, default=400)
    ap.add_argument("--K", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--graph-steps", type=int, default=8)
    ap.add_argument("--forward", default="hip", choices=["torch", "hip"])
    args = ap.parse_args()
    N, T = args.N, args.T
    blocks = {k: torch.as_tensor(v).cuda() for k, v in a3_synthetic_blocks(N, args.K, seed=1).items()}
    env = VecA3Env(specs.A3Spec(mass=41.5), N, Engine(0), ReplayA3Physics(blocks), A3_GEOM_BODYID, A3_FLOOR_BODY,
                   A3_RFOOT_BODY, A3_LFOOT_BODY, rs=np.random.RandomState(0))
    torch.manual_seed(0)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    kw = {}
    if args.forward == "torch":
        from olympic_hip.vecstep import TorchForward
        kw["forward"] = TorchForward(pi, vf)
    res = {}
    variants = [("eager", False, False), ("graph", True, False)]
    if args.forward == "hip":
        variants.append(("persistent", False, True))
    for label, graph, persistent in variants:
        for _ in range(2):
            env.device_rollout(pi, vf, T, T, graph=graph, graph_steps=args.graph_steps, persistent=persistent, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            env.device_rollout(pi, vf, T, T, graph=graph, graph_steps=args.graph_steps, persistent=persistent, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        res[label] = {"rollout_s": dt, "us_per_vec_step": 1e6 * dt / T, "env_steps_per_s": N * T / dt,
                      "info": env._dev_rollout.last_info}
    if args.forward == "hip":
        res["persistent"]["kernel_us_per_vec_step"] = persistent_kernel_us(env, T) / T
    print(json.dumps({"N": N, "T": T, "forward": args.forward, "graph_steps": args.graph_steps, **res}))


def persistent_kernel_us(env, T, reps=5):
    """HIP-event time of the K13 launch alone: rewind the step counters, launch, repeat."""
    r = env._dev_rollout
    fw = r._fw
    mu, v = fw.outputs(r.N)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(reps):
        r.traj_len.zero_()
        r.side_count.zero_()
        r.side_t.fill_(-1)
        r.pool_count.zero_()
        r.ctr[0::2] = 0
        a.record()
        r.launch.persistent(fw.packed_a, fw.norm_a, fw.packed_c, fw.norm_c, mu, v)
        b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b) * 1e3
    return tot / reps


if __name__ == "__main__":
    main()
