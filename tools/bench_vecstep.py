#!/usr/bin/env python3
"""Config 3 sampling, per vec step: the device-resident rollout (one fused K10 launch per step + the
policy / critic forward), eager and replayed from HIP graphs.  Prints one JSON object; run under
`rocprofv3 --kernel-trace --stats` for the kernel durations."""
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
    for label, graph in (("eager", False), ("graph", True)):
        for _ in range(2):
            env.device_rollout(pi, vf, T, T, graph=graph, graph_steps=args.graph_steps, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            env.device_rollout(pi, vf, T, T, graph=graph, graph_steps=args.graph_steps, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        res[label] = {"rollout_s": dt, "us_per_vec_step": 1e6 * dt / T, "env_steps_per_s": N * T / dt,
                      "info": env._dev_rollout.last_info}
    print(json.dumps({"N": N, "T": T, "forward": args.forward, "graph_steps": args.graph_steps, **res}))


if __name__ == "__main__":
    main()
