#!/usr/bin/env python3
"""What a config-3 rollout costs besides its one K13 launch: run a few persistent rollouts back to back; under
`rocprofv3 --kernel-trace` + tools/trace_summary.py this lists every other kernel (reset launch, noise block, weight
re-pack, bootstrap critic pass, pool refill, small torch ops).  Prints the wall time per rollout."""
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

N, T, reps = 4096, 400, int(sys.argv[1]) if len(sys.argv) > 1 else 5
blocks = {k: torch.as_tensor(v).cuda() for k, v in a3_synthetic_blocks(N, 32, seed=1).items()}
env = VecA3Env(specs.A3Spec(mass=41.5), N, Engine(0), ReplayA3Physics(blocks), A3_GEOM_BODYID, A3_FLOOR_BODY,
               A3_RFOOT_BODY, A3_LFOOT_BODY, rs=np.random.RandomState(0))
torch.manual_seed(0)
pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
for _ in range(2):
    env.device_rollout(pi, vf, T, T, graph=False, persistent=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    env.device_rollout(pi, vf, T, T, graph=False, persistent=True)
torch.cuda.synchronize()
print(json.dumps({"rollout_ms": 1e3 * (time.perf_counter() - t0) / reps, "reps": reps + 2}))
