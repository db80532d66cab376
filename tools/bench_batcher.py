#!/usr/bin/env python3
"""End-to-end per-step regime THROUGH the host boundary (never the headline `value`):
action on device -> K5 -> D2H ctrl -> host physics threads (kinematic stand-in for mj_step)
-> H2D qpos/qvel -> K1, for N environments.  Prints env-steps/s and the time split."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import torch  # noqa: E402

from olympic_hip import specs  # noqa: E402
from olympic_hip.batcher import HostBatcher  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402
from olympic_hip.synthetic import h1_synthetic_block  # noqa: E402


def main():
    sp = specs.unitree_h1("walk")
    eng = Engine(0).il_configure(sp)
    out = {}
    for N in (4096, 32768):
        for th in (1, 16):
            qpos, qvel, act = h1_synthetic_block(sp, 1, N, seed=1)
            b = HostBatcher(eng, N, n_threads=th, dt=0.01)
            b.qpos[:], b.qvel[:] = qpos[0], qvel[0]
            a = torch.as_tensor(act[0]).cuda()
            for _ in range(20):
                b.step(a)
            torch.cuda.synchronize()
            K = 300
            t0 = time.perf_counter()
            acc = dict(ctrl_d2h_s=0.0, physics_s=0.0, h2d_enqueue_s=0.0)
            for _ in range(K):
                b.step(a)
                for k, v in b.last_timing().items():
                    acc[k] += v
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[f"N{N}_threads{th}"] = dict(us_per_vec_step=dt / K * 1e6, env_steps_per_s=N * K / dt,
                                            split_us={k: v / K * 1e6 for k, v in acc.items()},
                                            pcie_bytes_per_step=N * (8 * 11 + 8 * 34))
            b.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
