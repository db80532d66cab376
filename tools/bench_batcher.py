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
    for N, th, mapped in ((4096, 1, 0), (4096, 16, 0), (32768, 1, 0), (32768, 16, 0), (4096, 16, 1), (4096, 16, 2), (4096, 16, 3),
                          (32768, 16, 1), (32768, 16, 3)):
        if True:
            qpos, qvel, act = h1_synthetic_block(sp, 1, N, seed=1)
            b = HostBatcher(eng, N, n_threads=th, dt=0.01).set_mapped(int(mapped))
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
            out[f"N{N}_threads{th}" + (f"_mapped{mapped}" if mapped else "")] = dict(us_per_vec_step=dt / K * 1e6, env_steps_per_s=N * K / dt,
                                            split_us={k: v / K * 1e6 for k, v in acc.items()},
                                            pcie_bytes_per_step=N * (8 * 11 + 8 * 34))
            b.close()
    # RL robot: PD target down, the whole StickFigureA3 readback (1788 B per env at C = 16) up, K3 + K2
    import numpy as np
    from olympic_hip.batcher import A3HostBatcher
    a3 = specs.A3Spec(mass=41.5)
    eng.a3_configure(a3, np.zeros((4, a3.period)))
    eng.contact_configure(np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32), 0, 7, 10)
    for N, mapped, compact in ((4096, 0, 0), (4096, 1, 0), (4096, 1, 1), (32768, 0, 0), (32768, 1, 1)):
        b = A3HostBatcher(eng, N, 16, None, n_threads=16).set_mapped(mapped).set_compact(compact)
        rng = np.random.default_rng(0)
        nc_all = np.minimum(rng.poisson(4, N), 16)                 # config 3: ncon ~ Poisson(4) clipped to the slots
        for e in range(N):
            sl = b.slots(e)
            sl["root_quat"][:] = [1, 0, 0, 0]
            k_ = int(nc_all[e])
            sl["ncon"][0] = k_
            sl["geom2"][:k_] = 7 + 3 * (np.arange(k_) % 2)            # right / left foot alternately
            sl["force6"][:k_] = rng.normal(0, 100, (k_, 6))
        z = lambda dt, *sh: torch.zeros((N,) + sh, dtype=dt, device="cuda")
        st = dict(phase=z(torch.int32), t1=z(torch.int32), t2=z(torch.int32) + 1, reached_frames=z(torch.int32),
                  target_reached=z(torch.uint8), mode=z(torch.int32) + 2, seq_len=z(torch.int32) + 20,
                  sequence=z(torch.float64, 20, 4), goal=z(torch.float64, 8))
        a = torch.zeros((N, 12), device="cuda")
        for _ in range(20):
            b.step(a, st)
        torch.cuda.synchronize()
        K = 300
        t0 = time.perf_counter()
        acc = dict(target_d2h_s=0.0, physics_s=0.0, h2d_kernels_enqueue_s=0.0)
        for _ in range(K):
            b.step(a, st)
            for k, v in b.last_timing().items():
                acc[k] += v
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"A3_N{N}_threads16" + ("_mapped" if mapped else "") + ("_compact" if compact else "")] = dict(us_per_vec_step=dt / K * 1e6, env_steps_per_s=N * K / dt,
                                         split_us={k: v / K * 1e6 for k, v in acc.items()},
                                         pcie_bytes_per_step=N * (8 * 12 + 1788))
        b.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
