#!/usr/bin/env python3
"""Host-side time per section of one vectorised rollout step (config 3, synthetic readback):
where the ~0.6 ms per vec step goes.  Prints one JSON object (microseconds per step)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench_ppo_iter import synthetic_blocks  # noqa: E402
from olympic_hip import _abi, specs  # noqa: E402
from olympic_hip.a3 import ReplayA3Physics, VecA3Env  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402
from olympic_hip.ppo import MLPCritic, MLPGaussianActor  # noqa: E402


@torch.no_grad()
def main():
    N, K = 4096, 200
    gen = torch.Generator(device="cuda").manual_seed(1)
    sp = specs.A3Spec(mass=41.5)
    env = VecA3Env(sp, N, Engine(0), ReplayA3Physics(synthetic_blocks(N, 32, gen)),
                   np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32), 0, 7, 10)
    env.state["seq_len"].fill_(20)
    env.state["mode"].fill_(_abi.MODE_FORWARD)
    env.state["t2"].fill_(1)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    state = torch.zeros((N, 41), device="cuda")
    acc = dict(policy=0.0, critic=0.0, env_step=0.0, pd_target=0.0, physics=0.0, contact_reduce=0.0, a3_step=0.0)

    def timed(key, fn, sync):
        t0 = time.perf_counter()
        out = fn()
        if sync:
            torch.cuda.synchronize()
        acc[key] += time.perf_counter() - t0
        return out
    res = {}
    for sync in (False, True):
        for k in acc:
            acc[k] = 0.0
        for it in range(K + 20):
            if it == 20:
                torch.cuda.synchronize()
                for k in acc:
                    acc[k] = 0.0
                t_all = time.perf_counter()
            a = timed("policy", lambda: pi(state, deterministic=False), sync)
            timed("critic", lambda: vf(state), sync)
            t0 = time.perf_counter()
            tgt = timed("pd_target", lambda: env.eng.a3_pd_target(a.contiguous()), sync)
            inp = timed("physics", lambda: env.physics.step(tgt), sync)
            cr = timed("contact_reduce", lambda: env.eng.contact_reduce(inp["ncon"], inp["geom1"], inp["geom2"], inp["force6"],
                                                                        inp["cpos_z"], want_idx=False), sync)
            kin = {k: inp[k] for k in ("qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel", "rf_vel",
                                       "root_pos", "root_quat", "head_pos")}
            kin.update(grf_l=cr["grf_l"], grf_r=cr["grf_r"], min_z=cr["min_z"], n_r=cr["n_r"], n_l=cr["n_l"], bad=cr["bad"])
            o = timed("a3_step", lambda: env.eng.a3_step(kin, env.state), sync)
            acc["env_step"] += time.perf_counter() - t0
            state = o["obs"]
        torch.cuda.synchronize()
        total = (time.perf_counter() - t_all) / K * 1e6
        res["device_synced_each_call" if sync else "async_host_time"] = dict(
            total_us=total, **{k: v / K * 1e6 for k, v in acc.items()})
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
