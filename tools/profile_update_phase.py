"""Where an update phase's wall clock goes (config 3: 1 638 400 rows, minibatches of 65 536, 3 epochs): KernelUpdate.step
alone (HIP events and wall), the permutation draws, the begin() pass.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "olympics-mujoco_amd")):
    sys.path.insert(0, p)
from olympic_hip._ffi import HipTimer  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402
from olympic_hip.ppo import KernelUpdate, MLPCritic, MLPGaussianActor  # noqa: E402


def main():
    import copy
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1638400
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    eng = Engine(0)
    torch.manual_seed(0)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    old = copy.deepcopy(pi)
    obs, act = torch.randn(n, 41, device="cuda"), torch.randn(n, 12, device="cuda") * 0.3
    ret, adv = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    ku = KernelUpdate(eng, pi, vf, old, 0.2, 0.5, 0.0)
    out = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ku.begin(obs)
    torch.cuda.synchronize()
    out["begin_first_call_ms"] = (time.perf_counter() - t0) * 1e3
    for name, mu in (("begin_ms", None), ("begin_with_the_rollouts_means_ms", ku.old_mu.clone())):
        ku.begin(obs, mu)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ku.begin(obs, mu)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) * 1e3
    ku.begin(obs)
    t0 = time.perf_counter()
    perm_host = torch.randperm(n)
    out["randperm_host_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    perm = perm_host.to(torch.int32).to("cuda")
    torch.cuda.synchronize()
    out["perm_to_device_ms"] = (time.perf_counter() - t0) * 1e3
    nb = n // B
    stats = torch.zeros((nb, 6), dtype=torch.float64, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize()
        tm = HipTimer()
        t0 = time.perf_counter()
        tm.start(eng._s())
        for b in range(nb):
            ku.step(obs, act, ret, adv, perm[b * B:(b + 1) * B], stats[b])
        t_enq = time.perf_counter() - t0
        tm.stop(eng._s())
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        out["epoch_%d" % rep] = dict(updates=nb, host_enqueue_ms_per_update=t_enq / nb * 1e3, wall_ms_per_update=wall / nb * 1e3,
                                     hip_event_ms_per_update=tm.elapsed_ms() / nb)
    for rep in range(2):                                  # the same epoch as ONE C call (oly_ppo_update_epoch)
        torch.cuda.synchronize()
        tm = HipTimer()
        t0 = time.perf_counter()
        tm.start(eng._s())
        ku.epoch(obs, act, ret, adv, perm, B, nb, stats)
        t_enq = time.perf_counter() - t0
        tm.stop(eng._s())
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        out["epoch_one_call_%d" % rep] = dict(updates=nb, host_enqueue_ms_per_update=t_enq / nb * 1e3,
                                              wall_ms_per_update=wall / nb * 1e3, hip_event_ms_per_update=tm.elapsed_ms() / nb)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
