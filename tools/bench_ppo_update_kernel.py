"""K14 alone: oly_ppo_update_grads on minibatches gathered from a config-3 sized buffer, HIP events on the launch stream.

    python tools/bench_ppo_update_kernel.py [--rows 1638400] [--batches 64,4096,65536] [--mirror] [--out FILE]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "olympics-mujoco_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

FLOP_FWD_BWD = lambda i, a: 2 * (i * 256 + 65536 + 256 * a) + 2 * (2 * 256 * a + 2 * 65536 + 256 * i)   # noqa: E731


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1638400)
    ap.add_argument("--batches", default="64,1024,4096,16384,65536")
    ap.add_argument("--mirror", action="store_true")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from helpers import ppo_update_case
    from olympic_hip._ffi import HipTimer, lib
    from olympic_hip.engine import Engine
    eng = Engine(0)
    c = ppo_update_case(0, n=4096, mirror=a.mirror)
    reps_rows = a.rows // 4096
    d = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()
    big = lambda x: d(x).repeat(*([reps_rows] + [1] * (x.ndim - 1)))
    obs, act, adv, ret, omu = big(c["obs"]), big(c["action"]), big(c["adv"]), big(c["ret"]), big(c["old_mu"])
    mir = big(c["mir_obs"]) if a.mirror else None
    pa = eng.mlp_pack(*[d(x) for x in c["actor"]], d(c["a_mean"]), d(c["a_std"]))
    pc = eng.mlp_pack(*[d(x) for x in c["critic"]])
    ga = torch.empty(int(lib().oly_ppo_update_grad_floats(41, 256, 12)), device="cuda")
    gc = torch.empty(int(lib().oly_ppo_update_grad_floats(41, 256, 1)), device="cuda")
    scal = torch.zeros(6, dtype=torch.float64, device="cuda")
    sd, lsd = d(c["sd"]), d(c["log_sd"])
    kw = dict(mir_obs=mir, act_src=d(c["act_src"]), act_sign=d(c["act_sign"])) if a.mirror else {}
    n = obs.shape[0]
    out = dict(rows=n, mirror=a.mirror, results=[])
    x = torch.randn(64 << 20, device="cuda")
    for _ in range(30):                                     # wake the device (profiles/r03/clock_ramp.json)
        x.mul_(1.0001)
    for B in [int(b) for b in a.batches.split(",")]:
        ws_n, p_a, p_c = eng.ppo_update_plan(B, 41, 12, a.mirror)
        ws = torch.empty(ws_n, device="cuda")
        perm = torch.randperm(n, device="cuda")[:B].to(torch.int32)
        run = lambda: eng.ppo_update_grads(obs, act, adv, ret, omu, pa, pc, sd, lsd, sd, lsd, ga, gc, scal, ws, idx=perm,
                                           normalize_actor=True, mirror_coeff=0.4, **kw)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        t = HipTimer()
        s = eng._s()
        t.start(s)
        for _ in range(a.reps):
            run()
        t.stop(s)
        ms = t.elapsed_ms() / a.reps
        flop = B * (FLOP_FWD_BWD(41, 12) * (2 if a.mirror else 1) + FLOP_FWD_BWD(41, 1))   # the recomputed forward is not credited
        rec = dict(B=B, parts=(p_a, p_c), ms=ms, tflops=flop / ms / 1e9, frac_of_f32_mfma_peak=flop / ms / 1e9 / 157.3)
        out["results"].append(rec)
        print(json.dumps(rec), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
