#!/usr/bin/env python3
"""K2 (oly_a3_step): one lane per environment against sixteen, over N.  Checks the two kernels give identical bytes on
the same inputs, then times each with HIP events (back-to-back launches on the engine's stream).  One JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from olympic_hip import specs  # noqa: E402
from olympic_hip._ffi import HipTimer  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402


def make(N, seed, dev="cuda"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    rnd = lambda shape, dt=torch.float64: torch.randn(shape, generator=g, dtype=dt).to(dev)
    inp = dict(qpos=rnd((N, 25)), qvel=rnd((N, 24)), act_len=rnd((N, 12)), act_vel=rnd((N, 12)),
               lf_pos=rnd((N, 3)) * 0.3, rf_pos=rnd((N, 3)) * 0.3, lf_vel=rnd((N, 3)) * 0.2, rf_vel=rnd((N, 3)) * 0.2,
               root_pos=rnd((N, 3)) * 0.3 + torch.tensor([0.0, 0.0, 0.8], dtype=torch.float64, device=dev),
               root_quat=rnd((N, 4)), head_pos=rnd((N, 3)) * 0.3,
               grf_l=rnd((N,)).abs() * 100, grf_r=rnd((N,)).abs() * 100, min_z=rnd((N,)) * 0.01,
               n_r=torch.randint(0, 3, (N,), generator=g, dtype=torch.int32).to(dev),
               n_l=torch.randint(0, 3, (N,), generator=g, dtype=torch.int32).to(dev),
               bad=(torch.rand((N,), generator=g) < 0.02).to(torch.uint8).to(dev))
    ri = lambda lo, hi: torch.randint(lo, hi, (N,), generator=g, dtype=torch.int32).to(dev)
    st = dict(phase=ri(0, 60), t1=ri(0, 19), t2=ri(0, 20), reached_frames=ri(0, 3),
              target_reached=torch.zeros(N, dtype=torch.uint8, device=dev), mode=ri(0, 4), seq_len=ri(1, 21),
              sequence=rnd((N, 20, 4)) * 0.3, goal=torch.zeros((N, 8), dtype=torch.float64, device=dev))
    return inp, st


def outs(N, dt, dev="cuda"):
    return dict(obs=torch.empty((N, 41), dtype=dt, device=dev), rew6=torch.empty((N, 6), dtype=torch.float32, device=dev),
                reward=torch.empty(N, dtype=torch.float32, device=dev), done=torch.empty(N, dtype=torch.uint8, device=dev))


def run(eng, lanes, inp, st, o, reps=0):
    os.environ["OLY_K2_LANES"] = str(lanes)
    s = {k: v.clone() for k, v in st.items()}
    eng.a3_step(inp, s, obs_f64=o["obs"].dtype == torch.float64, out=o)
    torch.cuda.synchronize()
    res = ({k: v.clone() for k, v in o.items()}, {k: v.clone() for k, v in s.items()})
    if not reps:
        return res, None
    t = HipTimer()
    stream = eng.ctx.stream()
    for _ in range(20):
        eng.a3_step(inp, s, obs_f64=o["obs"].dtype == torch.float64, out=o)
    t.start(stream)
    for _ in range(reps):
        eng.a3_step(inp, s, obs_f64=o["obs"].dtype == torch.float64, out=o)
    t.stop(stream)
    return res, 1e3 * t.elapsed_ms() / reps


def main():
    eng = Engine(0)
    sp = specs.A3Spec(mass=41.5)
    rs = np.random.RandomState(3)
    eng.a3_configure(sp, rs.uniform(-1, 1, (4, sp.period)))
    out = {}
    for N in (1, 17, 1000, 4096, 16384, 65536, 262144, 1048576):
        inp, st = make(N, seed=N)
        row = {}
        for dt in (torch.float32, torch.float64):
            o = outs(N, dt)
            (a, sa), ta = run(eng, 1, inp, st, o, reps=200 if N <= 65536 else 30)
            (b, sb), tb = run(eng, 16, inp, st, o, reps=200 if N <= 65536 else 30)
            same = all(torch.equal(a[k].view(torch.uint8), b[k].view(torch.uint8)) for k in a) and \
                all(torch.equal(sa[k].view(torch.uint8), sb[k].view(torch.uint8)) for k in sa)
            row["f32" if dt == torch.float32 else "f64"] = dict(lane1_us=ta, lane16_us=tb, identical=bool(same))
        out[str(N)] = row
    os.environ.pop("OLY_K2_LANES", None)
    print(json.dumps(out, indent=1))
    assert all(v["identical"] for r in out.values() for v in r.values())


if __name__ == "__main__":
    main()
