#!/usr/bin/env python3
"""Where a step of the persistent rollout kernel (K13) spends its time: HIP-event time of the launch at
N = 4096, T = 400 with phases switched off (OLY_K13_SKIP: bit 0 = no MFMA layers, bit 1 = no environment step:
outputs are garbage).  One process per variant (the knob is read once).
--stamps: bit 3 as well: the first forward wave and the first environment wave of workgroup 0 sum, per interval of the
step (the six barriers), the s_memtime ticks they worked and the ticks they then waited at the barrier, and the
environment wave the parts of its two longest intervals (buf.values is garbage afterwards)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, os.path.join(%r, "olympics-mujoco_amd")); sys.path.insert(0, os.path.join(%r, "tools"))
import numpy as np, torch
from bench_vecstep import persistent_kernel_us
from olympic_hip import specs
from olympic_hip.a3 import ReplayA3Physics, VecA3Env
from olympic_hip.engine import Engine
from olympic_hip.ppo import MLPCritic, MLPGaussianActor
from olympic_hip.synthetic import A3_FLOOR_BODY, A3_GEOM_BODYID, A3_LFOOT_BODY, A3_RFOOT_BODY, a3_synthetic_blocks
N, T = int(sys.argv[1]), int(sys.argv[2])
blocks = {k: torch.as_tensor(v).cuda() for k, v in a3_synthetic_blocks(N, 32, seed=1).items()}
env = VecA3Env(specs.A3Spec(mass=41.5), N, Engine(0), ReplayA3Physics(blocks), A3_GEOM_BODYID, A3_FLOOR_BODY,
               A3_RFOOT_BODY, A3_LFOOT_BODY, rs=np.random.RandomState(0))
torch.manual_seed(0)
pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
skip = int(os.environ.get("OLY_K13_SKIP", "0"))
env._dev_rollout = None
try:
    env.device_rollout(pi, vf, T, T, graph=False, persistent=True)
except Exception as e:
    print(json.dumps({"skip": skip, "error": repr(e)[:200]})); sys.exit(0)
res = {"skip": skip, "us_per_step": persistent_kernel_us(env, T, reps=8) / T}
if skip & 8:
    v = env._dev_rollout.buf.values.reshape(-1)[:36].double().cpu().numpy()
    tick_hz = v[:12].sum() / (v[12] / 1e8)          # s_memtime ticks per second (s_memrealtime runs at 100 MHz)
    names = ["1", "2", "3", "4", "5", "6"]
    res["s_memtime_hz"] = tick_hz
    res["forward_wave_us_per_step"] = {"work": dict(zip(names, (v[0:6] / T / tick_hz * 1e6).round(3).tolist())),
                                        "barrier_wait": dict(zip(names, (v[6:12] / T / tick_hz * 1e6).round(3).tolist()))}
    res["environment_wave_us_per_step"] = {"work": dict(zip(names, (v[16:22] / T / tick_hz * 1e6).round(3).tolist())),
                                            "barrier_wait": dict(zip(names, (v[22:28] / T / tick_hz * 1e6).round(3).tolist()))}
    sub = ["1_stage_readback", "1_contacts", "1_request_next", "6_assemble_row", "6_post_row_and_images", "6_rewards", "6_side_row", "6_reset_and_state"]
    res["environment_wave_parts_us_per_step"] = dict(zip(sub, (v[28:36] / T / tick_hz * 1e6).round(3).tolist()))
print(json.dumps(res))
''' % (ROOT, ROOT)


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    N = int(pos[0]) if len(pos) > 0 else 4096
    T = int(pos[1]) if len(pos) > 1 else 400
    out = {}
    variants = [(0, "everything", None), (1, "no_mfma_layers", None), (2, "no_environment_step", None), (3, "neither", None)]
    if "--stamps" in sys.argv:    # bit 3: per-interval work / barrier-wait times of workgroup 0 (s_memtime)
        variants = [(8, "everything", None), (10, "no_environment_step", None), (9, "no_mfma_layers", None)]
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    diag_lib = graft.build(diag=True)          # -DOLY_DIAG: the shipped library does not read OLY_K13_SKIP
    for skip, label, _ in variants:
        env = dict(os.environ, OLY_K13_SKIP=str(skip), OLYMPIC_HIP_LIB=diag_lib)
        r = subprocess.run([sys.executable, "-c", CHILD, str(N), str(T)], env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        out[label] = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
    print(json.dumps({"N": N, "T": T, **out}, indent=1))


if __name__ == "__main__":
    main()
