"""Where a K14 tile's cycles go: a -DOLY_DIAG build stamps s_memtime at every phase boundary (workgroup 0 = actor part 0,
first critic workgroup; every wave; the first 64 work items).  Prints per-phase mean cycles: work (stamp -> barrier
arrival) and barrier wait (arrival -> release) for wave 0 (runs the loss) and wave 5, plus the kernel's wall time.

    python tools/time_k14.py [B] [--mirror]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "olympics-mujoco_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import __graft_entry__ as graft  # noqa: E402

os.environ["OLYMPIC_HIP_LIB"] = graft.build(diag=True)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from helpers import ppo_update_case  # noqa: E402
from olympic_hip._ffi import HipTimer, lib  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402

PHASES = ["L1", "bar1", "L2", "bar2", "L3", "bar3", "loss", "bar4", "dH2+dW3+dW2", "bar5", "dH1+dW1", "stage+preload", "bar6"]


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(pos[0]) if pos else 65536
    mirror = "--mirror" in sys.argv
    eng = Engine(0)
    c = ppo_update_case(0, n=4096, mirror=mirror)
    d = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()
    reps = max(1, B // 4096)
    big = lambda x: d(x).repeat(*([reps] + [1] * (x.ndim - 1)))
    obs, act, adv, ret, omu = big(c["obs"]), big(c["action"]), big(c["adv"]), big(c["ret"]), big(c["old_mu"])
    mir = big(c["mir_obs"]) if mirror else None
    n = obs.shape[0]
    pa = eng.mlp_pack(*[d(x) for x in c["actor"]], d(c["a_mean"]), d(c["a_std"]))
    pc = eng.mlp_pack(*[d(x) for x in c["critic"]])
    ga = torch.empty(int(lib().oly_ppo_update_grad_floats(41, 256, 12)), device="cuda")
    gc = torch.empty(int(lib().oly_ppo_update_grad_floats(41, 256, 1)), device="cuda")
    scal = torch.zeros(6, dtype=torch.float64, device="cuda")
    sd, lsd = d(c["sd"]), d(c["log_sd"])
    kw = dict(mir_obs=mir, act_src=d(c["act_src"]), act_sign=d(c["act_sign"])) if mirror else {}
    Bq = min(B, n)
    ws_n, p_a, p_c = eng.ppo_update_plan(Bq, 41, 12, mirror)
    slots = 2 * 8 * 64 * 16
    ws = torch.zeros(ws_n + 2 * slots + 8, device="cuda")
    perm = torch.randperm(n, device="cuda")[:Bq].to(torch.int32)
    run = lambda: eng.ppo_update_grads(obs, act, adv, ret, omu, pa, pc, sd, lsd, sd, lsd, ga, gc, scal, ws, idx=perm,
                                       normalize_actor=True, mirror_coeff=0.4, **kw)
    x = torch.randn(32 << 20, device="cuda")
    for _ in range(20):
        x.mul_(1.0001)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    t = HipTimer()
    t.start(eng._s())
    for _ in range(10):
        run()
    t.stop(eng._s())
    ms = t.elapsed_ms() / 10
    torch.cuda.synchronize()
    off = (ws.numel() - 2 * slots) & ~1
    st = ws[off:off + 2 * slots].view(torch.int64).cpu().numpy().reshape(2, 8, 64, 16)
    out = dict(B=Bq, mirror=mirror, parts=(p_a, p_c), kernel_pair_ms=ms)
    items = min(64, (-(-Bq // 16) + p_a - 1) // p_a * (3 if mirror else 1))
    for net, name in ((0, "actor"), (1, "critic")):
        for w in (0, 5):
            s = st[net, w, 1:max(2, items - 1), :13].astype(np.float64)       # skip the first and the last item
            s = s[(s > 0).all(1)]
            if len(s) == 0:
                continue
            dur = np.diff(s, axis=1)                     # 12 intervals between the 13 stamps
            mean = dur.mean(0)
            rec = dict(zip(PHASES[:12], mean.round(0).tolist()), item_cycles=float((s[1:, 0] - s[:-1, 0]).mean()) if len(s) > 1 else None)
            s13 = st[net, w, 1:max(2, items - 1), 13].astype(np.float64)[:len(s)]
            if (s13 > 0).all():               # stamp 13: between the loss and the deferred dW2 (inside the "loss" interval)
                rec["loss_only"] = float((s13 - s[:, 6]).mean())
                rec["deferred_dW2"] = float((s[:, 7] - s13).mean())
            out[f"{name}_wave{w}"] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
