"""Config 4 (BASELINE.json): the VAIL discriminator reward through K12 (oly_col_stats + oly_disc_forward),
timed with HIP events on the launch stream, against the layer-by-layer path it replaces.

  python tools/bench_disc_forward.py [--rows 4096 1638400] [--iters 50] [--json out.json]

FLOP per sample (algorithmic, multiply + add): 2 (D 256 + 256 128 + 128 256 + 128) = 147 712 at D = 32.
Peak: 157.3 TFLOP/s dense f32 MFMA (MI355X_MICROARCH.md)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
from olympic_hip.engine import Engine                                  # noqa: E402
from olympic_hip.gail import DiscriminatorReward, VariationalDiscriminator   # noqa: E402

PEAK_F32_MFMA = 157.3e12


def flop_per_sample(D=32):
    return 2 * (D * 256 + 256 * 128 + 128 * 256 + 128)


def timed(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[4096, 400 * 4096])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    eng = Engine(0)
    torch.manual_seed(0)
    net = VariationalDiscriminator().cuda()
    out = []
    for B in a.rows:
        g = torch.Generator(device="cuda").manual_seed(7)
        x = torch.randn((B, 32), device="cuda", generator=g)
        eps = torch.randn((B, 128), device="cuda", generator=g)
        dr = DiscriminatorReward(eng, net, state_mask=np.arange(32))
        packed = dr.packed()
        cs = eng.col_stats(x)
        bufs = dict(reward=torch.empty(B, device="cuda"))
        iters = a.iters if B <= 65536 else max(5, a.iters // 5)
        t_kernel = timed(lambda: eng.disc_forward(x, packed, colstats=cs, eps=eps, out=bufs), iters)
        t_fused = timed(lambda: dr.forward(x, eps, out=bufs), iters)
        t_plain = timed(lambda: eng.disc_reward(dr.logits_unfused(x, eps)[0]), iters)
        fl = flop_per_sample() * B
        rec = dict(rows=B, disc_forward_us=round(t_kernel, 2), stats_plus_forward_us=round(t_fused, 2),
                   layer_by_layer_us=round(t_plain, 2), tflops=round(fl / t_kernel / 1e6, 2),
                   frac_of_f32_mfma_peak=round(fl / (t_kernel * 1e-6) / PEAK_F32_MFMA, 4),
                   samples_per_s=round(B / (t_fused * 1e-6), 1), hbm_bytes_per_sample=32 * 4 + 128 * 4 + 4)
        print(json.dumps(rec), flush=True)
        out.append(rec)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
