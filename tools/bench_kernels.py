#!/usr/bin/env python3
"""Per-kernel timing of the secondary kernels (K2, K3, K4, K6, K7, K8) at BASELINE sizes,
HIP events on the launch stream, algorithmic bytes per launch / time -> GB/s.
Prints one JSON object.  The headline kernel (K1) is measured by bench.py."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from olympic_hip import _abi, specs  # noqa: E402
from olympic_hip._ffi import HipTimer  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402


def timeit(eng, fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    t = HipTimer()
    s = eng.ctx.stream()
    t.start(s)
    for _ in range(reps):
        fn()
    t.stop(s)
    return t.elapsed_ms() / reps


def big(eng):
    """HBM-bound sizes (>= 0.5 GB of algorithmic traffic per launch): fraction of the 8 TB/s peak
    each secondary kernel reaches when launch latency and tail effects are negligible."""
    dev = eng.device
    g = torch.Generator(device="cuda").manual_seed(0)
    out = {}

    def rnd(shape, dt=torch.float32):
        return torch.empty(shape, dtype=dt, device=dev).normal_(0, 1, generator=g)

    def rec(name, ms, nbytes):
        out[name] = dict(ms=ms, GB=nbytes / 1e9, GBps=nbytes / ms / 1e6, frac_of_8TBps=nbytes / ms / 1e6 / 8000.0)
    T, N = 400, 262144
    r, v, nv = rnd((T, N)), rnd((T, N)), rnd((T, N))
    fl = ((torch.rand((T, N), device=dev, generator=g) < 1 / 300).to(torch.uint8) * 3)
    ret, adv = torch.empty_like(r), torch.empty_like(r)
    for mode, name, bpe in ((_abi.SCAN_RETURN, "K6_return_[400,262144]", 17), (_abi.SCAN_GAE, "K6_gae_[400,262144]", 21)):
        rec(name, timeit(eng, lambda: eng.return_scan(mode, 0.99, 0.97, r, v, nv, fl, ret, adv), reps=10), bpe * T * N)
    st = torch.empty(3, dtype=torch.float64, device=dev)
    rec("K6_return_fused_stats_[400,262144]",
        timeit(eng, lambda: eng.return_scan(_abi.SCAN_RETURN, 0.99, 0.97, r, v, nv, fl, ret, adv, stats3=st), reps=10),
        17 * T * N)
    r64 = r.double()
    rec("K6_return_f64_rewards_fused_stats_[400,262144]",
        timeit(eng, lambda: eng.return_scan(_abi.SCAN_RETURN, 0.99, 0.97, r64, v, nv, fl, ret, adv, stats3=st), reps=10),
        21 * T * N)
    del r64
    rec("K7_stats_[104857600]", timeit(eng, lambda: eng.adv_stats(adv, st), reps=10), 4 * T * N)
    rec("K7_normalize_[104857600]", timeit(eng, lambda: eng.adv_normalize(adv, st, 1, 1e-5), reps=10), 8 * T * N)
    x = adv.view(-1, 32)
    rec("K7_col_stats_[3276800,32]", timeit(eng, lambda: eng.col_stats(x), reps=10), 4 * x.numel())
    rec("K8_reward_[104857600]", timeit(eng, lambda: eng.disc_reward(adv.view(-1), ret.view(-1)), reps=10), 8 * T * N)
    mean, std = torch.zeros(32, dtype=torch.float64, device=dev), torch.ones(32, dtype=torch.float64, device=dev)
    o = ret.view(-1, 32)
    rec("K8_standardize_[3276800,32]", timeit(eng, lambda: eng.disc_standardize(x, None, mean, std, o), reps=10), 8 * x.numel())
    rec("obs_filter_[3276800,32]", timeit(eng, lambda: eng.obs_filter(x, mean, std, 1e-8, 10.0, out=o), reps=10), 8 * x.numel())
    del r, v, nv, fl, ret, adv, x, o
    # K9 at full-batch size
    Bp, A = 4 * 1638400, 12
    mu, omu, act = rnd((Bp, A)) * 0.3, rnd((Bp, A)) * 0.3, rnd((Bp, A)) * 0.3
    sd = torch.full((1,), 0.22, device=dev)
    av, rt, vl = rnd((Bp,)), rnd((Bp,)), rnd((Bp,))
    rec("K9_ppo_loss_[6553600,12]", timeit(eng, lambda: eng.ppo_loss(mu, sd, omu, sd, act, av, rt, vl, 0.2, 0.5), reps=10),
        (16 * A + 16) * Bp)
    del mu, omu, act, av, rt, vl
    # K3 / K2 at 1 Mi envs
    gb = np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32)
    eng.contact_configure(gb, 0, 7, 10)
    Nc, C = 1 << 20, 16
    ncon = torch.randint(0, 17, (Nc,), device=dev, generator=g, dtype=torch.int32)
    g1 = torch.zeros((Nc, C), dtype=torch.int32, device=dev)
    g2 = torch.randint(0, 13, (Nc, C), device=dev, generator=g, dtype=torch.int32)
    f6 = rnd((Nc, C, 6), torch.float64)
    pz = rnd((Nc, C), torch.float64)
    rec("K3_contacts_[1048576,16]", timeit(eng, lambda: eng.contact_reduce(ncon, g1, g2, f6, pz, want_idx=False), reps=10),
        (4 + C * 64 + 60) * Nc)
    eng.grf_configure(np.array([0, -1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 2, 2], np.int32), [(0, 1), (0, 2)])
    W = 4
    rec("K3_il_ground_forces_[4,262144,16]",
        timeit(eng, lambda: eng.il_ground_forces(ncon[:Nc].view(W, -1), g1.view(W, Nc // W, C), g2.view(W, Nc // W, C),
                                                 f6.view(W, Nc // W, C, 6), check=False), reps=10), (4 + C * 8 + 2 * 24) * Nc)   # ncon + geom pairs + the two selected force rows
    del f6, pz, g1, g2
    Wd, Nd, Kd = 10, 1 << 20, 6          # dense per-substep rows as the packed host batcher stages them (H1: 2 pairs x 3)
    steps = rnd((Wd, Nd, Kd), torch.float64)
    mean_o = torch.empty((Nd, Kd), dtype=torch.float64, device=dev)
    rec("K3_il_grf_window_dense_[10,1048576,6]", timeit(eng, lambda: eng.il_grf_window(steps, mean_o), reps=10),
        8 * Kd * (Wd + 1) * Nd)
    del steps, mean_o
    # K11 at a throughput size: 1 Mi rows through actor + critic (fp32 MFMA), reported as TFLOP/s too
    from olympic_hip.mlp import FusedMLPForward
    from olympic_hip.ppo import MLPCritic, MLPGaussianActor
    pi_, vf_ = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    fw = FusedMLPForward(eng, pi_, vf_)
    xm = rnd((1 << 20, 41))
    ms = timeit(eng, lambda: fw(xm), reps=10)
    flop = 2 * (1 << 20) * ((41 * 256 + 256 * 256 + 256 * 12) + (41 * 256 + 256 * 256 + 256))
    out["K11_mlp_forward2_[1048576,41]"] = dict(ms=ms, TFLOPs=flop / ms / 1e9, frac_of_f32_mfma_peak_157=flop / ms / 1e9 / 157.3,
                                                 GBps=(41 + 13) * 4 * (1 << 20) / ms / 1e6)
    del xm
    sp = specs.A3Spec(mass=41.5)
    eng.a3_configure(sp, np.zeros((4, sp.period)))
    Na = 1 << 20
    inp = dict(qpos=rnd((Na, 25), torch.float64), qvel=rnd((Na, 24), torch.float64),
               act_len=rnd((Na, 12), torch.float64), act_vel=rnd((Na, 12), torch.float64),
               lf_pos=rnd((Na, 3), torch.float64), rf_pos=rnd((Na, 3), torch.float64),
               lf_vel=rnd((Na, 3), torch.float64), rf_vel=rnd((Na, 3), torch.float64),
               root_pos=rnd((Na, 3), torch.float64), root_quat=rnd((Na, 4), torch.float64),
               head_pos=rnd((Na, 3), torch.float64), grf_l=rnd((Na,), torch.float64).abs() * 100,
               grf_r=rnd((Na,), torch.float64).abs() * 100, min_z=rnd((Na,), torch.float64) * 0.01,
               n_r=torch.ones(Na, dtype=torch.int32, device=dev), n_l=torch.ones(Na, dtype=torch.int32, device=dev),
               bad=torch.zeros(Na, dtype=torch.uint8, device=dev))
    z32 = lambda val=0: torch.full((Na,), val, dtype=torch.int32, device=dev)
    st2 = dict(phase=z32(), t1=z32(), t2=z32(1), reached_frames=z32(), target_reached=torch.zeros(Na, dtype=torch.uint8, device=dev),
               mode=z32(2), seq_len=z32(20), sequence=rnd((Na, 20, 4), torch.float64),
               goal=torch.zeros((Na, 8), dtype=torch.float64, device=dev))
    o2 = dict(obs=torch.empty((Na, 41), dtype=torch.float32, device=dev), rew6=torch.empty((Na, 6), dtype=torch.float32, device=dev),
              reward=torch.empty(Na, dtype=torch.float32, device=dev), done=torch.empty(Na, dtype=torch.uint8, device=dev))
    rec("K2_a3_step_[1048576]", timeit(eng, lambda: eng.a3_step(inp, st2, out=o2), reps=10), (930 + 257) * Na)
    print(json.dumps(out, indent=1))


def main():
    eng = Engine(0)
    if "--big" in sys.argv:
        return big(eng)
    dev = eng.device
    g = torch.Generator(device="cuda").manual_seed(0)
    out = {}
    T, N = 400, 4096

    def rnd(shape, dt=torch.float32):
        return torch.empty(shape, dtype=dt, device=dev).normal_(0, 1, generator=g)

    # K6
    r, v, nv = rnd((T, N)), rnd((T, N)), rnd((T, N))
    fl = ((torch.rand((T, N), device=dev, generator=g) < 1 / 300).to(torch.uint8) * 3)
    ret, adv = torch.empty_like(r), torch.empty_like(r)
    for mode, name, bpe in ((_abi.SCAN_RETURN, "K6_return", 17), (_abi.SCAN_GAE, "K6_gae", 21)):
        ms = timeit(eng, lambda: eng.return_scan(mode, 0.99, 0.97, r, v, nv, fl, ret, adv))
        out[name] = dict(ms=ms, GBps=bpe * T * N / ms / 1e6, bytes_per_elem=bpe, elems=T * N)
    # K7
    st = torch.empty(3, dtype=torch.float64, device=dev)
    ms = timeit(eng, lambda: eng.return_scan(_abi.SCAN_RETURN, 0.99, 0.97, r, v, nv, fl, ret, adv, stats3=st))
    out["K6_return_fused_stats"] = dict(ms=ms, GBps=17 * T * N / ms / 1e6, note="scan + finishing launch")
    r64 = r.double()
    ms = timeit(eng, lambda: eng.return_scan(_abi.SCAN_RETURN, 0.99, 0.97, r64, v, nv, fl, ret, adv, stats3=st))
    out["K6_return_f64_rewards_fused_stats"] = dict(ms=ms, GBps=21 * T * N / ms / 1e6)
    ms = timeit(eng, lambda: eng.adv_stats(adv, st))
    out["K7_stats"] = dict(ms=ms, GBps=4 * T * N / ms / 1e6)
    ms = timeit(eng, lambda: eng.adv_normalize(adv, st, 1, 1e-5))
    out["K7_normalize"] = dict(ms=ms, GBps=8 * T * N / ms / 1e6)
    x = rnd((T * N // 8, 32))
    ms = timeit(eng, lambda: eng.col_stats(x))
    out["K7_col_stats_[204800,32]"] = dict(ms=ms, GBps=x.numel() * 4 / ms / 1e6)
    # K8
    B = 4096 * 100
    d = rnd((B,))
    rw = torch.empty_like(d)
    ms = timeit(eng, lambda: eng.disc_reward(d, rw))
    out["K8_reward"] = dict(ms=ms, GBps=8 * B / ms / 1e6)
    xs = rnd((B, 32))
    mean, std = torch.zeros(32, dtype=torch.float64, device=dev), torch.ones(32, dtype=torch.float64, device=dev)
    o = torch.empty_like(xs)
    ms = timeit(eng, lambda: eng.disc_standardize(xs, None, mean, std, o))
    out["K8_standardize_[409600,32]"] = dict(ms=ms, GBps=8 * xs.numel() / ms / 1e6)
    # K9: PPO loss terms (forward + gradients in one pass) at full-batch and minibatch sizes
    for Bp in (T * N, 4096):
        A = 12
        mu, omu, act = rnd((Bp, A)) * 0.3, rnd((Bp, A)) * 0.3, rnd((Bp, A)) * 0.3
        sd = torch.full((1,), 0.22, device=dev)
        av, rt, vl = rnd((Bp,)), rnd((Bp,)), rnd((Bp,))
        ms = timeit(eng, lambda: eng.ppo_loss(mu, sd, omu, sd, act, av, rt, vl, 0.2, 0.5))
        out[f"K9_ppo_loss_[{Bp},12]"] = dict(ms=ms, GBps=(16 * A + 16) * Bp / ms / 1e6, bytes_per_row=16 * A + 16)
    src = torch.tensor([6, 7, 8, 9, 10, 11, 0, 1, 2, 3, 4, 5], dtype=torch.int32, device=dev)
    sg = torch.ones(12, device=dev)
    ms = timeit(eng, lambda: eng.mirror_loss(mu, omu, src, sg))
    out["K9_mirror_loss_[4096,12]"] = dict(ms=ms)
    xo = rnd((T * N // 8, 32))
    mean, var = torch.zeros(32, dtype=torch.float64, device=dev), torch.ones(32, dtype=torch.float64, device=dev)
    ms = timeit(eng, lambda: eng.obs_filter(xo, mean, var, 1e-8, 10.0, out=xo))
    out["obs_filter_[204800,32]"] = dict(ms=ms, GBps=8 * xo.numel() / ms / 1e6)
    # K3
    gb = np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32)
    eng.contact_configure(gb, 0, 7, 10)
    Nc, C = 4096 * 16, 16
    ncon = torch.randint(0, 9, (Nc,), device=dev, generator=g, dtype=torch.int32)
    g1 = torch.zeros((Nc, C), dtype=torch.int32, device=dev)
    g2 = torch.randint(0, 13, (Nc, C), device=dev, generator=g, dtype=torch.int32)
    f6 = rnd((Nc, C, 6), torch.float64)
    pz = rnd((Nc, C), torch.float64)
    ms = timeit(eng, lambda: eng.contact_reduce(ncon, g1, g2, f6, pz, want_idx=False))
    out["K3_contacts_[65536,16]"] = dict(ms=ms, GBps=(4 + C * 64 + 60) * Nc / ms / 1e6, env_per_s=Nc / ms * 1e3)
    # K2
    sp = specs.A3Spec(mass=41.5)
    lut = np.zeros((4, sp.period))
    eng.a3_configure(sp, lut)
    Na = 4096 * 16
    inp = dict(qpos=rnd((Na, 25), torch.float64), qvel=rnd((Na, 24), torch.float64),
               act_len=rnd((Na, 12), torch.float64), act_vel=rnd((Na, 12), torch.float64),
               lf_pos=rnd((Na, 3), torch.float64), rf_pos=rnd((Na, 3), torch.float64),
               lf_vel=rnd((Na, 3), torch.float64), rf_vel=rnd((Na, 3), torch.float64),
               root_pos=rnd((Na, 3), torch.float64), root_quat=rnd((Na, 4), torch.float64),
               head_pos=rnd((Na, 3), torch.float64), grf_l=rnd((Na,), torch.float64).abs() * 100,
               grf_r=rnd((Na,), torch.float64).abs() * 100, min_z=rnd((Na,), torch.float64) * 0.01,
               n_r=torch.ones(Na, dtype=torch.int32, device=dev), n_l=torch.ones(Na, dtype=torch.int32, device=dev),
               bad=torch.zeros(Na, dtype=torch.uint8, device=dev))
    st2 = dict(phase=torch.zeros(Na, dtype=torch.int32, device=dev), t1=torch.zeros(Na, dtype=torch.int32, device=dev),
               t2=torch.ones(Na, dtype=torch.int32, device=dev), reached_frames=torch.zeros(Na, dtype=torch.int32, device=dev),
               target_reached=torch.zeros(Na, dtype=torch.uint8, device=dev),
               mode=torch.full((Na,), 2, dtype=torch.int32, device=dev),
               seq_len=torch.full((Na,), 20, dtype=torch.int32, device=dev),
               sequence=rnd((Na, 20, 4), torch.float64), goal=torch.zeros((Na, 8), dtype=torch.float64, device=dev))
    o2 = dict(obs=torch.empty((Na, 41), dtype=torch.float32, device=dev),
              rew6=torch.empty((Na, 6), dtype=torch.float32, device=dev),
              reward=torch.empty(Na, dtype=torch.float32, device=dev), done=torch.empty(Na, dtype=torch.uint8, device=dev))
    ms = timeit(eng, lambda: eng.a3_step(inp, st2, out=o2))
    out["K2_a3_step_[65536]"] = dict(ms=ms, GBps=(930 + 257) * Na / ms / 1e6, env_per_s=Na / ms * 1e3)
    # K1 single-step regime (T = 1): launch-latency-bound, reported as latency
    h1 = specs.unitree_h1("walk")
    eng.il_configure(h1)
    q1, v1 = rnd((1, N, 17), torch.float64), rnd((1, N, 17), torch.float64)
    a1 = rnd((1, N, 11))
    pv = torch.zeros(N, dtype=torch.float64, device=dev)
    o1 = dict(obs=torch.empty((1, N, 32), dtype=torch.float32, device=dev), reward=torch.empty((1, N), dtype=torch.float32, device=dev),
              absorbing=torch.empty((1, N), dtype=torch.uint8, device=dev), ctrl=torch.empty((1, N, 11), dtype=torch.float32, device=dev))
    ms = timeit(eng, lambda: eng.il_step(q1, v1, a1, pv, out=o1, want_fall_code=False), reps=200, warm=20)
    out["K1_single_step_T1_N4096"] = dict(us_per_launch=ms * 1e3, env_steps_per_s=N / ms * 1e3,
                                          note="launch-bound regime, 2 MB per launch")
    call, _ = eng.il_step_prepare(q1, v1, a1, pv, out=o1, want_fall_code=False)
    stream = eng.ctx.stream()
    ms = timeit(eng, lambda: call(stream), reps=500, warm=50)
    out["K1_single_step_T1_N4096_prepared"] = dict(us_per_launch=ms * 1e3, env_steps_per_s=N / ms * 1e3,
                                                   note="same launch through the prepared-call path")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
