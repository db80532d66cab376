#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the post-physics hot path for 4096 UnitreeH1.walk
environments per GPU (BASELINE.json configs[1]).

One "step" = one launch of the fused K1+K5 kernel (obs build + has-fallen + previous-obs
reward + action scale/clamp) over one [T=400, N=4096] block of synthetic qpos/qvel/action
already resident in HBM = 1 638 400 env-steps.  Launch regime: [T,N] block per launch (a
single vec step at N=4096 moves 2 MB and is launch-latency-bound).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Environments shard across ranks with no data-path
collective (weak scaling: every rank owns its own 4096 envs); the barriers bracketing the
timed region are the only communication.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "olympics-mujoco_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def alg_bytes_per_row(spec, fall_code):
    """Algorithmic HBM bytes per (step, env) row of K1+K5, each array counted once:
    qpos + qvel (f64) + action (f32) in; obs (f32) + reward (f32) + absorbing (u8) + ctrl (f32)
    out.  H1: 272 + 44 + 128 + 4 + 1 + 44 = 493 B (+1 with fall codes).  The carried reward
    state (16 B per env per LAUNCH) is not counted."""
    return (8 * (spec.nq + spec.nv + spec.n_grf) + 4 * spec.n_act + 4 * spec.n_obs + 4 + 1 + 4 * spec.nu
            + (1 if fall_code else 0))


def cpu_baseline(spec, seconds=12.0):
    """The CPU oracle (C port of the reference path, parity-pinned to golden vectors) on a
    bounded sample of the same workload, all host cores."""
    from oracle import oracle as orc
    from olympic_hip.synthetic import h1_synthetic_block
    T, N = 50, 4096
    qpos, qvel, act = h1_synthetic_block(spec, T, N, seed=1234)
    out = dict(obs=np.empty((T, N, spec.n_obs), np.float32), reward=np.empty((T, N), np.float32),
               absorbing=np.empty((T, N), np.uint8), ctrl=np.empty((T, N, spec.nu), np.float32))
    prev = np.full(N, 1.25)
    threads = orc.max_threads()
    res = {}
    for label, th in (("mt", threads), ("1t", 1)):
        orc.il_step_mt(spec, qpos, qvel, act, prev, out, threads=th)      # warm
        t0 = time.perf_counter()
        reps = 0
        budget = seconds * (0.7 if label == "mt" else 0.3)
        while time.perf_counter() - t0 < budget:
            orc.il_step_mt(spec, qpos, qvel, act, prev, out, threads=th)
            reps += 1
        dt = time.perf_counter() - t0
        res[label] = reps * T * N / dt
    return {"value": res["mt"], "unit": "env-steps/s", "cores": threads, "kind": "port",
            "value_1thread": res["1t"],
            "sample": f"oracle oly_il_step_cpu_mt on a [T={T},N={N}] slice of the same synthetic "
                      f"H1 block, repeated for ~{seconds:.0f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--T", type=int, default=400)
    ap.add_argument("--N", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fall-code", action="store_true", help="also write the fall-code byte")
    ap.add_argument("--robot", default="h1", choices=["h1", "atlas", "talos", "h1_arms", "h1_ff"],
                    help="h1 is the BASELINE config; the others exercise the same kernel on other tables")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL over xGMI; gloo only to "
                         "rehearse the multi-rank path on a one-GPU box)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    else:
        dist = None
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from olympic_hip import specs
    from olympic_hip._ffi import HipTimer
    from olympic_hip.engine import Engine
    from olympic_hip.synthetic import h1_synthetic_block

    spec = {"h1": lambda: specs.unitree_h1("walk"), "atlas": lambda: specs.atlas("walk"),
            "talos": lambda: specs.talos("walk"),
            "h1_arms": lambda: specs.unitree_h1("walk", disable_arms=False),
            "h1_ff": lambda: specs.unitree_h1("walk").with_foot_forces("UnitreeH1")}[args.robot]()
    eng = Engine(local_rank).il_configure(spec)
    T, N = args.T, args.N
    qpos_h, qvel_h, act_h = h1_synthetic_block(spec, T, N, seed=1234 + 17 * rank)
    qpos = torch.as_tensor(qpos_h).to(dev)
    qvel = torch.as_tensor(qvel_h).to(dev)
    act = torch.as_tensor(act_h).to(dev)
    del qpos_h, qvel_h, act_h
    prev = [torch.full((N,), 1.25, dtype=torch.float64, device=dev), torch.empty(N, dtype=torch.float64, device=dev)]
    out = dict(obs=torch.empty((T, N, spec.n_obs), dtype=torch.float32, device=dev),
               reward=torch.empty((T, N), dtype=torch.float32, device=dev),
               absorbing=torch.empty((T, N), dtype=torch.uint8, device=dev),
               ctrl=torch.empty((T, N, spec.nu), dtype=torch.float32, device=dev))
    if args.fall_code:
        out["fall_code"] = torch.empty((T, N), dtype=torch.uint8, device=dev)

    grf = (torch.empty((T, N, spec.n_grf), dtype=torch.float64, device=dev).normal_(0, 300) if spec.n_grf else None)

    def step(i):
        eng.il_step(qpos, qvel, act, prev[i & 1], prev[(i + 1) & 1], grf_mean=grf, out=out,
                    want_fall_code=args.fall_code)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            if args.backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    timer = HipTimer()
    stream = eng.ctx.stream()
    barrier()
    t0 = time.perf_counter()
    timer.start(stream)
    for i in range(args.steps):
        step(i)
    timer.stop(stream)
    torch.cuda.synchronize(dev)
    barrier()
    wall = time.perf_counter() - t0
    kern_ms = timer.elapsed_ms() / max(args.steps, 1)

    if dist is not None:
        w = torch.tensor([wall], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    rows = T * N
    fallen = float(out["absorbing"].float().mean().item())

    # "also report against a measured device-copy bandwidth on the box": a plain
    # device-to-device copy of a buffer the size of this launch's traffic (read + write counted),
    # outside the timed region, on the same stream with the same timer
    copy_gbps = None
    if rank == 0:
        nbytes = 404 * 1024 * 1024
        src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        ctimer = HipTimer()
        ctimer.start(stream)
        for _ in range(20):
            dst.copy_(src)
        ctimer.stop(stream)
        copy_gbps = 2 * nbytes * 20 / (ctimer.elapsed_ms() * 1e-3) / 1e9
        del src, dst

    if rank == 0:
        bpr = alg_bytes_per_row(spec, args.fall_code)
        achieved = bpr * rows / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_k1.json")
        if os.path.exists(tpath) and (T, N) == (400, 4096) and not args.fall_code and args.robot == "h1":
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "env-steps/sec, 4096 UnitreeH1.walk envs per GPU, obs+reward+done+ctrl HIP kernel",
            "value": world * rows * args.steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / max(args.steps, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": ("UnitreeH1.walk config-2" if args.robot == "h1" else args.robot + ".walk") +
                                   ": fused K1+K5 over one [T,N] block per step",
                       "T": T, "envs_per_gpu": N, "env_steps_per_step": rows * world,
                       "launch_regime": "[T,N] block per launch", "fallen_fraction": fallen,
                       "io": "qpos/qvel f64 + action f32 in; obs/reward/ctrl f32 + absorbing u8 out",
                       "parallelism": f"env-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "measured_copy_GBps": copy_gbps, "frac_of_measured_copy": achieved / copy_gbps,
                         "kernel": "il_tile_kernel<128,%s>" % args.robot if args.robot != "__dyn__" else
                                   "il_dyn_tile_kernel (runtime shape)", "kernel_ms": kern_ms,
                         "alg_bytes_per_env_step": bpr, "env_steps_per_launch": rows},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(spec)
        print(json.dumps(line), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
