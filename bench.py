#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the post-physics hot path for 4096 UnitreeH1.walk
environments per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W [--config 2|5]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

--config 2 (default, the BASELINE metric): one "step" = one launch of the fused K1+K5 kernel (obs
    build + has-fallen + previous-obs reward + action scale/clamp) over one [T=400, N=4096] block of
    synthetic qpos/qvel/action already resident in HBM = 1 638 400 env-steps.  Launch regime: [T,N]
    block per launch.  Environments shard across ranks with no data-path collective (weak scaling);
    the barriers bracketing the timed region are the only communication.  The line also carries a
    `per_step` block: the single-vec-step regime (one launch / one HIP graph per env.step()).
--config 5 (BASELINE.json configs[4]): one "step" = one PPO iteration tail per rank on a
    [T=400, N=4096] shard: K6 return scan (advantage statistics fused into the same pass) -> ONE
    all-gather of (count, sum, sumsq) = 24 B per rank (RCCL over xGMI) -> K7 normalisation that adds
    the rank triples on the device (rl/algos/ppo.py:200-230 merge + :335-336).

The default (config 2) line also carries, measured OUTSIDE the headline's timed region and each wrapped so that
it can never cost the headline:
  world 1:  `configs` = BASELINE configs 3, 4 and 5 on one GPU, each with its own roofline and an oracle
            `cpu_baseline` (config 3: the rollout's kernels K13 / K11 / K10; config 4: the fused VAIL reward K12;
            config 5: the iteration tail at world 1);
  world > 1: `config5_tail` = the iteration tail run in these same rank processes, so that a `--gpus 8` run
            records that the process group saw 8 ranks and what the one collective of the design costs.

With --gpus N > 1 and no WORLD_SIZE in the environment this script starts the N rank processes
itself (the parent never touches a GPU), relays rank 0's JSON line and fails if any rank fails.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "olympics-mujoco_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=[2, 5])
    ap.add_argument("--T", type=int, default=400)
    ap.add_argument("--N", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-per-step", action="store_true", help="skip the single-vec-step block of config 2")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs 3 / 4 / 5 block of the default line")
    ap.add_argument("--fall-code", action="store_true", help="also write the fall-code byte")
    ap.add_argument("--robot", default="h1", choices=["h1", "atlas", "talos", "h1_arms", "h1_ff"],
                    help="h1 is the BASELINE config; the others exercise the same kernel on other tables")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL over xGMI; gloo only to "
                         "rehearse the multi-rank path on a one-GPU box)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------ launcher
def spawn_ranks(args):
    """Parent of a multi-rank run: starts `--gpus` children of this same script, one per GPU, with the
    torch.distributed environment set; never initialises a GPU itself and never exec()s."""
    import socket
    if args.share_device and args.backend != "gloo":
        print("--share-device needs --backend gloo (RCCL refuses two ranks on one GPU)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, errs = [], []
    import tempfile
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        errs.append(tempfile.TemporaryFile(mode="w+"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[-1], text=True))
    # poll every child: a rank that dies before or at the rendezvous would otherwise leave rank 0 blocked in
    # init_process_group / a collective and this launcher in communicate() forever
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        time.sleep(0.05)
    if failed is None:
        failed = next((r for r, p in enumerate(procs) if p.returncode), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    out0 = procs[0].stdout.read() if procs[0].stdout else ""
    rcs = [p.returncode for p in procs]
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    errs[0].seek(0)
    sys.stderr.write(errs[0].read())          # rank 0's warnings, as before
    if failed is not None:
        errs[failed].seek(0)
        print(f"bench.py: rank {failed} failed first; rank exit codes {rcs}; its stderr tail:\n"
              + errs[failed].read()[-2000:], file=sys.stderr)
        return 1
    return 0


class Ranks:
    """torch.distributed plumbing of one rank (world 1: no process group at all)."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = args.backend
        self.dist = None
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        # under torchrun a process group is always formed (a one-rank launch exercises the same RCCL calls)
        if self.world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import datetime
            tmo = datetime.timedelta(seconds=300)          # a missing peer fails the rank instead of hanging it
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev, timeout=tmo)
            else:
                dist.init_process_group("gloo", timeout=tmo)
            self.dist = dist
        if self.world != args.gpus and self.rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={self.world}", file=sys.stderr)

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.dist is not None:
            if self.backend == "nccl":
                self.dist.barrier(device_ids=[self.local_rank])
            else:
                self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        w = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(w, op=self.dist.ReduceOp.MAX)
        return float(w.item())

    def close(self):
        if self.dist is not None:
            self.barrier()
            self.dist.destroy_process_group()


def timed_region(rk, stream, warmup, steps, step):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides.  The clock
    starts after the common barrier and stops when this rank's device has drained; the maximum over ranks
    (taken after the closing barrier) is the job's time, so the closing collective's own latency is not
    billed to the K steps.  Returns (wall seconds, HIP-event ms per step on the kernels' stream)."""
    from olympic_hip._ffi import HipTimer
    torch = rk.torch
    prewarm(rk, step)
    for i in range(warmup):
        step(i)
    timer = HipTimer()
    rk.barrier()
    t0 = time.perf_counter()
    timer.start(stream())
    for i in range(steps):
        step(i)
    timer.stop(stream())
    torch.cuda.synchronize(rk.dev)
    wall = time.perf_counter() - t0          # this rank's K steps are complete; the job's time is the slowest rank's
    rk.barrier()
    return rk.max_over_ranks(wall), timer.elapsed_ms() / max(steps, 1)


PREWARM = {}


def prewarm(rk, step, seconds=0.25):
    """Not a benchmark step: wakes the device.  On this pool an idle MI355X sits at sclk ~600 MHz (`rocm-smi
    --showclocks`) and takes tens of milliseconds of work to reach its running clocks: with the contract's W = 5
    warm-up launches (0.7 ms of work) the K timed launches measured 2.1 ms each instead of 0.13 ms, with W = 50
    0.132 ms (profiles/r03/clock_ramp.json).  So the same launch is repeated for about `seconds` BEFORE the W
    warm-up steps and the K timed steps; nothing here is timed or counted, and the line reports it.
    A step may hold a collective (config 5): every rank must run the SAME number of steps, so the counts are
    derived from per-step times that were MAX-reduced over the ranks, never from a rank's own clock."""
    import torch
    n, elapsed = 0, 0.0
    while True:
        torch.cuda.synchronize(rk.dev)
        t0 = time.perf_counter()
        for _ in range(8):
            step(n)
            n += 1
        torch.cuda.synchronize(rk.dev)
        dt = max(rk.max_over_ranks((time.perf_counter() - t0) / 8), 1e-7)     # identical on every rank
        elapsed += 8 * dt
        if elapsed >= seconds:
            break
        count = int(min((seconds - elapsed) / dt, 4096))
        for _ in range(count):
            step(n)
            n += 1
        elapsed += count * dt
    torch.cuda.synchronize(rk.dev)
    PREWARM.update({"launches": n, "seconds_estimated": elapsed,
                    "why": "device clocks ramp from idle over tens of ms on this pool; untimed, before the W warm-up steps"})


def event_ms(stream, reps, fn, wake_s=0.08):
    """Average HIP-event time of `fn()` over `reps` back-to-back calls on `stream()`, after `wake_s` seconds of the
    same call untimed (the sections of this script alternate with CPU-only baselines during which the device idles
    and drops its clocks, see prewarm)."""
    import torch
    from olympic_hip._ffi import HipTimer
    t0 = time.perf_counter()
    while True:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        if time.perf_counter() - t0 >= wake_s:
            break
    t = HipTimer()
    t.start(stream())
    for _ in range(reps):
        fn()
    t.stop(stream())
    return t.elapsed_ms() / reps


def copy_bandwidth(rk, stream):
    """"also report against a measured device-copy bandwidth on the box": a plain
    device-to-device copy of a buffer the size of the headline launch's traffic (read + write counted)."""
    torch = rk.torch
    nbytes = 404 * 1024 * 1024
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=rk.dev).normal_()
    dst = torch.empty_like(src)
    ms = event_ms(stream, 20, lambda: dst.copy_(src))
    return 2 * nbytes / (ms * 1e-3) / 1e9


# ------------------------------------------------------------------------------------ config 2
def alg_bytes_per_row(spec, fall_code):
    """Algorithmic HBM bytes per (step, env) row of K1+K5, each array counted once:
    qpos + qvel (f64) + action (f32) in; obs (f32) + reward (f32) + absorbing (u8) + ctrl (f32)
    out.  H1: 272 + 44 + 128 + 4 + 1 + 44 = 493 B (+1 with fall codes).  The carried reward
    state (16 B per env per LAUNCH) is not counted."""
    return (8 * (spec.nq + spec.nv + spec.n_grf) + 4 * spec.n_act + 4 * spec.n_obs + 4 + 1 + 4 * spec.nu
            + (1 if fall_code else 0))


def cpu_baseline_config2(spec, seconds=12.0):
    """The CPU oracle (C port of the reference path, parity-pinned to golden vectors) on a
    bounded sample of the same workload, all host cores."""
    import numpy as np
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


def per_step_block(rk, eng, spec, N):
    """The regime the reference API runs: ONE vec step per policy forward (rl/algos/ppo.py:169-196).
    (a) H1: one K1+K5 launch over [1,N] per env.step(), eager and replayed from a HIP graph;
    (b) config 3 (StickFigureA3 PPO sampling): everything between two policy forwards as ONE fused
        launch, the whole vec step (actor + critic forward, Gaussian sample from pre-drawn noise,
        env kernel, buffer stores, device-side resets) as ONE HIP graph replay.
    Wall clock over back-to-back steps, HIP-synchronised at both ends."""
    import torch
    from olympic_hip.synthetic import h1_synthetic_block
    out = {}
    qpos_h, qvel_h, act_h = h1_synthetic_block(spec, 1, N, seed=99)
    qpos, qvel, act = (torch.as_tensor(a).to(rk.dev) for a in (qpos_h, qvel_h, act_h))
    prev = torch.full((N,), 1.25, dtype=torch.float64, device=rk.dev)
    call, _ = eng.il_step_prepare(qpos, qvel, act, prev, want_fall_code=False)
    reps = 2000

    def wall_us(fn):
        for _ in range(50):
            fn()
        torch.cuda.synchronize(rk.dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(rk.dev)
        return 1e6 * (time.perf_counter() - t0) / reps
    us = wall_us(call)
    out["h1_eager_launch"] = {"us_per_vec_step": us, "env_steps_per_s": N / (us * 1e-6), "launches_per_step": 1}
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=rk.dev)
    side.wait_stream(torch.cuda.current_stream(rk.dev))
    with torch.cuda.stream(side):
        call()
    torch.cuda.current_stream(rk.dev).wait_stream(side)
    with torch.cuda.graph(g, stream=side):
        for _ in range(10):
            call()
    us10 = wall_us(g.replay) / 10
    out["h1_graph_of_10_steps"] = {"us_per_vec_step": us10, "env_steps_per_s": N / (us10 * 1e-6),
                                   "launches_per_step": 1, "note": "ten consecutive vec steps per graph replay"}
    try:
        hb = h1_host_batcher(eng, spec, N, qpos_h, qvel_h, act)
        up, down = N * 8 * (spec.nq + spec.nv), N * 8 * spec.nu
        pc = pcie_bandwidth(rk.dev, sizes=((64 << 20, "64MB"), (up, "state_rows"), (down, "controls")))
        # the path's roofline: the larger of the two directions at this box's measured pinned streaming rate (PCIe is full
        # duplex: a perfectly pipelined step would hide the other direction and the physics behind it) + nothing else
        floor_us = 1e6 * max(up / (pc["h2d_64MB"]["GBps"] * 1e9), down / (pc["d2h_64MB"]["GBps"] * 1e9))
        hb["roofline"] = {"bound": "pcie", "achieved": (up + down) / (hb["us_per_vec_step"] * 1e-6) / 1e9,
                          "peak": pc["h2d_64MB"]["GBps"], "unit": "GB/s (pinned H2D, 64 MB blocks, measured in this run)",
                          "frac": floor_us / hb["us_per_vec_step"], "traffic": None, "floor_us_per_vec_step": floor_us,
                          "measured_copies": pc,
                          "note": "frac = (time of the larger direction at the streaming rate) / (measured step): what chunked, "
                                  "double-buffered staging could approach; a copy of the step's own size costs the wall "
                                  "clock under measured_copies (latency-bound)"}
        out["h1_through_the_host_batcher"] = hb
    except Exception as e:                                   # never take the headline line down
        out["h1_through_the_host_batcher"] = {"error": repr(e)[:200]}
    try:
        out["a3_through_the_host_batcher"] = a3_host_batcher(eng, N)
    except Exception as e:
        out["a3_through_the_host_batcher"] = {"error": repr(e)[:200]}
    return out


def pcie_bandwidth(dev, sizes=((64 << 20, "large"),)):
    """Measured pinned-memory copy rates of this box, the roofline of every path that has host memory in the loop:
    H2D and D2H of a 64 MB block (streaming rate) and of the per-step sizes (latency included), wall clock per copy."""
    import torch
    out = {}
    for nbytes, label in sizes:
        h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        for name, src, dst in (("h2d", h, d), ("d2h", d, h)):
            for _ in range(5):
                dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize(dev)
            reps = 10 if nbytes >= (8 << 20) else 100
            dt = float("inf")
            for _ in range(3):                             # best of three batches (the first copies after pinning run slow)
                t0 = time.perf_counter()
                for _ in range(reps):
                    dst.copy_(src, non_blocking=True)
                torch.cuda.synchronize(dev)
                dt = min(dt, (time.perf_counter() - t0) / reps)
            out[f"{name}_{label}"] = {"bytes": nbytes, "us": 1e6 * dt, "GBps": nbytes / dt / 1e9}
    return out


def a3_host_batcher(eng, N, reps=200):
    """The RL robot's batcher (compact staging): PD targets to the host, the physics stand-in on the worker threads, the
    compact readback (base quaternion / angular velocity, actuator and site rows, the USED contact slots) up, K3 (CSR
    form) + K2.  PCIe included; never `value`."""
    import numpy as np
    import torch
    from olympic_hip import specs
    from olympic_hip.batcher import A3HostBatcher
    a3 = specs.A3Spec(mass=41.5)
    eng.a3_configure(a3, np.zeros((4, a3.period)))
    eng.contact_configure(np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32), 0, 7, 10)
    b = A3HostBatcher(eng, N, 16, None, n_threads=min(16, os.cpu_count() or 1)).set_mapped(1).set_compact(1)
    try:
        rng = np.random.default_rng(0)
        nc_all = np.minimum(rng.poisson(4, N), 16)                 # config 3: ncon ~ Poisson(4) clipped to the slots
        for e in range(N):
            sl = b.slots(e)
            sl["root_quat"][:] = [1, 0, 0, 0]
            k_ = int(nc_all[e])
            sl["ncon"][0] = k_
            sl["geom2"][:k_] = 7 + 3 * (np.arange(k_) % 2)
            sl["force6"][:k_] = rng.normal(0, 100, (k_, 6))
        z = lambda dt, *sh: torch.zeros((N,) + sh, dtype=dt, device="cuda")
        st = dict(phase=z(torch.int32), t1=z(torch.int32), t2=z(torch.int32) + 1, reached_frames=z(torch.int32),
                  target_reached=z(torch.uint8), mode=z(torch.int32) + 2, seq_len=z(torch.int32) + 20,
                  sequence=z(torch.float64, 20, 4), goal=z(torch.float64, 8))
        a = torch.zeros((N, 12), device="cuda")
        for _ in range(20):
            b.step(a, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            b.step(a, st)
        torch.cuda.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / reps
    finally:
        b.close()
    up = int(N * (8 * (7 + 12 * 2 + 3 * 6 + 4 + 3) + 4 + 8) + int(nc_all.sum()) * 64)     # fixed compact row + used 64-B records
    return {"us_per_vec_step": us, "env_steps_per_s": N / (us * 1e-6), "pcie_bytes_per_step": {"down": N * 8 * 12, "up": up},
            "note": "compact staging (oly_a3_batcher_set_compact), mapped PD targets; contact density Poisson(4) of 16 slots; "
                    "physics is a kinematic stand-in"}


def h1_host_batcher(eng, spec, N, qpos_h, qvel_h, act, reps=300):
    """PCIe INCLUDED (never `value`): the C++ host batcher's vec step - controls to the host, the physics
    stand-in (qpos += dt * qvel) on the worker threads, state rows to the device, K1 - with mapped staging."""
    import torch
    from olympic_hip.batcher import HostBatcher
    b = HostBatcher(eng, N, n_threads=min(16, os.cpu_count() or 1), dt=0.01).set_mapped(True)
    try:
        b.qpos[:], b.qvel[:] = qpos_h[0], qvel_h[0]
        a = act[0].contiguous()
        for _ in range(20):
            b.step(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            b.step(a)
        torch.cuda.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / reps
    finally:
        b.close()
    return {"us_per_vec_step": us, "env_steps_per_s": N / (us * 1e-6), "pcie_bytes_per_step": N * 8 * (spec.nq + spec.nv + spec.nu),
            "note": "host memory in the loop: PCIe- and sync-bound by construction; physics is a kinematic stand-in"}


F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 matrix peak (v_mfma_f32_32x32x2_f32)


def _a3_rollout_env(N, dev):
    import numpy as np
    import torch
    from olympic_hip import specs
    from olympic_hip.a3 import ReplayA3Physics, VecA3Env
    from olympic_hip.engine import Engine
    from olympic_hip.ppo import MLPCritic, MLPGaussianActor
    from olympic_hip.synthetic import A3_FLOOR_BODY, A3_GEOM_BODYID, A3_LFOOT_BODY, A3_RFOOT_BODY, a3_synthetic_blocks
    blocks = {k: torch.as_tensor(v).to(dev) for k, v in a3_synthetic_blocks(N, 32, seed=1).items()}
    env = VecA3Env(specs.A3Spec(mass=41.5), N, Engine(dev.index or 0), ReplayA3Physics(blocks), A3_GEOM_BODYID,
                   A3_FLOOR_BODY, A3_RFOOT_BODY, A3_LFOOT_BODY, rs=np.random.RandomState(0))
    torch.manual_seed(0)
    return env, MLPGaussianActor(41, 12).to(dev), MLPCritic(41).to(dev)


def config3_sampling(N, dev, T=400, reps=3):
    """BASELINE.json configs[2], the sampling half: N StickFigureA3 environments x T steps with the
    synthetic physics readback resident on the device, policy 41->256->256->12 + critic -> 1 (random
    normc-like init), Gaussian sampling, episode cuts and device-side resets, three ways:
      persistent      ONE launch per rollout (K13: a workgroup owns 32 environments through all T steps)
      graph_replay    per vec step one K11 launch (both MLPs) + one K10 launch, replayed from HIP graphs of 8 steps
      eager_launches  the same two launches per step, issued op by op
    Wall clock of whole rollouts (reset launch, bootstrap pass, reset-pool refill included) / T.  Also returns the
    environment and networks for the kernel-level timings."""
    import torch
    env, pi, vf = _a3_rollout_env(N, dev)
    res = {}
    for label, graph, persistent in (("eager_launches", False, False), ("graph_replay", True, False),
                                     ("persistent", False, True)):
        for _ in range(2):
            env.device_rollout(pi, vf, T, T, graph=graph, persistent=persistent)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            env.device_rollout(pi, vf, T, T, graph=graph, persistent=persistent)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps
        res[label] = {"us_per_vec_step": 1e6 * dt / T, "env_steps_per_s": N * T / dt}
    res["eager_launches"]["launches_per_step"] = res["graph_replay"]["launches_per_step"] = 2
    res["persistent"]["launches_per_rollout"] = 1
    info = env._dev_rollout.last_info
    res.update({"N": N, "T": T, "kernels": "a3_rollout_kernel (K13) | mlp_forward16_kernel (K11, 16-row tiles at this batch) + a3_vec_kernel (K10)",
                "timing": "wall clock of whole rollouts (reset launch, bootstrap pass, reset-pool refill included) / T",
                "resets_per_rollout": info.get("resets"), "bootstrap_rows": info.get("side_rows"),
                "round1_host_loop_us_per_vec_step": 360.0, "round2_graph_replay_us_per_vec_step": 33.9})
    return res, (env, pi, vf)


def _mlp_flop_per_row(n_in, n_act):
    """Algorithmic multiply-adds x 2 of the actor (n_in -> 256 -> 256 -> n_act) + critic (-> 1) forward."""
    return 2 * ((n_in * 256 + 256 * 256 + 256 * n_act) + (n_in * 256 + 256 * 256 + 256 * 1))


def _k10_bytes_per_env_step(spec, C):
    """Algorithmic HBM bytes of one K10 vec step per environment, each array once: the readback row (root / head /
    foot poses and velocities, base quaternion and angular velocity, actuator rows, ncon + C contact slots), mu /
    noise / the stored observation and value in; observation (buffer + next), action, PD target, float64 reward,
    value, flag, task state (ints + goal steps) out.  The 640-B step sequence is read every step (counted) and
    rewritten only on resets (not counted)."""
    nu, nobs = spec.nu, spec.n_obs
    rd = 8 * (4 + 3 * 6 + 4 + 3 + 2 * nu) + 4 + C * (4 + 4 + 48 + 8) + 4 * (2 * nu + nobs + 1) + 8 * 80 + 4 * 7
    wr = 4 * (2 * nobs + nu + 1) + 8 * nu + 8 + 1 + 4 * 5 + 1 + 8 * 8
    return rd + wr


def _cpu_rate_all_cores(make_job, seconds, cores=None):
    """Units per second of `cores` host threads, each looping its OWN job (a closure over its own slice of the workload
    that returns the units of one pass) until the deadline.  The oracle's C functions are single-threaded; ctypes
    releases the GIL around them, so Python threads do run them side by side.  -> (rate, cores)."""
    import threading
    from oracle import oracle as orc
    # the GPU box hands one GPU's job a CPU share of 16 cores (more threads only time-slice them; 128 OpenMP threads
    # run config 2's port ~10x faster than one): 32 threads cover the share with room for stalls
    cores = int(cores or min(orc.max_threads(), 32))
    jobs = [make_job(i, cores) for i in range(cores)]
    for j in jobs[:1]:
        j()                                     # warm (page in the oracle, first-touch the arrays)
    done = [0] * cores
    t0 = time.perf_counter()
    deadline = t0 + seconds

    def run(i):
        while time.perf_counter() < deadline:
            done[i] += jobs[i]()
    ths = [threading.Thread(target=run, args=(i,)) for i in range(cores)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return sum(done) / (time.perf_counter() - t0), cores


def _config3_cpu_job(N, seed=0):
    """One sampling step of N environments through the oracle: oly_mlp_forward_cpu (actor, critic) + oly_a3_vec_step_cpu."""
    import numpy as np
    from oracle import oracle as orc
    from olympic_hip import _abi, specs
    from olympic_hip.a3 import clock_lut
    from olympic_hip.synthetic import A3_FLOOR_BODY, A3_GEOM_BODYID, A3_LFOOT_BODY, A3_RFOOT_BODY, a3_synthetic_blocks
    from olympic_hip.vecstep import draw_reset_records
    spec = specs.A3Spec(mass=41.5)
    lut = clock_lut(spec.swing_duration, spec.stance_duration, 0.1, "grounded", 1 / spec.control_dt, spec.period)
    contact = (A3_GEOM_BODYID, A3_FLOOR_BODY, A3_RFOOT_BODY, A3_LFOOT_BODY)
    K, T, nobs, nu, depth = 8, 1 << 20, spec.n_obs, spec.nu, 4
    rng = np.random.default_rng(seed)
    blocks = a3_synthetic_blocks(N, K, seed=1 + seed)
    z = lambda dt, *sh: np.zeros((N,) + sh, dt)
    state = dict(phase=z(np.int32), t1=z(np.int32), t2=z(np.int32), reached_frames=z(np.int32), target_reached=z(np.uint8),
                 mode=np.full(N, _abi.MODE_STANDING, np.int32), seq_len=np.ones(N, np.int32),
                 sequence=z(np.float64, _abi.OLY_MAX_SEQ, 4), goal=z(np.float64, 8))
    Tb = 64                     # buffer rows; the step counter is rewound before it reaches them
    ro = dict(T=T, max_traj_len=400, deterministic=False, side_slots=4, pool_depth=depth, mu=z(np.float32, nu),
              value=z(np.float32), scale=np.full(nu, 0.2, np.float32), eps=rng.normal(0, 1, (Tb, N, nu)).astype(np.float32),
              state=z(np.float32, nobs), pd_target=z(np.float64, nu), buf_states=np.zeros((Tb, N, nobs), np.float32),
              buf_actions=np.zeros((Tb, N, nu), np.float32), buf_rewards=np.zeros((Tb, N)),
              buf_values=np.zeros((Tb, N), np.float32), buf_flags=np.zeros((Tb, N), np.uint8), buf_rew6=None,
              traj_len=z(np.int32), side_obs=np.zeros((N * 4, nobs), np.float32), side_t=np.full(N * 4, -1, np.int32),
              side_count=z(np.int32), pool=draw_reset_records(np.random.RandomState(seed), N * depth, spec, 5000)
              .view(np.uint8).reshape(-1).copy(), pool_count=z(np.int32), ctr=np.zeros(2, np.int32))
    wa = [rng.normal(0, 0.1, sh).astype(np.float32) for sh in ((256, nobs), (256,), (256, 256), (256,), (nu, 256), (nu,))]
    wc = [rng.normal(0, 0.1, sh).astype(np.float32) for sh in ((256, nobs), (256,), (256, 256), (256,), (1, 256), (1,))]
    orc.a3_vec_step(spec, lut, contact, blocks, state, ro, _abi.VSTEP_RESET_ALL)

    def job():
        ro["mu"][:] = orc.mlp_forward(ro["state"], *wa)
        ro["value"][:] = orc.mlp_forward(ro["state"], *wc)[:, 0]
        orc.a3_vec_step(spec, lut, contact, blocks, state, ro, 0)
        if ro["ctr"][0] >= Tb:
            ro["ctr"][0] = 0
            ro["side_count"][:] = 0
        return N
    return job


PORT_NOTE = ("the port evaluates every layer as scalar k-ordered fma chains (the matrix cores' exact order, kept for "
             "bit-exact parity); a BLAS forward, which is what the reference's torch CPU path runs, would be several "
             "times faster per core")


def cpu_baseline_config3(seconds=9.0, N=4096):
    """The oracle's restatement of one sampling step (oly_mlp_forward_cpu for actor and critic, then
    oly_a3_vec_step_cpu) on N environments split over all host threads, and on one thread."""
    rate, cores = _cpu_rate_all_cores(lambda i, c: _config3_cpu_job(max(16, N // c), seed=i), seconds * 0.65)
    one, _ = _cpu_rate_all_cores(lambda i, c: _config3_cpu_job(256, seed=0), seconds * 0.35, cores=1)
    return {"value": rate, "unit": "env-steps/s", "cores": cores, "kind": "port", "value_1thread": one,
            "sample": f"oracle actor + critic forward (oly_mlp_forward_cpu) + oly_a3_vec_step_cpu on {max(16, N // cores) * cores} "
                      f"environments split over {cores} threads for ~{seconds * 0.65:.0f} s (1 thread: 256 environments); " + PORT_NOTE}


def _update_flop_per_row(n_in, n_act, mirror):
    """Algorithmic FLOP of one row of a PPO minibatch update (2 x multiply-adds): forward + backward (data gradients of the
    two hidden layers, weight gradients of all three) of the actor and the critic; with the mirror loss the actor runs
    forward + backward on the mirrored row as well.  (The kernel's recomputed third forward is not credited.)"""
    fb = lambda o: 2 * (n_in * 256 + 65536 + 256 * o) + 2 * (2 * 256 * o + 2 * 65536 + 256 * n_in)
    return fb(n_act) * (2 if mirror else 1) + fb(1)


def config3_iteration(env, pi, vf, args, minibatch=65536, epochs=3, n_itr=5):
    """A WHOLE config-3 iteration as PPO.train runs it (rl/algos/ppo.py:305-420): rollout (K13) -> return scan +
    statistics + normalisation (K6 / K7) -> `epochs` x (n / minibatch) minibatch updates, each = K14 (forward, losses
    incl. the mirror-symmetry loss the reference's train_a3_walk.py switches on by default, backward) + clip + Adam +
    re-pack (oly_ppo_adam_step).  Wall clock from PPO.train's own instrumentation, steady iterations only."""
    import tempfile

    import numpy as np
    import torch
    from olympic_hip import specs
    from olympic_hip.ppo import PPO, KernelUpdate
    from olympic_hip.wrappers import SymmetricEnv
    N, T = env.num_envs, 400
    spec = specs.A3Spec(mass=41.5)
    out = {}
    for label, mirror in (("with_mirror_loss", True), ("without_mirror_loss", False)):
        hp = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=minibatch,
                  epochs=epochs, max_traj_len=T, use_gae=False, num_procs=N, max_grad_norm=0.05,
                  mirror_coeff=0.4 if mirror else 0.0, eval_freq=10 ** 9)
        ppo = PPO(hp, tempfile.mkdtemp(prefix="oly_bench_ppo_"))
        torch.manual_seed(0)
        pi_, vf_ = type(pi)(41, 12).to(env.eng.device), type(vf)(41).to(env.eng.device)
        sym = SymmetricEnv(lambda: env, mirrored_obs=list(spec.mirrored_obs), mirrored_act=list(spec.mirrored_acts),
                           clock_inds=list(spec.clock_inds))
        hist = ppo.train((lambda: sym) if mirror else (lambda: env), pi_, vf_, n_itr=n_itr, verbose=False)
        steady = hist[2:] or hist[1:] or hist
        sample_s = float(np.mean([h["sample_s"] for h in steady]))
        optim_s = float(np.mean([h["optim_s"] for h in steady]))
        n_upd = epochs * ((N * T) // minibatch)
        out[label] = {"sample_s": sample_s, "update_s": optim_s, "updates": n_upd, "update_ms_per_minibatch": 1e3 * optim_s / n_upd,
                      "env_steps_per_s": N * T / (sample_s + optim_s), "losses_finite": bool(np.isfinite(hist[-1]["losses"]).all())}
    # the update's kernels alone, HIP events on the launch stream: gradients (K14 main + finishing launch) and the
    # optimiser launch (clip + Adam, the stepped weights written into the packed streams; the norm's block partials come
    # from the finishing launch), on minibatches gathered from a full-size buffer
    n = N * T
    dev = env.eng.device
    obs, act = torch.randn(n, 41, device=dev), 0.3 * torch.randn(n, 12, device=dev)
    ret, adv = torch.randn(n, device=dev), torch.randn(n, device=dev)
    import copy
    stream = env.eng.ctx.stream
    kern = {}
    for label, mirror in (("with_mirror_loss", True), ("without_mirror_loss", False)):
        pi_, vf_ = type(pi)(41, 12).to(dev), type(vf)(41).to(dev)
        sym = SymmetricEnv(lambda: env, mirrored_obs=list(spec.mirrored_obs), mirrored_act=list(spec.mirrored_acts),
                           clock_inds=list(spec.clock_inds))
        ku = KernelUpdate(env.eng, pi_, vf_, copy.deepcopy(pi_), 0.2, 0.5, 0.4 if mirror else 0.0,
                          sym.mirror_clock_observation if mirror else None,
                          sym._act_src if mirror else None, sym._act_sgn if mirror else None)
        ku.begin(obs)
        row = {}
        for B in (minibatch, 64):
            idx = torch.randperm(n, device=dev)[:B].to(torch.int32)
            g_ms = event_ms(stream, 30 if B > 4096 else 200, lambda: ku.grads(obs, act, ret, adv, idx))
            a_ms = event_ms(stream, 200, lambda: ku.apply(norm_ready=True), wake_s=0.0)
            flop = _update_flop_per_row(41, 12, mirror) * B
            row[f"minibatch_{B}"] = {
                "gradients_ms": g_ms, "clip_adam_ms": a_ms, "update_ms": g_ms + a_ms, "launches": 3,
                "roofline": {"bound": "mfma", "achieved": flop / (g_ms * 1e-3) / 1e12, "peak": F32_MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": flop / (g_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                             "alg_flop_per_row": flop // B,
                             "note": "ppo_update_kernel + ppo_update_finish_kernel (K14), both launches in the time"}}
        kern[label] = row
    out["kernels"] = kern
    out["workload"] = (f"{N} envs x {T} steps per iteration, {epochs} epochs x {(N * T) // minibatch} minibatches of {minibatch} rows, "
                       "policy 41-256-256-12 + critic 41-256-256-1, Adam + grad-norm clip; rounds 1-3 ran this phase as "
                       "PyTorch GEMMs (0.24-0.30 s per iteration)")
    out["value"] = out["with_mirror_loss"]["env_steps_per_s"]
    out["unit"] = "env-steps/s over a whole iteration (sampling + update)"
    return out


def config3_block(rk, args):
    """BASELINE config 3 on one GPU: the sampling loop's wall clock per vec step, and its kernels against their
    rooflines, HIP-event timed on the launch stream: K13 (the whole rollout in one launch; f32 MFMA), K11 (actor +
    critic forward; f32 MFMA) and K10 (the vec step; HBM bytes), the last two as the two-launch path runs them."""
    import torch
    from olympic_hip import specs
    N, dev = args.N, rk.dev
    res, (env, pi, vf) = config3_sampling(N, dev)
    r = env._dev_rollout
    fw, T, spec = r._fw, r.buf.T, specs.A3Spec(mass=41.5)
    stream = env.eng.ctx.stream
    mu, v = fw.outputs(N)
    flop = _mlp_flop_per_row(spec.n_obs, spec.nu) * N

    def rewind():
        r.traj_len.zero_()
        r.side_count.zero_()
        r.side_t.fill_(-1)
        r.pool_count.zero_()
        r.ctr[0::2] = 0
    k13 = []
    for _ in range(4):
        rewind()
        k13.append(_one_event_ms(stream, lambda: r.launch.persistent(fw.packed_a, fw.norm_a, fw.packed_c, fw.norm_c, mu, v)))
    k13_us = 1e3 * min(k13) / T
    rewind()
    k11_us = 1e3 * event_ms(stream, 100, lambda: fw(r.state_obs))
    rewind()
    k10_us = 1e3 * event_ms(stream, 100, lambda: r.launch(0, mu, v), wake_s=0.0)   # 104 steps < T rows (K11 just woke the device)
    rewind()
    b10 = _k10_bytes_per_env_step(spec, int(r.blocks["geom1"].shape[2])) * N
    out = {"workload": f"A3 walk PPO sampling, {N} envs x {T} steps, synthetic readback resident on the device "
                       "(BASELINE configs[2]); update phase = PyTorch-ROCm, not timed here",
           "sampling": res, "value": res["persistent"]["env_steps_per_s"], "unit": "env-steps/s",
           "kernels": {
               "a3_rollout_kernel(K13)": {
                   "us_per_vec_step": k13_us, "launches_per_rollout": 1,
                   "roofline": {"bound": "mfma", "achieved": flop / (k13_us * 1e-6) / 1e12, "peak": F32_MFMA_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": flop / (k13_us * 1e-6) / 1e12 / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                                "alg_flop_per_env_step": flop // N,
                                "note": "a step = the forward (matrix cores, waves 0-3) beside the latency-bound environment "
                                        "step (fp64, waves 4-7) of the same 16 environments; f32-input MFMAs hold the "
                                        "vector ALU, so the two nearly add up: interval times in profiles/r03"}},
               "mlp_forward16_kernel(K11)": {
                   "us_per_launch": k11_us,
                   "roofline": {"bound": "mfma", "achieved": flop / (k11_us * 1e-6) / 1e12, "peak": F32_MFMA_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": flop / (k11_us * 1e-6) / 1e12 / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                                "alg_flop_per_env_step": flop // N}},
               "a3_vec_kernel(K10)": {
                   "us_per_launch": k10_us,
                   "roofline": {"bound": "hbm", "achieved": b10 / (k10_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": b10 / (k10_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                "alg_bytes_per_env_step": b10 // N,
                                "note": "10 MB per launch = 1.2 us of HBM time: launch- and dependent-latency-bound"}}},
           "timing": "HIP events on the launch stream, back-to-back launches (K11 / K10: 100 each; K13: best of 4 whole "
                     "rollouts / T)"}
    env._dev_rollout.close()
    try:
        out["iteration"] = config3_iteration(env, pi, vf, args)
    except Exception as e:                                   # never lose the sampling numbers to the second half
        out["iteration"] = {"error": f"{type(e).__name__}: {e}"}
    out["workload"] = out["workload"].replace("update phase = PyTorch-ROCm, not timed here", "whole iterations under `iteration`")
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_config3()
    return out


def _one_event_ms(stream, fn):
    from olympic_hip._ffi import HipTimer
    t = HipTimer()
    t.start(stream())
    fn()
    t.stop(stream())
    return t.elapsed_ms()


# ------------------------------------------------------------------------------------ config 4
def disc_flop_per_sample(D=32):
    """Algorithmic multiply-adds x 2 of the variational discriminator D -> 256 -> 128 -> (128, 128) -> 1."""
    return 2 * (D * 256 + 256 * 128 + 2 * 128 * 128 + 128)


def _config4_cpu_job(weights, B, seed, pipeline=True):
    """make_discrim_reward -> compute_gae(0.99, 0.97) -> biased-std normalisation on B samples (one [1, B] block)."""
    import numpy as np
    from oracle import oracle as orc
    from olympic_hip import _abi
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 1, (B, 32)).astype(np.float32)
    eps = rng.normal(0, 1, (B, 128)).astype(np.float32)
    v, nv = (rng.normal(0, 1, (1, B)).astype(np.float32) for _ in range(2))
    fl = ((rng.uniform(size=(1, B)) < 0.01) * (_abi.FLAG_LAST | _abi.FLAG_ABSORBING)).astype(np.uint8)

    def job():
        cs = orc.col_stats(x)
        r = orc.disc_forward(x, weights, colstats=cs, eps=eps)["reward"]
        if pipeline:
            _, adv = orc.return_scan(_abi.SCAN_GAE, 0.99, 0.97, r.reshape(1, B), v, nv, fl)
            orc.adv_normalize(adv, orc.adv_stats(adv), 0, 1e-8)
        return B
    return job


def cpu_baseline_config4(weights, seconds=9.0, B=4096):
    rate, cores = _cpu_rate_all_cores(lambda i, c: _config4_cpu_job(weights, max(32, B // c), i), seconds * 0.65)
    one, _ = _cpu_rate_all_cores(lambda i, c: _config4_cpu_job(weights, 1024, 0), seconds * 0.35, cores=1)
    return {"value": rate, "unit": "samples/s", "cores": cores, "kind": "port", "value_1thread": one,
            "sample": f"oracle column statistics + oly_disc_forward_cpu + GAE(0.97) scan + biased-std normalisation on "
                      f"{max(32, B // cores) * cores} samples split over {cores} threads for ~{seconds * 0.65:.0f} s (1 thread: 1024 "
                      "samples); " + PORT_NOTE}


def config4_block(rk, args):
    """BASELINE config 4: the VAIL discriminator reward (imitation_lib/imitation/gail_TRPO.py:320-327 ->
    VariationalNet.forward, utils/networks.py:258-284) for B = 4096 samples and for a [400,4096] block:
    oly_col_stats (the Standardizer's running update) + oly_disc_forward (K12, one launch on the f32 matrix cores),
    and the whole reward -> GAE(0.97) -> normalisation pipeline of GAIL.fit (gail_TRPO.py:116-129)."""
    import numpy as np
    import torch
    from olympic_hip.engine import Engine
    from olympic_hip.gail import DiscriminatorReward, VariationalDiscriminator
    dev = rk.dev
    eng = Engine(dev.index or 0)
    torch.manual_seed(0)
    net = VariationalDiscriminator().to(dev)
    stream = eng.ctx.stream
    out = {"workload": "UnitreeH1 VAIL reward, variational discriminator 32 -> 256 -> 128 -> (128, 128) -> 1, "
                       "eps supplied (BASELINE configs[3])", "unit": "samples/s", "shapes": {}}
    for label, B, reps in (("B4096", args.N, 200), ("T400xN4096", 400 * args.N, 10)):
        g = torch.Generator(device=dev).manual_seed(7)
        x = torch.randn((B, 32), device=dev, generator=g)
        eps = torch.randn((B, 128), device=dev, generator=g)
        dr = DiscriminatorReward(eng, net, state_mask=np.arange(32))
        packed = dr.packed()
        cs = eng.col_stats(x)
        bufs = dict(reward=torch.empty(B, device=dev))
        ms_k12 = event_ms(stream, reps, lambda: eng.disc_forward(x, packed, colstats=cs, eps=eps, out=bufs))
        ms_all = event_ms(stream, reps, lambda: dr.forward(x, eps, out=bufs))
        fl = disc_flop_per_sample() * B
        traffic = src = None
        tpath = os.path.join(ROOT, "profiles", "r03", "traffic_k12.json")
        if label == "T400xN4096" and B == 1638400 and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                src = "profiles/r03/traffic_k12.json (rocprofv3 PMC passes of round 3; read from the file, not measured in this run)"
            except Exception:
                traffic = None
        out["shapes"][label] = {
            "samples": B, "disc_forward_us": 1e3 * ms_k12,
            # DiscriminatorReward.forward as a user calls it from Python: per-call argument checks included (host-bound at
            # B = 4096; the same launches issued through the prepared call: pipeline.stages_us.reward_step below)
            "python_forward_call_us": 1e3 * ms_all,
            "samples_per_s": B / (ms_all * 1e-3),
            "roofline": {"bound": "mfma", "achieved": fl / (ms_k12 * 1e-3) / 1e12, "peak": F32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": fl / (ms_k12 * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_source": src, "kernel": "disc_forward_kernel<4>", "alg_flop_per_sample": disc_flop_per_sample(),
                         "hbm_bytes_per_sample": 4 * (32 + 128 + 1)}}
        # the pipeline SURVEY 8(d) names for this config: reward -> compute_gae(gamma .99, lambda .97) -> advantage
        # normalisation (ddof 0, 1e-8)  (gail_TRPO.py:116-129): one [T, N] block, critic values given
        from olympic_hip import _abi
        from olympic_hip.rollout import GAERollout, RolloutBuffer
        T = B // args.N
        buf = RolloutBuffer(T, args.N, 32, 1, dev)
        buf.values.normal_(0, 1, generator=g)
        buf.next_values.normal_(0, 1, generator=g)
        buf.flags.copy_(((torch.rand((T, args.N), device=dev, generator=g) < 0.01) * (_abi.FLAG_LAST | _abi.FLAG_ABSORBING)).to(torch.uint8))
        buf.ptr = T
        post = GAERollout(eng, gamma=0.99, lam=0.97)
        rew_out = dict(reward=buf.rewards.view(-1))

        rstep = dr.prepared(x, eps, out=rew_out)          # arguments validated once (the per-call checks cost ~20 us of host time)

        def pipeline():
            rstep()                                        # re-pack + Standardizer update + K12, reward straight into the block
            post.finish(buf, normalize=True)               # GAE scan with fused statistics, normalise (ddof 0)
        ms_pipe = event_ms(stream, reps, pipeline)
        ms_rew = event_ms(stream, reps, rstep, wake_s=0.0)
        dr.cache_packed = True                             # a frozen network: no re-pack launch in front of the statistics
        rstep_frozen = dr.prepared(x, eps, out=rew_out)
        ms_rew_frozen = event_ms(stream, reps, rstep_frozen, wake_s=0.0)
        dr.cache_packed = False
        ms_gae = event_ms(stream, reps, lambda: post.finish(buf, normalize=False), wake_s=0.0)
        ms_full = event_ms(stream, reps, lambda: post.finish(buf, normalize=True), wake_s=0.0)
        parts3 = post._stats.reshape(1, 3)
        ms_norm = event_ms(stream, reps, lambda: eng.adv_normalize(buf.advantages, parts3, 0, 1e-8), wake_s=0.0)
        scan_bytes = (21 + 8) * B                          # GAE scan 21 B / element, normalise 8 B / element
        # the config's own step (Standardizer update + reward) as one prepared C call: round 3's 31.8 us at B = 4096
        out["shapes"][label]["reward_step_us"] = 1e3 * ms_rew
        out["shapes"][label]["reward_step_frozen_weights_us"] = 1e3 * ms_rew_frozen
        out["shapes"][label]["pipeline"] = {
            "stages": "Standardizer update + K12 reward | K6 GAE(0.99, 0.97) + statistics | K7 normalise (ddof 0, 1e-8)",
            "us": 1e3 * ms_pipe, "samples_per_s": B / (ms_pipe * 1e-3),
            "stages_us": {"reward_step": 1e3 * ms_rew, "reward_step_frozen_weights": 1e3 * ms_rew_frozen,
                          "gae_scan_with_statistics": 1e3 * ms_gae,
                          "normalise": 1e3 * ms_norm},
            "launches": 4 + 2 + 1,
            "note": "reward_step = ONE C call: re-pack of the live weights (so a reward never uses weights an optimiser step or "
                    "a broadcast has replaced), two statistics launches, K12; *_frozen_weights skips the re-pack",
            "roofline_scan_and_normalise": {"bound": "hbm", "achieved": scan_bytes / (ms_full * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                            "unit": "GB/s", "frac": scan_bytes / (ms_full * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "traffic": None, "alg_bytes_per_sample": 29}}
        del x, eps, bufs, buf
    out["value"] = out["shapes"]["T400xN4096"]["samples_per_s"]
    out["round1_layer_by_layer_us_at_B4096"] = 249.0
    if not args.no_cpu_baseline:
        n = net
        w = {k: t.detach().cpu().numpy() for k, t in dict(
            enc_w0=n.encoder[0].weight, enc_b0=n.encoder[0].bias, enc_w1=n.encoder[1].weight, enc_b1=n.encoder[1].bias,
            mu_w=n.mu_out.weight, mu_b=n.mu_out.bias, lv_w=n.logvar_out.weight, lv_b=n.logvar_out.bias,
            dec_w=n.decoder.weight, dec_b=n.decoder.bias).items()}
        out["cpu_baseline"] = cpu_baseline_config4(w)
    return out


def config5_world1_block(rk, args):
    """BASELINE config 5's iteration tail on ONE GPU (its 8-GPU form is `bench.py --config 5 --gpus 8`; under a
    multi-rank default run the same tail is measured as `config5_tail`)."""
    from olympic_hip.engine import Engine
    m = measure_config5_tail(rk, Engine(rk.local_rank), 400, args.N, 100, 10)
    out = {"workload": "PPO iteration tail on one [400,4096] shard: K6 scan (statistics fused, f64 rewards) -> "
                       "(world 1: the all-gather is a view) -> K7 normalise", "value": m["value"], "unit": m["unit"],
           "ms_per_step": m["ms_per_step"], "stages_ms": m["stages_ms"], "roofline": m["roofline"]}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_config5(400, args.N, seconds=6.0)
    return out


def configs_block(rk, args):
    out = {}
    for name, fn in (("config3_a3_ppo_sampling", config3_block), ("config4_vail_reward", config4_block),
                     ("config5_tail_world1", config5_world1_block)):
        try:                       # a secondary block must never cost the headline line
            out[name] = fn(rk, args)
        except Exception as e:     # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def bench_config2(args, rk):
    import torch
    from olympic_hip import specs
    from olympic_hip.engine import Engine
    from olympic_hip.synthetic import h1_synthetic_block
    dev, rank, world = rk.dev, rk.rank, rk.world
    spec = {"h1": lambda: specs.unitree_h1("walk"), "atlas": lambda: specs.atlas("walk"),
            "talos": lambda: specs.talos("walk"),
            "h1_arms": lambda: specs.unitree_h1("walk", disable_arms=False),
            "h1_ff": lambda: specs.unitree_h1("walk").with_foot_forces("UnitreeH1")}[args.robot]()
    eng = Engine(rk.local_rank).il_configure(spec)
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

    wall, kern_ms = timed_region(rk, eng.ctx.stream, args.warmup, args.steps, step)
    rows = T * N
    fallen = float(out["absorbing"].float().mean().item())
    # N > 1: the design's ONE exchange step (rl/algos/ppo.py:200-230,335-336) measured in these same rank processes,
    # so that a driver --gpus N run of the default command records that the process group saw N ranks and what the
    # all-gather costs (config 2 itself has no data-path collective).  Every rank takes part.
    tail = None
    if world > 1 and not args.no_configs:
        try:
            tail = measure_config5_tail(rk, eng, 400, N, 50, 5)
        except Exception as e:     # noqa: BLE001  (never costs the headline line)
            tail = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank != 0:
        return None
    copy_gbps = copy_bandwidth(rk, eng.ctx.stream)
    bpr = alg_bytes_per_row(spec, args.fall_code)
    achieved = bpr * rows / (kern_ms * 1e-3) / 1e9
    traffic, traffic_round = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_k1.json")
    if os.path.exists(tpath) and (T, N) == (400, 4096) and not args.fall_code and args.robot == "h1":
        try:
            tj = json.load(open(tpath))
            traffic, traffic_round = tj.get("hbm_bytes_per_launch"), tj.get("round")
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
        "prewarm": dict(PREWARM),
        "config": {"workload": ("UnitreeH1.walk config-2" if args.robot == "h1" else args.robot + ".walk") +
                               ": fused K1+K5 over one [T,N] block per step",
                   "T": T, "envs_per_gpu": N, "env_steps_per_step": rows * world,
                   "launch_regime": "[T,N] block per launch", "fallen_fraction": fallen,
                   "io": "qpos/qvel f64 + action f32 in; obs/reward/ctrl f32 + absorbing u8 out",
                   "parallelism": f"env-sharded x{world}, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": (f"profiles/traffic_k1.json (rocprofv3 PMC passes of round {traffic_round}, FETCH_SIZE x 2 "
                                        "+ WRITE_SIZE; read from the file, not measured in this run)" if traffic else None),
                     "measured_copy_GBps": copy_gbps, "frac_of_measured_copy": achieved / copy_gbps,
                     "measured_copy_note": "torch's device-to-device copy_ of 404 MB timed the same way; it is slower than "
                                           "the guide's float4 copy (6.29 TB/s), so a ratio above 1 only says that",
                     "kernel": "il_tile_kernel<128,%s>" % args.robot, "kernel_ms": kern_ms,
                     "alg_bytes_per_env_step": bpr, "env_steps_per_launch": rows},
    }
    if tail is not None:
        tail.pop("wall_s", None)
        tail["workload"] = ("config-5 iteration tail per rank on its own [400,N] shard: K6 scan (statistics fused) -> ONE "
                            "all-gather of 24 B per rank -> K7 normalise; measured after the headline's timed region")
        line["config5_tail"] = tail
    if world == 1 and not args.no_per_step and args.robot == "h1":
        try:                       # a secondary block must never cost the headline line
            line["per_step"] = per_step_block(rk, eng, spec, N)
        except Exception as e:     # noqa: BLE001
            line["per_step"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_configs and args.robot == "h1":
        line["configs"] = configs_block(rk, args)
        c3 = line["configs"].get("config3_a3_ppo_sampling", {})
        if "per_step" in line and "sampling" in c3:
            line["per_step"]["config3_a3_ppo_sampling"] = c3["sampling"]
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_config2(spec)
    return line


# ------------------------------------------------------------------------------------ config 5
def _config5_cpu_job(T, N, seed):
    import numpy as np
    from oracle import oracle as orc
    rng = np.random.default_rng(5 + seed)
    r = rng.uniform(-0.3, 1.0, (T, N))
    v, nv = (rng.normal(0, 1, (T, N)).astype(np.float32) for _ in range(2))
    fl = ((rng.uniform(size=(T, N)) < 1 / 300) * 2).astype(np.uint8)

    def job():
        _, adv = orc.return_scan_r64(0.99, r, v, nv, fl)
        orc.adv_normalize(adv, orc.adv_stats(adv), 1, 1e-5)
        return T * N
    return job


def cpu_baseline_config5(T, N, seconds=9.0):
    rate, cores = _cpu_rate_all_cores(lambda i, c: _config5_cpu_job(T, max(8, N // c), i), seconds * 0.65)
    one, _ = _cpu_rate_all_cores(lambda i, c: _config5_cpu_job(T, N, 0), seconds * 0.35, cores=1)
    return {"value": rate, "unit": "env-steps/s", "cores": cores, "kind": "port", "value_1thread": one,
            "sample": f"oracle return scan (f64 rewards) + statistics + normalisation on one [T={T},N={N}] shard, its "
                      f"environments split over {cores} threads (each normalises its own slice: an upper bound of a "
                      f"shared-statistics run), ~{seconds * 0.65:.0f} s; 1 thread: the whole shard"}


def measure_config5_tail(rk, eng, T, N, steps, warmup):
    """K6 (+ fused statistics) -> all-gather of 3 doubles per rank -> K7 normalise, per rank on its own
    [T,N] shard of the N*world environments (rl/algos/ppo.py:200-230,335-336).  Returns a dict with the timed
    region's wall time, the per-stage times and the roofline of scan + normalise; every rank must call it."""
    import torch
    from olympic_hip import _abi, dist as odist
    dev, rank, world = rk.dev, rk.rank, rk.world
    g = torch.Generator(device=dev).manual_seed(500 + rank)
    rew = torch.empty((T, N), dtype=torch.float64, device=dev).uniform_(-0.3, 1.0, generator=g)
    val = torch.empty((T, N), dtype=torch.float32, device=dev).normal_(0, 1, generator=g)
    nval = torch.empty((T, N), dtype=torch.float32, device=dev).normal_(0, 1, generator=g)
    last = torch.rand((T, N), device=dev, generator=g) < 1 / 300
    absorb = last & (torch.rand((T, N), device=dev, generator=g) < 0.5)
    flags = (last.to(torch.uint8) * _abi.FLAG_LAST) | (absorb.to(torch.uint8) * _abi.FLAG_ABSORBING)
    ret = torch.empty((T, N), dtype=torch.float32, device=dev)
    adv = torch.empty((T, N), dtype=torch.float32, device=dev)
    st = torch.zeros(3, dtype=torch.float64, device=dev)

    def scan():
        eng.return_scan(_abi.SCAN_RETURN, 0.99, 0.95, rew, val, nval, flags, ret, adv, stats3=st)

    def gather():
        return odist.gather_stats(st)

    def step(i):
        scan()
        eng.adv_normalize(adv, gather(), 1, 1e-5)

    wall, ms_step = timed_region(rk, eng.ctx.stream, warmup, steps, step)
    # stage breakdown, outside the timed region: each stage alone, back to back, HIP events
    reps = max(20, min(steps, 200))
    ms_scan = event_ms(eng.ctx.stream, reps, scan)
    parts = gather()
    ms_norm = event_ms(eng.ctx.stream, reps, lambda: eng.adv_normalize(adv, parts, 1, 1e-5))
    rk.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        gather()
    torch.cuda.synchronize(dev)
    ms_gather = 1e3 * (time.perf_counter() - t0) / reps
    ms_gather = rk.max_over_ranks(ms_gather)
    # the all-gathered triples are identical on every rank, their tree sum is the global statistic
    tot = odist.tree_sum(parts)
    assert float(tot[0]) == float(T * N * world), (float(tot[0]), T * N * world)
    elems = T * N
    b_scan, b_norm = 21, 8         # f64 reward 8 + value 4 + next value 4 + flag 1 in, ret 4 + adv 4 out | adv in + out
    ranks_seen = rk.dist.get_world_size() if rk.dist is not None else 1
    return {
        "wall_s": wall, "steps": steps, "value": world * elems * steps / wall, "unit": "env-steps/s",
        "ms_per_step": 1e3 * wall / max(steps, 1), "ranks_seen": ranks_seen, "gathered_rows": int(parts.shape[0]),
        "backend": rk.backend if rk.dist is not None else None, "T": T, "envs_per_gpu": N,
        "stages_ms": {"scan_with_fused_stats": ms_scan, "all_gather_host_wall": ms_gather, "normalise": ms_norm,
                      "whole_step_hip_events": ms_step},
        "roofline": {"bound": "hbm", "achieved": (b_scan + b_norm) * elems / ((ms_scan + ms_norm) * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (b_scan + b_norm) * elems / ((ms_scan + ms_norm) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "scan_pipe_kernel + normalize_kernel",
                     "alg_bytes_per_element": {"scan": b_scan, "normalise": b_norm,
                                               "note": "SURVEY 8(d) prices K6 at 17 B (f32 rewards) and K7 at 12 B (a "
                                                       "separate statistics pass, 4 B, which the fused scan no longer makes)"},
                     "scan_GBps": b_scan * elems / (ms_scan * 1e-3) / 1e9,
                     "normalise_GBps": b_norm * elems / (ms_norm * 1e-3) / 1e9,
                     "elements_per_launch": elems,
                     "note": "the [400,4096] tail is bound by the 400-step dependent fp64 chain and launch latency, "
                             "not by its 48 MB (DESIGN 4, K6)"},
    }


def bench_config5(args, rk):
    from olympic_hip.engine import Engine
    world = rk.world
    eng = Engine(rk.local_rank)
    T, N = args.T, args.N
    m = measure_config5_tail(rk, eng, T, N, args.steps, args.warmup)
    if rk.rank != 0:
        return None
    line = {
        "metric": "env-steps/sec through the PPO iteration tail (return scan + advantage statistics + "
                  "all-gather + normalisation), 4096 envs x 400 steps per GPU",
        "value": m["value"],
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": m["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "prewarm": dict(PREWARM),
        "config": {"workload": "config-5: K6 scan (statistics fused) -> all-gather 24 B/rank -> K7 normalise, "
                               "one [T,N] shard per rank",
                   "T": T, "envs_per_gpu": N, "envs_total": N * world, "backend": m["backend"],
                   "ranks_seen": m["ranks_seen"],
                   "rewards": "f64 (un-narrowed, as env.step returns them)",
                   "parallelism": f"env-sharded x{world}, one all-gather of 3 doubles per rank per step"},
        "stages_ms": m["stages_ms"],
        "roofline": m["roofline"],
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_config5(T, N)
    return line


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if os.environ.get("OLY_BENCH_TEST_DIE_RANK") == os.environ.get("RANK", "0"):
        sys.exit(3)            # test hook (tests/test_gpu_multirank.py): a rank that dies before the rendezvous
    rk = Ranks(args)
    line = (bench_config5 if args.config == 5 else bench_config2)(args, rk)
    if rk.rank == 0 and line is not None:
        print(json.dumps(line), flush=True)
    rk.close()


if __name__ == "__main__":
    main()
