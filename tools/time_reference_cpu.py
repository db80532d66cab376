#!/usr/bin/env python3
"""time the REFERENCE's own Python functions on this
container's CPU (the reference never travels to the GPU box).  Runs only where
/root/reference exists.  Per env-step, exactly as the reference calls them after physics:
_create_observation -> is_absorbing/_has_fallen -> reward (TargetVelocityReward) ->
_preprocess_action; and PPOBuffer.finish_path per trajectory of T=400.
Physics (mj_step) and mushroom's _build_obs gather are excluded (packages not installed)."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests", "golden"))


def setup():
    import _ref_stubs as stubs
    ns = stubs.load_reference()
    H1 = ns.h1.UnitreeH1
    env = H1.__new__(H1)
    env._disable_arms, env._disable_back_joint = True, False
    env._algorithm_type = ns.enums.AlgorithmType.IMITATION_LEARNING
    env._use_foot_forces, env._use_absorbing_states = False, True
    jr, mr, _ = env._get_xml_modifications()
    spec = [e for e in H1._get_observation_specification()
            if e[0] not in ["q_" + j for j in jr] + ["dq_" + j for j in jr]]
    env.obs_helper = stubs.FakeObservationHelper(spec)
    env._reward_function = ns.reward.TargetVelocityReward(target_velocity=1.25, x_vel_idx=15)
    env.norm_act_mean, env.norm_act_delta = np.zeros(11), np.full(11, 0.95)
    return env, ns


def h1_loop(args):
    seed, n = args
    env, _ = setup()
    rng = np.random.default_rng(seed)
    rows = rng.normal(0, 0.3, (n, 34))
    acts = rng.uniform(-1, 1, (n, 11))
    prev = env._create_observation(rows[0])
    t0 = time.perf_counter()
    for i in range(n):
        obs = env._create_observation(rows[i])
        ab = env.is_absorbing(obs)
        env.reward(prev, acts[i], obs, ab)
        env._preprocess_action(acts[i])
        prev = obs
    return n / (time.perf_counter() - t0)


def finish_path_rate(ns, T=400, reps=20):
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    for _ in range(reps):
        buf = ns.ppo.PPOBuffer(0.99, 0.95)
        for _ in range(T):
            buf.store(np.zeros((1, 4), np.float32), np.zeros((1, 2), np.float32),
                      np.array([rng.uniform()]), np.array([[0.1]], np.float32))
        buf.finish_path(last_val=np.array([[0.5]], np.float32))
    return reps * T / (time.perf_counter() - t0)


if __name__ == "__main__":
    n = 4096 * 8
    r1 = h1_loop((0, n))
    print(f"H1 per-env path, 1 process: {r1:.3e} env-steps/s ({1e6 / r1:.2f} us/step)")
    cores = mp.cpu_count()
    with mp.Pool(cores) as pool:
        t0 = time.perf_counter()
        pool.map(h1_loop, [(s, n) for s in range(cores)])
        dt = time.perf_counter() - t0
    print(f"H1 per-env path, {cores} processes: {cores * n / dt:.3e} env-steps/s (incl. pool start)")
    _, ns = setup()
    print(f"PPOBuffer store+finish_path, T=400, 1 process: {finish_path_rate(ns):.3e} steps/s")
