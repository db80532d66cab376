"""GPU tests of the host facade: the reference's env API (make/reset/step/
play_trajectory_from_velocity/create_dataset), the vectorised PPO rollout post-processing and
the VAIL discriminator reward, each against the oracle / golden vectors."""
import numpy as np
import pytest
import torch

from olympic_hip import _abi, specs
from helpers import h1_rows_from_full, h1_synthetic_block, ulp_diff

pytestmark = pytest.mark.gpu


def host(t):
    return t.cpu().numpy()


# ----------------------------------------------------------------- config 1: plumbing, N = 1
def test_make_reset_step_single_env(oracle):
    from olympic_hip.envs import LocoEnvBase
    env = LocoEnvBase.make("UnitreeH1.walk.real", seed=3)
    assert env.info.observation_space.shape == (32,) and env.info.action_space.shape == (11,)
    assert env.info.gamma == 0.99 and env.info.horizon == 1000 and env.dt == 0.01
    obs = env.reset()
    assert obs.shape == (32,) and obs.dtype == np.float64
    # the reset observation is the trajectory sample minus x,y (loco_env_base.py:603, 737)
    smp = host(env.vec._sample)[0]
    assert np.array_equal(obs, smp[2:34])
    assert smp[0] == 0.0 and smp[1] == 0.0                       # x,y re-zeroed at the reset step
    prev_x = obs[15]
    o2, r, absorbing, info = env.step(np.zeros(11))
    assert isinstance(r, float) and isinstance(absorbing, bool) and info == {}
    # kinematic stand-in: state unchanged, reward reads the PREVIOUS obs (utils/reward.py:73)
    assert np.array_equal(o2, obs)
    assert abs(r - np.exp(-np.square(prev_x - 1.25))) <= 1.2e-7
    assert env.get_obs_idx("dq_pelvis_tx") == [15] and len(env.get_kinematic_obs_mask()) == 32
    ds = env.create_dataset()
    n = env.vec.trajectories.trajectory_length * env.vec.trajectories.number_of_trajectories
    assert ds["states"].shape == (n - 1, 32) and ds["last"].sum() == 2


def test_play_trajectory_from_velocity_matches_host_replay(oracle):
    """500 replay steps: device cursor + Euler integration + obs/has-fallen equal the
    reference algorithm (loco_env_base.py:505-542) evaluated on the host with the oracle."""
    from olympic_hip.envs import LocoEnvBase
    env = LocoEnvBase.make("UnitreeH1.walk.real", seed=11)
    sp = env.spec
    obs_rec, fallen = env.play_trajectory_from_velocity(n_episodes=1, n_steps_per_episode=500)
    assert obs_rec.shape == (500, 1, 32)
    tr = env.vec.trajectories
    table = tr.table
    # host replay with the same reset indices (seed 11 -> first draw)
    rng = np.random.default_rng(11)
    L, J = tr.trajectory_length, tr.number_of_trajectories
    exp = []
    tn, st = rng.integers(0, J, 1), rng.integers(0, L, 1)
    ct, cs, org, smp = oracle.traj_reset(table, tn, st)
    cur = smp[:, :17].copy()
    for _ in range(500):
        smp = oracle.traj_euler(17, 0.01, cur, smp)
        cur = smp[:, :17].copy()
        cs, smp, at_end = oracle.traj_next(table, ct, cs, org, smp)
        if at_end[0]:
            tn, st = rng.integers(0, J, 1), rng.integers(0, L, 1)
            ct, cs, org, smp = oracle.traj_reset(table, tn, st)
            cur = smp[:, :17].copy()
        exp.append(smp[0, 2:34].copy())
    assert np.array_equal(host(obs_rec)[:, 0], np.array(exp))
    assert not host(fallen).any()                                  # the synthetic gait never falls


def test_vec_env_matches_oracle_over_an_episode(oracle):
    """N = 4096 replayed synthetic physics, 20 steps: obs/reward/absorbing per step equal the
    oracle's [T,N] block evaluation (config 2 per-step regime)."""
    from olympic_hip.envs import ReplayPhysics, VecLocoEnv
    sp = specs.unitree_h1("walk")
    T, N = 20, 4096
    qpos, qvel, act = h1_synthetic_block(sp, T, N, seed=77, fall_frac="wide")
    phys = ReplayPhysics(sp, torch.as_tensor(qpos).cuda(), torch.as_tensor(qvel).cuda())
    env = VecLocoEnv(sp, N, device=0, physics=phys, random_start=False)
    prev0 = np.zeros(N)
    ref = oracle.il_step(sp, qpos, qvel, act, prev0)
    env._prev.zero_()
    for t in range(T):
        o, r, a, info = env.step(torch.as_tensor(act[t]).cuda())
        assert np.array_equal(host(o), ref["obs"][t])
        assert np.array_equal(host(a), ref["absorbing"][t].astype(bool))
        assert ulp_diff(host(r), ref["reward"][t]).max() <= 1
        assert np.array_equal(host(info["ctrl"]), ref["ctrl"][t])
    assert int(env.episode_steps[0]) == T


# ----------------------------------------------------------------- config 3: PPO post-processing
def test_ppo_rollout_block(oracle):
    from olympic_hip.engine import Engine
    from olympic_hip.rollout import PPORollout, RolloutBuffer
    eng = Engine(0)
    T, N = 400, 4096
    g = torch.Generator(device="cuda").manual_seed(5)
    buf = RolloutBuffer(T, N, 41, 12, eng.device)
    buf.rewards.uniform_(-0.3, 1.0, generator=g)
    buf.values.normal_(0, 1, generator=g)
    buf.next_values.normal_(0, 1, generator=g)
    ends = torch.rand((T, N), device="cuda", generator=g) < 1 / 300
    dones = ends & (torch.rand((T, N), device="cuda", generator=g) < 0.7)
    buf.flags.copy_((ends.to(torch.uint8) * _abi.FLAG_LAST) | (dones.to(torch.uint8) * _abi.FLAG_ABSORBING))
    buf.ptr = T
    pr = PPORollout(eng, gamma=0.99, lam=0.95, eps=1e-5)
    ret, adv = pr.finish(buf, normalize=False)
    e_ret, e_adv = oracle.return_scan(_abi.SCAN_RETURN, 0.99, 0.95, host(buf.rewards), host(buf.values),
                                      host(buf.next_values), host(buf.flags))
    assert np.array_equal(host(ret), e_ret) and np.array_equal(host(adv), e_adv)     # bit-exact
    raw = adv.clone()
    pr.normalize(adv, ddof=1, eps=1e-5)
    ref = (raw - raw.mean()) / (raw.std() + 1e-5)            # the reference's torch expression
    assert (adv - ref).abs().max().item() < 5e-5
    assert abs(adv.mean().item()) < 1e-5 and abs(adv.std().item() - 1) < 1e-4
    er, el = buf.episode_stats()
    assert len(er) == len(el) == int((ends | (torch.arange(T, device="cuda") == T - 1).unsqueeze(1)).sum())
    assert sum(el) == T * N
    assert abs(sum(er) - float(buf.rewards.double().sum())) < 1e-9 * T * N


def test_collect_with_torch_policy(oracle):
    """PPO.sample in lock step: flags/next_values bookkeeping around resets and time limits."""
    from olympic_hip.envs import VecLocoEnv
    from olympic_hip.rollout import PPORollout, RolloutBuffer, collect
    from olympic_hip.trajectory import Trajectory, synthetic_h1_trajectory_files
    sp = specs.unitree_h1("walk")
    files = synthetic_h1_trajectory_files(sp, n_traj=2, length=500)
    low = np.concatenate([sp.joint_lo, -np.inf * np.ones(17)])
    high = np.concatenate([sp.joint_hi, np.inf * np.ones(17)])
    low[:6], high[:6] = -np.inf, np.inf
    tr = Trajectory(keys=list(sp.obs_keys), low=low[2:], high=high[2:], joint_pos_idx=np.arange(17),
                    traj_files=files, clip_trajectory_to_joint_ranges=True, warn=False)
    N, T = 256, 24
    env = VecLocoEnv(sp, N, device=0, trajectory=tr, seed=2)
    torch.manual_seed(0)
    policy = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.Tanh(), torch.nn.Linear(64, 11)).cuda()
    critic = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.Tanh(), torch.nn.Linear(64, 1)).cuda()
    buf = RolloutBuffer(T, N, 32, 11, env.device)
    collect(env, policy, critic, buf, max_traj_len=10)
    fl = host(buf.flags)
    assert (fl[9] & _abi.FLAG_LAST).all() and (fl[19] & _abi.FLAG_LAST).all() and (fl[T - 1] & _abi.FLAG_LAST).all()
    assert not (fl & _abi.FLAG_ABSORBING).any()                     # kinematic stand-in never falls
    assert not (fl[[0, 5, 12]] & _abi.FLAG_LAST).any()
    ret, adv = PPORollout(env.eng, 0.99, 0.95).finish(buf, normalize=False)
    e_ret, _ = oracle.return_scan(_abi.SCAN_RETURN, 0.99, 0.95, host(buf.rewards), host(buf.values),
                                  host(buf.next_values), fl)
    assert np.array_equal(host(ret), e_ret)
    # reward of the step right after a reset reads the RESET observation (self._obs)
    x_reset = host(buf.states)[10, :, 15]
    np.testing.assert_allclose(host(buf.rewards)[10], np.exp(-(x_reset.astype(np.float64) - 1.25) ** 2), rtol=3e-6)


# ----------------------------------------------------------------- config 4: VAIL reward
def test_vail_discriminator_reward_matches_reference(golden):
    from olympic_hip.engine import Engine
    from olympic_hip.gail import DiscriminatorReward, VariationalDiscriminator
    g = golden("vail_disc.npz")
    eng = Engine(0)
    net = VariationalDiscriminator().load_reference_arrays(g).cuda()
    dr = DiscriminatorReward(eng, net, state_mask=np.arange(32))
    x, eps = torch.as_tensor(g["x"]).cuda(), torch.as_tensor(g["eps"]).cuda()
    d, mu, logvar = dr.logits(x, eps)
    np.testing.assert_allclose(host(dr.stand.mean), g["st_mean"], rtol=2e-5, atol=2e-6)   # ref sums in f32
    np.testing.assert_allclose(host(dr.stand.std), g["st_std"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(host(mu), g["mu"], rtol=2e-4, atol=2e-5)                 # GEMM order
    np.testing.assert_allclose(host(d), g["d"].reshape(-1), rtol=2e-3, atol=2e-3)
    dr2 = DiscriminatorReward(eng, net, state_mask=np.arange(32))
    r = host(dr2(x, eps))
    ok = np.abs(g["d"].reshape(-1)) < 8
    np.testing.assert_allclose(r[ok], g["reward"][ok], rtol=5e-3, atol=5e-3)
    # the standardiser updates on EVERY forward, reward evaluation included (networks.py:68-70)
    assert float(dr2.stand.colstats[0, 0]) == len(x)
    dr2(x, eps)
    assert float(dr2.stand.colstats[0, 0]) == 2 * len(x)


def test_gail_gae_pipeline(oracle):
    from olympic_hip.engine import Engine
    from olympic_hip.rollout import GAERollout, RolloutBuffer
    eng = Engine(0)
    T, N = 250, 4096
    g = torch.Generator(device="cuda").manual_seed(9)
    buf = RolloutBuffer(T, N, 32, 11, eng.device)
    buf.rewards.uniform_(0, 3, generator=g)
    buf.values.normal_(0, 1, generator=g)
    buf.next_values.normal_(0, 1, generator=g)
    last = torch.rand((T, N), device="cuda", generator=g) < 1 / 100
    ab = last & (torch.rand((T, N), device="cuda", generator=g) < 0.5)
    buf.flags.copy_((last.to(torch.uint8) * _abi.FLAG_LAST) | (ab.to(torch.uint8) * _abi.FLAG_ABSORBING))
    gr = GAERollout(eng, gamma=0.99, lam=0.97)
    ret, adv = gr.finish(buf, normalize=False)
    e_ret, e_adv = oracle.return_scan(_abi.SCAN_GAE, 0.99, 0.97, host(buf.rewards), host(buf.values),
                                      host(buf.next_values), host(buf.flags))
    assert np.array_equal(host(adv), e_adv) and np.array_equal(host(ret), e_ret)
    raw = host(adv).astype(np.float64)
    gr.normalize(adv, ddof=0, eps=1e-8)
    ref = (raw - raw.mean()) / (raw.std() + 1e-8)                  # gail_TRPO.py:128 (numpy, biased)
    np.testing.assert_allclose(host(adv), ref, rtol=2e-5, atol=2e-6)


# ----------------------------------------------------------------- config 3: A3 vec env
def test_vec_a3_env_replays_golden_sequence(golden):
    """Host reset (reference RNG draws) + pd_target -> contact_reduce -> a3_step per step, fed
    with the recorded physics readback, reproduces the reference's WalkingTask/get_obs."""
    from olympic_hip.a3 import ReplayA3Physics, VecA3Env
    from olympic_hip.engine import Engine
    g = golden("a3_task.npz")
    E, K = g["phase"].shape
    sp = specs.A3Spec(mass=float(g["mass"]))
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    blocks = {n: dev(np.swapaxes(g[n], 0, 1)) for n in
              ("qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel", "rf_vel", "root_pos",
               "root_quat", "head_pos", "ncon", "geom1", "geom2", "force6", "cpos_z")}
    env = VecA3Env(sp, E, Engine(0), ReplayA3Physics(blocks), g["geom_bodyid"], int(g["floor_body"]),
                   int(g["rfoot_body"]), int(g["lfoot_body"]), obs_f64=True)
    assert np.array_equal(env.lut, g["clock_lut"])
    for e in range(E):                                   # per-env seeds, as the fixture was generated
        np.random.seed(1000 + e)
        env.iteration_count = int(g["iter_count"][e])
        env.reset_task([e], g["reset_lfoot"][e:e + 1], g["reset_rfoot"][e:e + 1], g["reset_root_quat"][e:e + 1])
    assert np.array_equal(host(env.state["mode"]), g["mode"]) and np.array_equal(host(env.state["phase"]), g["phase0"])
    act = torch.zeros((E, 12), device="cuda")
    for k in range(K):
        obs, rew, done, rew6 = env.step(act)
        assert np.array_equal(host(env.state["phase"]), g["phase"][:, k])
        assert np.array_equal(host(env.state["t1"]), g["t1"][:, k]) and np.array_equal(host(env.state["t2"]), g["t2"][:, k])
        assert np.array_equal(host(done), g["done"][:, k])
        np.testing.assert_allclose(host(obs), g["obs"][:, k], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(host(rew), g["reward"][:, k], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(host(rew6), g["rew6"][:, k], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("minibatch,n_itr", [(2048, 2), (256, 2)])
@pytest.mark.parametrize("mode", ["torch_losses", "fused", "fused_graph", "fused_graph_fresh", "kernel", "fused_target_kl",
                                  "kernel_target_kl"])
def test_ppo_train_iterations_on_vec_a3(golden, tmp_path, mode, minibatch, n_itr):
    """Config 3 end to end: VecA3Env (synthetic physics readback) -> PPO.train: rollout,
    return scan + adv-norm on the device, clipped-surrogate updates in PyTorch."""
    from olympic_hip.a3 import ReplayA3Physics, VecA3Env
    from olympic_hip.engine import Engine
    from olympic_hip.ppo import PPO, MLPCritic, MLPGaussianActor
    g = golden("a3_task.npz")
    N, K = 512, 40
    sp = specs.A3Spec(mass=41.5)
    gen = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s: torch.empty(s, dtype=torch.float64, device="cuda").normal_(0, 1, generator=gen)
    E = g["phase"].shape[0]
    seq = torch.as_tensor(g["sequence"]).cuda()[torch.randint(0, E, (N,), device="cuda", generator=gen)]
    base = seq[:, 1, :3]
    blocks = dict(qpos=rnd(K, N, 25), qvel=rnd(K, N, 24), act_len=rnd(K, N, 12), act_vel=rnd(K, N, 12),
                  lf_pos=base + 0.1 * rnd(K, N, 3), rf_pos=base + 0.3 * rnd(K, N, 3), lf_vel=0.2 * rnd(K, N, 3),
                  rf_vel=0.2 * rnd(K, N, 3), root_pos=base + torch.tensor([0, 0, 0.8], device="cuda") + 0.05 * rnd(K, N, 3),
                  root_quat=rnd(K, N, 4), head_pos=base + torch.tensor([0, 0, 1.2], device="cuda") + 0.05 * rnd(K, N, 3),
                  ncon=torch.randint(0, 5, (K, N), device="cuda", generator=gen, dtype=torch.int32),
                  geom1=torch.zeros((K, N, 16), dtype=torch.int32, device="cuda"),
                  geom2=torch.randint(8, 13, (K, N, 16), device="cuda", generator=gen, dtype=torch.int32),
                  force6=100 * rnd(K, N, 16, 6), cpos_z=0.01 * rnd(K, N, 16))
    blocks["root_quat"] = blocks["root_quat"] / blocks["root_quat"].norm(dim=-1, keepdim=True)   # xquat is unit
    blocks = {k: v.contiguous() for k, v in blocks.items()}

    class Env(VecA3Env):
        def __init__(self):
            super().__init__(sp, N, Engine(0), ReplayA3Physics(blocks), g["geom_bodyid"], 0, 7, 10)
            self.device = self.eng.device
            self.state["sequence"].copy_(seq)
            self.state["seq_len"].fill_(20)
            self.state["mode"].fill_(_abi.MODE_FORWARD)
            self.state["t2"].fill_(1)

        def reset(self, env_mask=None):
            if env_mask is None:
                self.state["phase"].zero_()
            else:
                self.state["phase"][env_mask] = 0
            return torch.zeros((N, 41), device="cuda")
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=minibatch, epochs=2,
                max_traj_len=16, use_gae=False, num_procs=N, max_grad_norm=0.05, mirror_coeff=0.0,
                eval_freq=2 if mode == "fused" else 100)
    ppo = PPO(args, str(tmp_path))
    ppo.use_device_rollout = False               # this test drives the per-step host loop (custom reset above)
    ppo.fused_loss, ppo.use_graph = mode != "torch_losses", mode.startswith("fused_graph")
    ppo.update_kernel = mode.startswith("kernel")   # K14: forward + losses + backward of a minibatch in one launch
    if mode.endswith("target_kl"):
        ppo.target_kl = 1e-8
    ppo.device_permutation = False               # every mode cuts the same host-drawn permutation: same seed, same batches
    if mode == "fused_graph_fresh":
        ppo.graph_recapture_every = 1            # every update runs as the FIRST replay of a new capture
    torch.manual_seed(0)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    w0 = pi.means.weight.detach().clone()
    hist = ppo.train(Env, pi, vf, n_itr=n_itr, verbose=False)
    assert len(hist) == n_itr and all(np.isfinite(h["losses"]).all() for h in hist)
    assert not torch.equal(w0, pi.means.weight)                      # the optimiser stepped
    if mode.endswith("target_kl"):
        # ppo.py:391-393, 412-414: the first minibatch of an iteration sees ratio one (KL exactly 0) and steps; the second
        # sees the stepped policy, breaches 1.5 target_kl and ends the whole update phase BEFORE its own optimiser step
        steps = ppo.kupd.steps if mode.startswith("kernel") else int(ppo.actor_optimizer.state[pi.means.weight]["step"])
        assert steps == n_itr, steps
        assert all(h["losses"][3] > 1.5e-8 / 2 for h in hist)            # mean KL of the two minibatches it evaluated
        return
    assert ppo.total_steps == n_itr * 16 * N
    lines = open(ppo.train_fn).read().strip().splitlines()
    assert lines[0] == "ep_returns,ep_lens" and len(lines) == n_itr + 1
    if mode == "fused":                      # eval_freq = 2: one deterministic evaluation + checkpoints
        ev = open(ppo.eval_fn).read().strip().splitlines()
        assert ev[0] == "test_ep_returns,test_ep_lens" and len(ev) == 2 and "eval_return" in hist[1]
        import os
        assert os.path.exists(os.path.join(str(tmp_path), "actor_1.pt")) and os.path.exists(os.path.join(str(tmp_path), "critic.pt"))
        assert ppo.highest_reward == hist[1]["eval_return"]
    # the update paths are the same algorithm: same seed -> same trained weights.  Across
    # implementations that holds to fp32 rounding for a few updates (16 at minibatch 2048; Adam
    # amplifies rounding noise over longer runs, so no bound is asserted at 128 updates).  A graph
    # replayed many times against one re-captured before every update runs the SAME kernels, so
    # those two must agree bit for bit: that is the check that replays do not go stale.
    _TRAINED[mode, minibatch] = torch.cat([p.detach().reshape(-1) for p in list(pi.parameters()) + list(vf.parameters())]).cpu()
    if ("torch_losses", minibatch) in _TRAINED and mode != "torch_losses" and minibatch == 2048:
        a, b = _TRAINED["torch_losses", minibatch], _TRAINED[mode, minibatch]
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max())
    if mode == "fused_graph_fresh":
        assert torch.equal(_TRAINED["fused_graph", minibatch], _TRAINED["fused_graph_fresh", minibatch])
    if mode == "fused_graph":
        # the eager fused update and its graph replay are the same kernels with the same optimiser
        # arithmetic (Adam capturable on both sides): bit-identical weights after all updates
        assert torch.equal(_TRAINED["fused", minibatch], _TRAINED["fused_graph", minibatch])


_TRAINED = {}


def test_discriminator_reward_follows_writes_through_dot_data(golden):
    """parameters written through `.data` (what dist.broadcast_parameters and load paths do) do not bump
    `_version`; the fused reward must still use the new weights (the stream is re-packed per call)."""
    from olympic_hip.engine import Engine
    from olympic_hip.gail import DiscriminatorReward, VariationalDiscriminator
    g = golden("vail_disc.npz")
    eng = Engine(0)
    net = VariationalDiscriminator().load_reference_arrays(g).cuda()
    dr = DiscriminatorReward(eng, net, state_mask=np.arange(32))
    gen = torch.Generator(device="cuda").manual_seed(0)
    x = torch.empty((300, 32), device="cuda").normal_(0, 1, generator=gen)
    eps = torch.empty((300, 128), device="cuda").normal_(0, 1, generator=gen)
    before = dr.forward(x, eps, want=("logits",))["logits"].clone()
    v0 = net.decoder.weight._version
    net.decoder.weight.data.mul_(2.0)
    net.encoder[0].bias.data.add_(0.25)
    assert net.decoder.weight._version == v0                     # the write the version counter does not see
    after = dr.forward(x, eps, want=("logits",))["logits"]
    assert not torch.equal(before, after)
    ref, _, _ = dr.logits_unfused(x, eps)                        # torch GEMMs on the CURRENT parameters
    np.testing.assert_allclose(host(after), host(ref), rtol=2e-4, atol=2e-4)
    dr.cache_packed = True                                       # the opt-in cache: explicit invalidation
    p1 = dr.packed().clone()
    net.decoder.weight.data.mul_(0.5)
    assert torch.equal(dr.packed(), p1)                          # the documented hazard of the cache
    dr.invalidate()
    assert not torch.equal(dr.packed(), p1)
    dr.cache_packed = False
    net.decoder.weight.data.mul_(3.0)
    assert not torch.equal(dr.packed().clone(), p1)


@pytest.mark.parametrize("T,N", [(1, 4096), (5, 4096)])
def test_config4_three_stage_pipeline_against_the_oracle(golden, oracle, T, N):
    """BASELINE config 4 as SURVEY 8(d) states it, B = N = 4096: Standardizer update + VAIL reward (one C call:
    oly_disc_reward_step) -> compute_gae(0.99, 0.97) -> (adv - mean) / (biased std + 1e-8)  (gail_TRPO.py:116-129),
    every stage against the oracle: reward / logits bit-exact, GAE targets and raw advantages bit-exact, statistics 1e-12,
    normalised advantages one float32 ulp.  Two batches in a row: the second accumulates onto the running statistics."""
    from olympic_hip.engine import Engine
    from olympic_hip.gail import DiscriminatorReward, VariationalDiscriminator
    from olympic_hip.rollout import GAERollout, RolloutBuffer
    g = golden("vail_disc.npz")
    eng = Engine(0)
    net = VariationalDiscriminator().load_reference_arrays(g).cuda()
    w = {k: np.asarray(g[k]) for k in ("enc_w0", "enc_b0", "enc_w1", "enc_b1", "mu_w", "mu_b", "lv_w", "lv_b", "dec_w", "dec_b")}
    dr = DiscriminatorReward(eng, net, state_mask=np.arange(32))
    post = GAERollout(eng, gamma=0.99, lam=0.97)
    gen = torch.Generator(device="cuda").manual_seed(T)
    B = T * N
    cs_ref = None
    for batch in range(2):
        x = torch.empty((B, 32), device="cuda").normal_(0.3 * batch, 1 + batch, generator=gen)
        eps = torch.empty((B, 128), device="cuda").normal_(0, 1, generator=gen)
        buf = RolloutBuffer(T, N, 32, 1, "cuda")
        buf.values.normal_(0, 1, generator=gen)
        buf.next_values.normal_(0, 1, generator=gen)
        last = torch.rand((T, N), device="cuda", generator=gen) < 0.1
        absorbing = last & (torch.rand((T, N), device="cuda", generator=gen) < 0.5)
        buf.flags.copy_((last.to(torch.uint8) * _abi.FLAG_LAST) | (absorbing.to(torch.uint8) * _abi.FLAG_ABSORBING))
        buf.ptr = T
        o = dr.forward(x, eps, want=("reward", "logits"), out=dict(reward=buf.rewards.view(-1)))
        # stage 1 against the oracle: running column statistics, then the fused forward on them
        cs = oracle.col_stats(host(x))
        cs_ref = cs if cs_ref is None else cs_ref + cs
        np.testing.assert_allclose(host(dr.stand.colstats), cs_ref, rtol=1e-12)
        e = oracle.disc_forward(host(x), w, colstats=host(dr.stand.colstats), eps=host(eps))
        assert np.array_equal(host(o["logits"]), e["logits"]) and np.array_equal(host(buf.rewards).reshape(-1), e["reward"])
        # stages 2 + 3
        v_target, adv = post.finish(buf, normalize=True)
        e_ret, e_adv = oracle.return_scan(_abi.SCAN_GAE, 0.99, 0.97, host(buf.rewards), host(buf.values), host(buf.next_values),
                                          host(buf.flags))
        assert np.array_equal(host(v_target), e_ret)
        st = oracle.adv_stats(e_adv)
        np.testing.assert_allclose(host(post._stats), st, rtol=1e-12)
        e_norm = oracle.adv_normalize(e_adv, st, 0, 1e-8)
        assert np.abs(host(adv) - e_norm).max() <= np.spacing(np.float32(1.0)) * max(1.0, np.abs(e_norm).max())


def test_gail_fit_reward_and_advantage_pipeline(golden, oracle):
    """Config 4 at N = 4096: discriminator reward -> GAE(0.97) -> biased-std normalisation."""
    from olympic_hip.engine import Engine
    from olympic_hip.gail import DiscriminatorReward, GAILAdvantage, VariationalDiscriminator
    g = golden("vail_disc.npz")
    eng = Engine(0)
    net = VariationalDiscriminator().load_reference_arrays(g).cuda()
    torch.manual_seed(0)
    critic = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 1)).cuda()
    T, N = 8, 4096
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.empty((T, N, 32), device="cuda").normal_(0, 1, generator=gen)
    xn = torch.empty((T, N, 32), device="cuda").normal_(0, 1, generator=gen)
    eps = torch.empty((T * N, 128), device="cuda").normal_(0, 1, generator=gen)
    last = torch.rand((T, N), device="cuda", generator=gen) < 0.1
    absorbing = last & (torch.rand((T, N), device="cuda", generator=gen) < 0.5)
    r_env = torch.zeros((T, N), device="cuda")
    pipe = GAILAdvantage(eng, DiscriminatorReward(eng, net, state_mask=np.arange(32)), critic, 0.99, 0.97)
    r, v_target, adv = pipe(x, xn, r_env, absorbing, last, eps)
    assert torch.isfinite(r).all() and (r >= 0).all()              # -log(1 - p + 1e-8) >= -log(1+1e-8)
    v = critic(x.reshape(-1, 32)).reshape(T, N)
    vn = critic(xn.reshape(-1, 32)).reshape(T, N)
    flags = (last.to(torch.uint8) * _abi.FLAG_LAST) | (absorbing.to(torch.uint8) * _abi.FLAG_ABSORBING)
    e_ret, e_adv = oracle.return_scan(_abi.SCAN_GAE, 0.99, 0.97, host(r), host(v.detach()), host(vn.detach()), host(flags))
    assert np.array_equal(host(v_target), e_ret)                   # v_target = adv + v, pre-normalisation
    ref = (e_adv.astype(np.float64) - e_adv.mean(dtype=np.float64)) / (e_adv.astype(np.float64).std() + 1e-8)
    np.testing.assert_allclose(host(adv), ref, rtol=2e-5, atol=2e-6)


# ----------------------------------------------------------------- next row f1: host batcher
@pytest.mark.parametrize("mapped", [False, True], ids=["copies", "mapped"])
def test_host_batcher_kinematic_and_callback(oracle, mapped):
    from olympic_hip.batcher import HostBatcher
    from olympic_hip.engine import Engine
    sp = specs.unitree_h1("walk")
    eng = Engine(0).il_configure(sp)
    N = 777
    qpos, qvel, act = h1_synthetic_block(sp, 3, N, seed=31, fall_frac="wide")
    b = HostBatcher(eng, N, n_threads=4, dt=0.01, obs_f64=True).set_mapped(mapped)
    b.qpos[:], b.qvel[:] = qpos[0], qvel[0]
    prev = np.linspace(0.5, 2.0, N)
    b.set_prev(prev)
    q, v = qpos[0].copy(), qvel[0].copy()
    for t in range(3):
        obs, rew, ab = b.step(torch.as_tensor(act[t]).cuda())
        torch.cuda.synchronize()
        q = q + 0.01 * v                                       # built-in kinematic stand-in
        assert np.array_equal(b.qpos, q)
        ref = oracle.il_step(sp, q[None], v[None], None, prev, obs_f64=True)
        assert np.array_equal(host(obs), ref["obs"][0]) and np.array_equal(host(ab), ref["absorbing"][0])
        assert ulp_diff(host(rew), ref["reward"][0]).max() <= 1
        prev = ref["prev"]
    tm = b.last_timing()
    assert tm["physics_s"] >= 0 and tm["ctrl_d2h_s"] > 0
    b.close()
    # a Python physics callback sees the un-normalised, clamped, actuator-ordered controls in fp64
    seen = {}

    def phys(env, ctrl, qp, qv):
        seen[env] = ctrl.copy()
        qp[2] = -1.0                                           # pelvis height out of range: fallen
    b2 = HostBatcher(eng, 5, n_threads=2, physics=phys).set_mapped(mapped)
    a = torch.as_tensor(act[0][:5]).cuda()
    obs, rew, ab = b2.step(a)
    torch.cuda.synchronize()
    exp = np.zeros((5, 11))
    exp[:, sp.act_to_ctrl] = np.clip(act[0][:5].astype(np.float64) * 0.95, -0.95, 0.95)
    assert sorted(seen) == [0, 1, 2, 3, 4] and all(np.array_equal(seen[e], exp[e]) for e in range(5))
    assert host(ab).all() and (host(b2.fall_code) == 1).all()
    b2.close()


def test_il_ctrl_matches_il_step(oracle):
    from olympic_hip.engine import Engine
    import ctypes as C
    from olympic_hip import _ffi
    sp = specs.unitree_h1("walk")
    eng = Engine(0).il_configure(sp)
    N = 1000
    qpos, qvel, act = h1_synthetic_block(sp, 1, N, seed=2)
    a = torch.as_tensor(act[0] * 1.4).cuda().contiguous()
    out = torch.empty((N, 11), dtype=torch.float64, device="cuda")
    eng.ctx.call("oly_il_ctrl", N, _ffi.ptr(a), _ffi.ptr(out), _abi.OUT_CTRL_F64, eng._s())
    ref = oracle.il_step(sp, qpos, qvel, host(a)[None], np.zeros(N), ctrl_f64=True)
    assert np.array_equal(host(out), ref["ctrl"][0])


def test_get_normalization_params_and_block_eval(oracle):
    from olympic_hip.envs import LocoEnvBase
    from olympic_hip.rollout import get_normalization_params
    env = LocoEnvBase.make("UnitreeH1.walk.real", num_envs=64, seed=5).vec
    torch.manual_seed(0)
    policy = torch.nn.Linear(32, 11).cuda()
    seen = []
    orig = env.eng.col_stats

    def spy(x, cs=None):
        seen.append(host(x).copy())
        return orig(x, cs)
    env.eng.col_stats = spy
    mean, std = get_normalization_params(64 * 5, policy, env, 0.1)
    env.eng.col_stats = orig
    states = np.concatenate(seen).astype(np.float64)
    assert states.shape == (320, 32)
    np.testing.assert_allclose(mean, states.mean(0), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(std, np.sqrt(states.var(0) + 1e-8), rtol=1e-8)
    # block evaluation = the [T,N] regime of bench.py through the facade
    sp = env.spec
    qpos, qvel, act = h1_synthetic_block(sp, 6, 64, seed=9, fall_frac="wide")
    prev = np.zeros(64)
    o = env.evaluate_block(torch.as_tensor(qpos).cuda(), torch.as_tensor(qvel).cuda(), torch.as_tensor(act).cuda(),
                           torch.as_tensor(prev).cuda())
    ref = oracle.il_step(sp, qpos, qvel, act, prev, obs_f64=True)
    assert np.array_equal(host(o["obs"]), ref["obs"]) and np.array_equal(host(o["absorbing"]), ref["absorbing"])


# ----------------------------------------------------------------- checkpoint / resume, normalisers
def test_vec_env_state_dict_resume(oracle):
    """Stop after 7 steps, restore into a fresh env, continue: identical to the uninterrupted run."""
    from olympic_hip.envs import ReplayPhysics, VecLocoEnv
    sp = specs.unitree_h1("walk")
    T, N = 14, 512
    qpos, qvel, act = h1_synthetic_block(sp, T, N, seed=5, fall_frac="wide")
    cu = lambda a: torch.as_tensor(a).cuda()

    def run(env, t0, t1):
        out = []
        for t in range(t0, t1):
            o, r, a, _ = env.step(cu(act[t]))
            out.append((host(o), host(r), host(a)))
        return out
    full_env = VecLocoEnv(sp, N, physics=ReplayPhysics(sp, cu(qpos), cu(qvel)), random_start=False)
    full = run(full_env, 0, T)
    a_env = VecLocoEnv(sp, N, physics=ReplayPhysics(sp, cu(qpos), cu(qvel)), random_start=False)
    run(a_env, 0, 7)
    ckpt = a_env.state_dict()
    b_env = VecLocoEnv(sp, N, physics=ReplayPhysics(sp, cu(qpos[7:]), cu(qvel[7:])), random_start=False)
    b_env.load_state_dict(ckpt)
    rest = run(b_env, 7, T)
    for (o1, r1, a1), (o2, r2, a2) in zip(full[7:], rest):
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(a1, a2)
    assert np.array_equal(host(b_env.episode_steps), host(full_env.episode_steps))


def test_runningmeanstd_incremental_equals_batch():
    """The reference's only unit test (rl/envs/normalize.py:210-225) on the device class."""
    from olympic_hip.engine import Engine
    from olympic_hip.normalize import RunningMeanStd
    eng = Engine(0)
    rng = np.random.default_rng(0)
    for shape in ((), (2,)):
        xs = [rng.standard_normal((n,) + shape).astype(np.float32) for n in (3, 4, 5)]
        rms = RunningMeanStd(eng, epsilon=0.0, shape=shape)
        x = np.concatenate(xs, axis=0)
        for xi in xs:
            rms.update(torch.as_tensor(xi).cuda())
        ms1 = [x.mean(axis=0, dtype=np.float64).reshape(-1), x.var(axis=0, dtype=np.float64).reshape(-1)]
        ms2 = [host(rms.mean), host(rms.var)]
        assert np.allclose(ms1[0], ms2[0]) and np.allclose(ms1[1], ms2[1])


def test_normalize_wrapper_on_vec_env():
    """Normalize(VecLocoEnv): online statistics + clipped filter per vec step; frozen afterwards."""
    from olympic_hip.envs import ReplayPhysics, VecLocoEnv
    from olympic_hip.normalize import Normalize
    sp = specs.unitree_h1("walk")
    T, N = 6, 1024
    qpos, qvel, act = h1_synthetic_block(sp, T, N, seed=9, fall_frac="none")
    cu = lambda a: torch.as_tensor(a).cuda()
    env = Normalize(VecLocoEnv(sp, N, physics=ReplayPhysics(sp, cu(qpos), cu(qvel)), random_start=False), clipob=5.0)
    raw = []
    for t in range(T):
        o, r, d, _ = env.step(cu(act[t]))
        raw.append(host(env.venv._obs))
        assert o.shape == (N, sp.n_obs) and float(o.abs().max()) <= 5.0
    allx = np.concatenate(raw).astype(np.float64)
    np.testing.assert_allclose(host(env.ob_rms.mean), allx.mean(0), rtol=1e-6, atol=1e-6)    # eps-count prior 1e-4
    np.testing.assert_allclose(host(env.ob_rms.var), allx.var(0), rtol=1e-5, atol=1e-8)
    want = np.clip((raw[-1] - host(env.ob_rms.mean)) / np.sqrt(host(env.ob_rms.var) + 1e-8), -5, 5).astype(np.float32)
    assert np.array_equal(host(o), want)
    env.online = False
    cnt = env.ob_rms.count
    env.step(cu(act[0]))
    assert env.ob_rms.count == cnt


def test_discriminator_trainer_separates_demo_from_policy():
    """GAIL._fit_discriminator on the device: after a few hundred steps the VAIL reward of
    demonstration-like states exceeds that of policy-like states, and beta stays >= 0."""
    from olympic_hip.engine import Engine
    from olympic_hip.gail import DiscriminatorReward, DiscriminatorTrainer, VariationalDiscriminator, VDBLoss
    eng = Engine(0)
    gen = torch.Generator(device="cuda").manual_seed(0)
    torch.manual_seed(0)
    demo = torch.empty((4000, 32), device="cuda").normal_(0.5, 1.0, generator=gen)
    plcy = lambda n: torch.empty((n, 32), device="cuda").normal_(-0.5, 1.0, generator=gen)
    net = VariationalDiscriminator(32).cuda()
    rew = DiscriminatorReward(eng, net, state_mask=np.arange(32))
    loss = VDBLoss(info_constraint=0.5, lr_beta=1e-5)
    tr = DiscriminatorTrainer(rew, demo.cpu().numpy(), loss, lr=1e-3, n_epochs=1)
    first = tr.fit(plcy(1000), gen)[0]
    for _ in range(150):
        last = tr.fit(plcy(1000), gen)[0]
    assert np.isfinite([first, last]).all() and last < first and loss._beta >= 0
    r_demo = rew(demo[:1000].contiguous(), generator=gen).mean().item()
    r_plcy = rew(plcy(1000), generator=gen).mean().item()
    assert r_demo > r_plcy
    assert float(rew.stand.colstats[0, 0]) == 151 * 2000 + 2 * 1000      # standardiser saw every forward


def test_stickfigure_a3_reference_api(golden):
    """partial(StickFigureA3, algorithm_type=REINFORCEMENT_LEARNING) as train_a3_walk.py uses it:
    robot.mirrored_*, spaces, scalar reset/step with the rewards dict, SymmetricEnv wrapping, and
    the registry path of show_a3_walk.py."""
    from functools import partial
    from olympic_hip.a3 import AlgorithmType, ReplayA3Physics, StickFigureA3
    from olympic_hip.envs import LocoEnvBase
    from olympic_hip.wrappers import SymmetricEnv
    g = golden("a3_task.npz")
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    keys = ("qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel", "rf_vel", "root_pos", "root_quat",
            "head_pos", "ncon", "geom1", "geom2", "force6", "cpos_z")
    e = 3
    blocks = {n: dev(np.swapaxes(g[n], 0, 1)[:, e:e + 1]) for n in keys}

    def make_phys():
        return ReplayA3Physics(blocks, mass=float(g["mass"]))
    env_fn = partial(StickFigureA3, algorithm_type=AlgorithmType.REINFORCEMENT_LEARNING, physics=make_phys())
    env = env_fn()
    assert env.observation_space.shape[0] == 41 and env.action_space.shape[0] == 12 and env.base_obs_len == 41
    assert env.robot.mirrored_obs == g["mirrored_obs"].tolist() if "mirrored_obs" in g.files else len(env.robot.mirrored_obs) == 41
    assert env.robot.clock_inds == [31, 32] and len(env.robot.mirrored_acts) == 12
    # the fixture's reset for env e: same RNG seed, same foot / root poses -> same task state
    np.random.seed(1000 + e)
    env.robot.iteration_count = int(g["iter_count"][e])
    env.vec.iteration_count = env.robot.iteration_count
    env.vec.reset_task([0], g["reset_lfoot"][e:e + 1], g["reset_rfoot"][e:e + 1], g["reset_root_quat"][e:e + 1])
    assert int(env.vec.state["mode"][0]) == int(g["mode"][e]) and int(env.vec.state["phase"][0]) == int(g["phase0"][e])
    K = 12
    for k in range(K):
        obs, total, done, rewards = env.step(np.zeros(12))
        assert obs.shape == (41,) and obs.dtype == np.float64 and isinstance(total, float) and isinstance(done, bool)
        assert list(rewards) == list(StickFigureA3.REWARD_NAMES)
        np.testing.assert_allclose(obs, g["obs"][e, k], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(total, float(g["reward"][e, k]), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose([rewards[n] for n in StickFigureA3.REWARD_NAMES], g["rew6"][e, k], rtol=2e-6, atol=1e-7)
        assert done == bool(g["done"][e, k])
    # reset(): random initial state + WalkingTask.reset, obs of the un-advanced task
    st0 = {k: v.clone() for k, v in env.vec.state.items()}
    o = env._get_obs()
    assert all(torch.equal(st0[k], env.vec.state[k]) for k in st0) and o.shape == (41,)
    ob = env.reset()
    assert ob.shape == (41,) and np.isfinite(ob).all() and int(env.vec.state["reached_frames"][0]) == 0
    ph = int(env.vec.state["phase"][0])
    np.testing.assert_allclose(ob[31:33], [np.sin(2 * np.pi * ph / 88), np.cos(2 * np.pi * ph / 88)], atol=1e-12)
    # SymmetricEnv around it, as the training script builds it
    sym = SymmetricEnv(env_fn, mirrored_obs=env.robot.mirrored_obs, mirrored_act=env.robot.mirrored_acts,
                       clock_inds=env.robot.clock_inds)
    assert sym.observation_space.shape[0] == 41
    m = sym.mirror_clock_observation(torch.as_tensor(ob[None], dtype=torch.float32))
    assert m.shape == (1, 41)
    # registry path
    env2 = LocoEnvBase.make("StickFigureA3.run.real", algorithm_type=AlgorithmType.REINFORCEMENT_LEARNING,
                            physics=make_phys())
    assert isinstance(env2, StickFigureA3)
    with pytest.raises(NotImplementedError):
        StickFigureA3(algorithm_type=AlgorithmType.IMITATION_LEARNING, physics=make_phys())


def test_graphed_rollout_equals_eager_rollout(golden):
    """PPO.sample_vec with the actor/critic forward replayed as one HIP graph: a deterministic
    rollout (no sampling noise) must fill the buffer exactly as the op-by-op rollout does; a
    stochastic one must draw actions with the policy's mean and standard deviation."""
    from olympic_hip.a3 import ReplayA3Physics, VecA3Env
    from olympic_hip.engine import Engine
    from olympic_hip.ppo import PPO, MLPCritic, MLPGaussianActor
    sys_path = __import__("sys").path
    import os
    sys_path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from bench_ppo_iter import synthetic_blocks
    N, T = 256, 24
    gen = torch.Generator(device="cuda").manual_seed(2)
    blocks = synthetic_blocks(N, 16, gen)
    sp = specs.A3Spec(mass=41.5)
    gb = np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32)

    class Env(VecA3Env):
        def __init__(self):
            super().__init__(sp, N, Engine(0), ReplayA3Physics(blocks), gb, 0, 7, 10)
            self.device = self.eng.device
            self.state["seq_len"].fill_(20)
            self.state["mode"].fill_(_abi.MODE_FORWARD)
            self.state["t2"].fill_(1)

        def reset(self, env_mask=None):
            if env_mask is None:
                self.state["phase"].zero_()
                self.physics.k = 0
            else:
                self.state["phase"][env_mask] = 0
            return torch.zeros((N, 41), device="cuda")
    torch.manual_seed(0)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    ppo = PPO.__new__(PPO)
    ppo.use_device_rollout = False               # the per-step host loop; the device rollout has its own test
    ppo.fused_forward = False                    # the modules' torch forward (eager vs its graph replay)
    bufs = []
    for graph in (False, True):
        ppo.use_graph_rollout = graph
        bufs.append(ppo.sample_vec(Env(), pi, vf, T, 10, deterministic=True))
    a, b = bufs
    for name in ("states", "actions", "rewards", "values", "next_values", "flags"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert int((a.flags & _abi.FLAG_LAST).bool().sum()) >= 2 * N           # time-limit cuts happened
    ppo.use_graph_rollout = True
    s = ppo.sample_vec(Env(), pi, vf, T, 10, deterministic=False, anneal=0.5)
    mu = pi(s.states.reshape(-1, 41)).reshape(T, N, 12)
    z = (s.actions - mu) / (float(pi.fixed_std) * 0.5)
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02   # N(mu, (std * anneal)^2)
    assert torch.equal(s.values, vf(s.states.reshape(-1, 41)).reshape(T, N))
    # the default host loop evaluates both MLPs with ONE K11 launch per vec step: same rollout up to the
    # summation order of the matrix products
    ppo.fused_forward = True
    f = ppo.sample_vec(Env(), pi, vf, T, 10, deterministic=True)
    assert torch.equal(f.states, a.states) and torch.equal(f.flags, a.flags) and torch.equal(f.rewards, a.rewards)
    assert torch.allclose(f.actions, a.actions, rtol=1e-5, atol=1e-6) and torch.allclose(f.values, a.values, rtol=1e-5, atol=1e-6)


def test_examples_run(tmp_path):
    """The two example scripts (the reference's play_walking_trajectory and train_a3_walk command
    lines) run end to end on the GPU with small settings."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "play_walking_trajectory.py"), "UnitreeH1.walk.real",
                        "--episodes", "1", "--steps", "50"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "replayed 50 steps" in r.stdout, r.stdout + r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "train_a3_walk.py"), "train", "--num_procs", "64",
                        "--n_itr", "2", "--max_traj_len", "20", "--minibatch_size", "256", "--input_norm_steps", "640",
                        "--eval_freq", "2", "--logdir", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "done: 2 iterations" in r.stdout, r.stdout + r.stderr[-3000:]
    assert os.path.exists(os.path.join(str(tmp_path), "actor.pt")) and os.path.exists(os.path.join(str(tmp_path), "eval.txt"))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "vail_discriminator_step.py"), "--num_envs", "256",
                        "--steps", "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "discriminator reward mean" in r.stdout and " std 1.0000" in r.stdout, r.stdout + r.stderr[-3000:]


@pytest.mark.parametrize("mapped,compact", [(False, False), (True, False), (True, True)], ids=["copies", "mapped", "compact"])
def test_a3_host_batcher_replays_golden_sequence(golden, mapped, compact):
    """oly_a3_batcher_*: host thread pool + pinned staging + one H2D copy + K3 + K2 per step.  A
    Python physics callback plays the fixture's recorded readback (what a MuJoCo callback would
    write after its PD substeps) and checks the PD targets it receives; results equal the
    reference's WalkingTask / get_obs outputs."""
    from olympic_hip.a3 import VecA3Env, ReplayA3Physics
    from olympic_hip.batcher import A3HostBatcher
    from olympic_hip.engine import Engine
    g = golden("a3_task.npz")
    E, K = g["phase"].shape
    sp = specs.A3Spec(mass=float(g["mass"]))
    eng = Engine(0)
    env = VecA3Env(sp, E, eng, ReplayA3Physics({"qpos": torch.zeros((1, E, 25), dtype=torch.float64, device="cuda")}),
                   g["geom_bodyid"], int(g["floor_body"]), int(g["rfoot_body"]), int(g["lfoot_body"]))
    for e in range(E):
        np.random.seed(1000 + e)
        env.iteration_count = int(g["iter_count"][e])
        env.reset_task([e], g["reset_lfoot"][e:e + 1], g["reset_rfoot"][e:e + 1], g["reset_root_quat"][e:e + 1])
    C_ = g["geom1"].shape[-1]
    step_no = {"k": 0}
    seen_targets = np.zeros((E, 12))

    def physics(e, target, slots):
        k = step_no["k"]
        seen_targets[e] = target
        for n in ("qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel", "rf_vel", "root_pos", "root_quat",
                  "head_pos", "geom1", "geom2", "force6", "cpos_z"):
            slots[n][...] = g[n][e, k]
        slots["ncon"][0] = g["ncon"][e, k]
    b = A3HostBatcher(eng, E, C_, physics, n_threads=2, obs_f64=True).set_mapped(mapped).set_compact(compact)
    act = torch.zeros((E, 12), device="cuda")
    for k in range(K):
        step_no["k"] = k
        obs, rew, done, rew6 = b.step(act, env.state)
        torch.cuda.synchronize()
        assert np.array_equal(seen_targets, np.tile(sp.motor_offset, (E, 1)))        # zero action -> offsets
        assert np.array_equal(env.state["phase"].cpu().numpy(), g["phase"][:, k])
        assert np.array_equal(done.cpu().numpy().astype(bool), g["done"][:, k])
        np.testing.assert_allclose(obs.cpu().numpy(), g["obs"][:, k], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(rew.cpu().numpy(), g["reward"][:, k], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(rew6.cpu().numpy(), g["rew6"][:, k], rtol=2e-6, atol=1e-7)
    t = b.last_timing()
    assert t["physics_s"] > 0 and t["h2d_kernels_enqueue_s"] > 0
    # slots() / upload() / with_physics=False: evaluate a hand-written state without stepping the physics
    s0 = b.slots(0)
    assert s0["qpos"].shape == (25,) and s0["force6"].shape == (C_, 6)
    keep = {k2: v.clone() for k2, v in env.state.items()}
    b.upload()
    o2, *_ = b.step(None, env.state, with_physics=False)
    assert np.isfinite(o2.cpu().numpy()).all()
    for k2, v in keep.items():
        env.state[k2].copy_(v)
    b.close()


@pytest.mark.parametrize("packed", [False, True])
def test_host_batcher_with_foot_force_contacts(oracle, packed):
    """IL batcher with use_foot_forces: the physics callback writes W = 10 contact snapshots per
    env and control step; obs = [joint obs, window-mean ground forces / 1000].  packed: the worker
    threads reduce the slots to per-pair first-contact forces on the host (dense rows over PCIe,
    oly_il_grf_window on the device): identical observations."""
    from olympic_hip.batcher import HostBatcher
    from olympic_hip.engine import Engine
    sp = specs.unitree_h1("walk").with_foot_forces("UnitreeH1")
    eng = Engine(0).il_configure(sp)
    N, W, Cc, K = 96, 10, 8, 4
    rng = np.random.default_rng(6)
    qpos, qvel, act = h1_synthetic_block(sp, K, N, seed=2, fall_frac="wide")
    con = dict(ncon=rng.integers(0, Cc + 1, (K, W, N)).astype(np.int32),
               geom1=np.zeros((K, W, N, Cc), np.int32),
               geom2=rng.choice([12, 22, 5, 30], size=(K, W, N, Cc)).astype(np.int32),
               force6=rng.normal(0, 300, (K, W, N, Cc, 6)))
    step = {"k": 0}

    def physics(e, ctrl, q, v, c):
        k = step["k"]
        q[:], v[:] = qpos[k, e], qvel[k, e]
        c["ncon"][:] = con["ncon"][k, :, e]
        c["geom1"][:] = con["geom1"][k, :, e]
        c["geom2"][:] = con["geom2"][k, :, e]
        c["force6"][:] = con["force6"][k, :, e]
    b = HostBatcher(eng, N, n_threads=2, obs_f64=True)
    with pytest.raises(Exception, match="oly_batcher_enable_contacts"):
        b.step(torch.zeros((N, sp.n_act), device="cuda"))           # foot-force model without contact staging
    eng.grf_configure(sp.geom_group, sp.grf_pairs)
    b.enable_contacts(W, Cc, physics, packed=packed)
    prev = rng.normal(1.25, 0.3, N)
    b.set_prev(prev)
    means = np.stack([oracle.il_ground_forces(sp.geom_group, sp.grf_pairs, con["ncon"][k], con["geom1"][k],
                                              con["geom2"][k], con["force6"][k])[1] for k in range(K)])
    ref = oracle.il_step(sp, qpos, qvel, None, prev, grf_mean=means, obs_f64=True)
    for k in range(K):
        step["k"] = k
        obs, rew, ab = b.step(torch.as_tensor(act[k]).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(obs.cpu().numpy(), ref["obs"][k])
        assert np.array_equal(ab.cpu().numpy(), ref["absorbing"][k])
        assert ulp_diff(rew.cpu().numpy(), ref["reward"][k]).max() <= 1
    # raw data.ncon beyond the staged slots (the reference scans every contact, UnitreeH1.py:113-123): exact while both
    # sensor pairs have their first contact among the slots, an error (never a silently dropped force) otherwise
    k = K - 1
    step["k"] = k
    con["geom2"][k, :, 5, :2] = [22, 12]                               # env 5: foot_r, foot_l in the first two slots
    con["ncon"][k, 3, 5] = Cc + 7
    means_k = oracle.il_ground_forces(sp.geom_group, sp.grf_pairs, con["ncon"][k], con["geom1"][k], con["geom2"][k],
                                      con["force6"][k], want_overflow=True)
    assert not means_k[2].any()
    b.set_prev(prev)
    ref_k = oracle.il_step(sp, qpos[k:k + 1], qvel[k:k + 1], None, prev, grf_mean=means_k[1][None], obs_f64=True)
    obs, rew, ab = b.step(torch.as_tensor(act[k]).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(obs.cpu().numpy(), ref_k["obs"][0])
    con["geom2"][k, 3, 7, :] = 22                                      # env 7, substep 3: no foot_l contact in any slot
    con["ncon"][k, 3, 7] = Cc + 1
    with pytest.raises(Exception, match="environment 7 has more contacts than the 8 staged slots"):
        b.step(torch.as_tensor(act[k]).cuda())
    b.close()
