"""K10, the one-launch vec step of the PPO rollout loop (rl/algos/ppo.py:169-196), and the
device-resident rollout built on it: against the oracle's row-by-row restatement, against the
separate K3 + K2 kernels, and graph replay against the eager loop."""
import os

import numpy as np
import pytest
import torch

from olympic_hip import _abi, specs
from olympic_hip.synthetic import A3_FLOOR_BODY, A3_GEOM_BODYID, A3_LFOOT_BODY, A3_RFOOT_BODY, a3_synthetic_blocks
from olympic_hip.vecstep import REC, draw_reset_records

pytestmark = pytest.mark.gpu
CONTACT = (A3_GEOM_BODYID, A3_FLOOR_BODY, A3_RFOOT_BODY, A3_LFOOT_BODY)


@pytest.fixture(scope="module")
def eng(golden):
    from olympic_hip.engine import Engine
    e = Engine(0)
    e.a3_configure(specs.A3Spec(mass=41.5), golden("a3_task.npz")["clock_lut"])
    e.contact_configure(*CONTACT)
    return e


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _host_rollout(spec, N, K, T, max_traj_len, depth, seed, det=False, p_bad=0.05, p_low=0.02):
    """Everything oly_a3_vec_step touches, as numpy arrays (the oracle works on these in place)."""
    rng = np.random.default_rng(seed)
    rs = np.random.RandomState(seed)
    blocks = a3_synthetic_blocks(N, K, seed=seed, p_bad=p_bad, p_low=p_low)
    pool = draw_reset_records(rs, N * depth, spec, iter_count=6000)
    slots = T // max_traj_len + 2
    nobs, nu = spec.n_obs, spec.nu
    state = dict(phase=np.zeros(N, np.int32), t1=np.zeros(N, np.int32), t2=np.zeros(N, np.int32),
                 reached_frames=np.zeros(N, np.int32), target_reached=np.zeros(N, np.uint8),
                 mode=np.full(N, _abi.MODE_STANDING, np.int32), seq_len=np.ones(N, np.int32),
                 sequence=np.zeros((N, _abi.OLY_MAX_SEQ, 4)), goal=np.zeros((N, 8)))
    ro = dict(T=T, max_traj_len=max_traj_len, deterministic=det, side_slots=slots, pool_depth=depth,
              mu=np.zeros((N, nu), np.float32), value=np.zeros(N, np.float32),
              scale=None if det else rng.uniform(0.05, 0.4, nu).astype(np.float32),
              eps=None if det else rng.normal(0, 1, (T, N, nu)).astype(np.float32),
              state=np.zeros((N, nobs), np.float32), pd_target=np.zeros((N, nu)),
              buf_states=np.zeros((T, N, nobs), np.float32), buf_actions=np.zeros((T, N, nu), np.float32),
              buf_rewards=np.zeros((T, N)), buf_values=np.zeros((T, N), np.float32),
              buf_flags=np.zeros((T, N), np.uint8), buf_rew6=np.zeros((T, N, 6), np.float32),
              traj_len=np.zeros(N, np.int32), side_obs=np.zeros((N * slots, nobs), np.float32),
              side_t=np.full(N * slots, -1, np.int32), side_count=np.zeros(N, np.int32),
              pool=pool.view(np.uint8).reshape(-1).copy(), pool_count=np.zeros(N, np.int32),
              ctr=np.array([0, 3], np.int32))
    return blocks, state, ro, rng


def _to_device(eng, blocks, state, ro):
    N = len(state["phase"])
    d_blocks = {k: dev(v) for k, v in blocks.items()}
    d_state = {k: dev(v) for k, v in state.items()}
    d_ro = {k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in ro.items()}
    ctr = torch.zeros(eng.a3_vec_ctr_len(N), dtype=torch.int32, device="cuda")
    ctr[0:-2:2], ctr[1:-2:2] = int(ro["ctr"][0]), int(ro["ctr"][1])      # the last pair is (overrun mark, pad)
    d_ro["ctr"] = ctr
    return d_blocks, d_state, d_ro


@pytest.mark.parametrize("N,T,max_len,det", [(300, 9, 3, False), (4096, 6, 4, False), (37, 7, 100, True)])
def test_vec_step_vs_oracle(eng, golden, oracle, N, T, max_len, det):
    """RESET_ALL + T steps, the kernel and the oracle side by side on the same inputs.  Integer
    state, flags, cursors, counters, actions (f32 mul + add), PD targets: bit-exact.  Float64 values
    that pass through sin / cos / tan / exp / atan2 (device libm vs glibc): 1e-11 relative;
    observations are those values narrowed to float32: at most one float32 ulp."""
    spec = specs.A3Spec(mass=41.5)
    lut = golden("a3_task.npz")["clock_lut"]
    blocks, state, ro, rng = _host_rollout(spec, N, 5, T, max_len, 2, seed=N + T, det=det)
    d_blocks, d_state, d_ro = _to_device(eng, blocks, state, ro)
    launch = eng.a3_vec_prepare(d_blocks, d_state, d_ro)
    h = lambda t_: t_.cpu().numpy()

    def compare(step):
        for k in ("phase", "t1", "t2", "reached_frames", "target_reached", "mode", "seq_len"):
            assert np.array_equal(h(d_state[k]), state[k]), (step, k)
        np.testing.assert_allclose(h(d_state["sequence"]), state["sequence"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(h(d_state["goal"]), state["goal"], rtol=1e-11, atol=1e-12)
        for k in ("traj_len", "side_count", "side_t", "pool_count", "buf_flags"):
            assert np.array_equal(h(d_ro[k]), ro[k]), (step, k)
        assert np.array_equal(h(d_ro["buf_actions"]), ro["buf_actions"])
        assert np.array_equal(h(d_ro["pd_target"]), ro["pd_target"])
        assert np.array_equal(h(d_ro["buf_values"]), ro["buf_values"])
        np.testing.assert_allclose(h(d_ro["buf_rewards"]), ro["buf_rewards"], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(h(d_ro["buf_rew6"]), ro["buf_rew6"], rtol=2e-6, atol=1e-7)
        for k in ("state", "buf_states", "side_obs"):
            a, b = h(d_ro[k]), ro[k]
            assert np.abs(a - b).max() <= np.spacing(np.float32(1.0)) * max(1.0, np.abs(b).max()), (step, k)
        c = h(d_ro["ctr"]).reshape(-1, 2)
        assert (c[:-1, 0] == ro["ctr"][0]).all() and (c[:-1, 1] == ro["ctr"][1]).all() and c[-1, 0] == 0

    launch(_abi.VSTEP_RESET_ALL)
    oracle.a3_vec_step(spec, lut, CONTACT, blocks, state, ro, _abi.VSTEP_RESET_ALL)
    compare("reset")
    assert (ro["pool_count"] == 1).all() and ro["ctr"].tolist() == [0, 3]
    for t in range(T):
        mu = rng.normal(0, 0.3, (N, spec.nu)).astype(np.float32)
        val = rng.normal(0, 1, N).astype(np.float32)
        ro["mu"][...], ro["value"][...] = mu, val
        # the observation the "policy" sees is the previous step's output: keep both sides on the oracle's
        d_ro["state"].copy_(dev(ro["state"]))
        launch(0, dev(mu), dev(val))
        oracle.a3_vec_step(spec, lut, CONTACT, blocks, state, ro, 0)
        compare(t)
    assert ro["ctr"].tolist() == [T, 3 + T]
    fl = ro["buf_flags"]
    assert (fl[-1] & _abi.FLAG_LAST).all()                       # the block end cuts every environment
    assert ((fl & _abi.FLAG_LAST) != 0).sum() > N                # and time limits / dones cut more
    assert (ro["side_count"] <= ro["side_slots"]).all()
    cut_not_done = ((fl & _abi.FLAG_LAST) != 0) & ((fl & _abi.FLAG_ABSORBING) == 0)
    assert cut_not_done.sum() == (ro["side_t"] >= 0).sum()       # one bootstrap row per non-terminal cut


def test_vec_step_reset_matches_reference_reset(eng, golden):
    """The device-side env.reset() (OLY_VSTEP_RESET_ALL) on the reference-generated a3_reset fixture: same
    checks as the oracle's (test_oracle_golden.py), through the kernel; and the host path of the facade
    (reset_task + _get_obs) on the same fixture."""
    from helpers import a3_reset_fixture_rollout, check_a3_reset_fixture
    g = golden("a3_reset.npz")
    spec = specs.A3Spec(mass=41.5)
    blocks, state, ro = a3_reset_fixture_rollout(g, spec)
    d_blocks, d_state, d_ro = _to_device(eng, blocks, state, ro)
    eng.a3_vec_prepare(d_blocks, d_state, d_ro)(_abi.VSTEP_RESET_ALL)
    check_a3_reset_fixture(g, {k: v.cpu().numpy() for k, v in d_state.items()}, d_ro["state"].cpu().numpy())
    # host path: WalkingTaskReset with the fixture's seeds + the facade's reset observation
    from olympic_hip.a3 import AlgorithmType, ReplayA3Physics, StickFigureA3
    keys = ("qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel", "rf_vel", "root_pos", "root_quat",
            "head_pos", "ncon", "geom1", "geom2", "force6", "cpos_z")
    for e in (0, 5, 17, 40):
        one = {k: d_blocks[k][:, e:e + 1].contiguous() for k in keys}
        env = StickFigureA3(algorithm_type=AlgorithmType.REINFORCEMENT_LEARNING, physics=ReplayA3Physics(one, mass=41.5))
        env.vec.iteration_count = int(g["iter_count"][e])
        np.random.seed(int(g["seed"][e]))
        env.vec.reset_task([0], g["lfoot"][e:e + 1], g["rfoot"][e:e + 1], g["root_quat"][e:e + 1])
        st = {k: v.cpu().numpy() for k, v in env.vec.state.items()}
        assert int(st["mode"][0]) == int(g["mode"][e]) and int(st["phase"][0]) == int(g["phase"][e])
        np.testing.assert_allclose(st["sequence"][0], g["sequence"][e], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(env._get_obs(), g["obs"][e], rtol=1e-11, atol=1e-12)     # float64 facade


def test_vec_step_more_than_sixteen_contact_slots(eng, golden, oracle):
    """C = 20 contact slots per environment: the second 16-slot pass loads on demand; still the oracle's
    numbers (contact counts and force sums cross the pass boundary)."""
    spec = specs.A3Spec(mass=41.5)
    lut = golden("a3_task.npz")["clock_lut"]
    N, T, C = 77, 4, 20
    blocks, state, ro, rng = _host_rollout(spec, N, 3, T, 3, 2, seed=9, det=True)
    b20 = a3_synthetic_blocks(N, 3, seed=9, C=C, p_bad=0.3, p_low=0.02)
    b20["ncon"] = rng.integers(0, C + 1, b20["ncon"].shape).astype(np.int32)      # up to all 20 slots in use
    blocks = b20
    d_blocks, d_state, d_ro = _to_device(eng, blocks, state, ro)
    launch = eng.a3_vec_prepare(d_blocks, d_state, d_ro)
    launch(_abi.VSTEP_RESET_ALL)
    oracle.a3_vec_step(spec, lut, CONTACT, blocks, state, ro, _abi.VSTEP_RESET_ALL)
    for t in range(T):
        d_ro["state"].copy_(dev(ro["state"]))
        launch(0)
        oracle.a3_vec_step(spec, lut, CONTACT, blocks, state, ro, 0)
        assert np.array_equal(d_ro["buf_flags"].cpu().numpy(), ro["buf_flags"])
        np.testing.assert_allclose(d_ro["buf_rewards"].cpu().numpy(), ro["buf_rewards"], rtol=1e-11, atol=1e-13)
        for k in ("phase", "t1", "t2", "reached_frames", "mode", "seq_len"):
            assert np.array_equal(d_state[k].cpu().numpy(), state[k]), k
    assert (blocks["ncon"] > 16).any()


def test_vec_step_equals_the_separate_kernels(eng, golden):
    """The fused launch against oly_contact_reduce + oly_a3_step + oly_a3_pd_target +
    oly_rollout_cuts on the same readback row: the same libm entry points on the same arguments, so
    observations, rewards, task state and flags are bit-identical."""
    spec = specs.A3Spec(mass=41.5)
    N, T = 1000, 4
    blocks, state, ro, rng = _host_rollout(spec, N, 4, T, 100, 2, seed=5, det=True, p_bad=0.1, p_low=0.05)
    d_blocks, d_state, d_ro = _to_device(eng, blocks, state, ro)
    launch = eng.a3_vec_prepare(d_blocks, d_state, d_ro)
    launch(_abi.VSTEP_RESET_ALL)
    traj_len = torch.zeros(N, dtype=torch.int32, device="cuda")
    n_cut = torch.zeros(1, dtype=torch.int32, device="cuda")
    for t in range(T - 1):           # the last step resets nobody but cuts everybody: covered above
        kk = (3 + t) % 4
        ref_state = {k: v.clone() for k, v in d_state.items()}
        cr = eng.contact_reduce(d_blocks["ncon"][kk], d_blocks["geom1"][kk], d_blocks["geom2"][kk],
                                d_blocks["force6"][kk], d_blocks["cpos_z"][kk], want_idx=False)
        kin = {k: d_blocks[k][kk] for k in ("qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel",
                                            "rf_vel", "root_pos", "root_quat", "head_pos")}
        kin.update({k: cr[k] for k in ("grf_l", "grf_r", "min_z", "n_r", "n_l", "bad")})
        o = eng.a3_step(kin, ref_state)
        mu = torch.randn((N, spec.nu), device="cuda")
        flags = torch.zeros(N, dtype=torch.uint8, device="cuda")
        eng.rollout_cuts(o["done"], traj_len, flags, n_cut, 100, False)
        launch(0, mu, torch.zeros(N, device="cuda"))
        assert torch.equal(d_ro["buf_actions"][t], mu)
        assert torch.equal(d_ro["pd_target"], eng.a3_pd_target(mu))
        assert torch.equal(d_ro["buf_flags"][t], flags)
        assert torch.equal(d_ro["buf_rewards"][t].float(), o["reward"])       # K2 narrows its f64 total to f32
        assert torch.equal(d_ro["buf_rew6"][t], o["rew6"])
        keep = ~(flags & _abi.FLAG_LAST).bool()                                # environments that were not reset
        assert int(keep.sum()) > N // 2 and int((~keep).sum()) > 5
        assert torch.equal(d_ro["state"][keep], o["obs"][keep])
        for k in ("phase", "t1", "t2", "reached_frames", "target_reached", "goal"):
            assert torch.equal(d_state[k][keep], ref_state[k][keep]), k
        # reset environments: un-advanced task, goal steps zero, robot part of the observation unchanged
        rs_ = ~keep
        assert torch.equal(d_ro["state"][rs_][:, :31], o["obs"][rs_][:, :31])
        assert not d_ro["state"][rs_][:, 33:].any() and not d_state["goal"][rs_].any()
        assert not d_state["reached_frames"][rs_].any() and not d_state["t1"][rs_].any()
        traj_len = d_ro["traj_len"].clone()


def _make_env(N, K, seed, p_bad=0.02, C=16):
    from olympic_hip.a3 import ReplayA3Physics, VecA3Env
    from olympic_hip.engine import Engine
    host = a3_synthetic_blocks(N, K, seed=seed, C=C, p_bad=p_bad, p_low=0.01)
    if C != 16:            # use every slot count up to C (the default Poisson(4) rarely passes 10)
        host["ncon"] = np.random.default_rng(seed).integers(0, C + 1, host["ncon"].shape).astype(np.int32)
    blocks = {k: dev(v) for k, v in host.items()}
    env = VecA3Env(specs.A3Spec(mass=41.5), N, Engine(0), ReplayA3Physics(blocks), *CONTACT, rs=np.random.RandomState(seed))
    env.device = env.eng.device
    return env


@pytest.mark.parametrize("deterministic", [True, False])
def test_device_rollout_graph_replay_equals_eager_loop(deterministic):
    """The same launches replayed from HIP graphs (8 steps per graph, a 3-step eager tail) must fill
    the rollout buffer exactly like the op-by-op loop: states, actions, float64 rewards, values,
    next_values, flags, task state; and the buffer obeys the rollout's invariants."""
    from olympic_hip.ppo import MLPCritic, MLPGaussianActor
    N, T, max_len = 512, 27, 10
    torch.manual_seed(0)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    out = []
    for graph in (False, True):
        env = _make_env(N, 7, seed=3)
        torch.manual_seed(11)                                     # the action noise block
        buf = env.device_rollout(pi, vf, T, max_len, deterministic=deterministic, anneal=0.7, graph=graph,
                                 persistent=False)
        out.append((buf, {k: v.clone() for k, v in env.state.items()}, dict(env._dev_rollout.last_info)))
    (a, sa, ia), (b, sb, ib) = out
    for name in ("states", "actions", "rewards", "values", "next_values", "flags"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert all(torch.equal(sa[k], sb[k]) for k in sa) and ia == ib
    assert a.rewards.dtype == torch.float64
    last = (a.flags & _abi.FLAG_LAST).bool()
    absorbing = (a.flags & _abi.FLAG_ABSORBING).bool()
    assert bool(last[-1].all()) and int(last.sum()) >= N * (T // max_len) and not bool((absorbing & ~last).any())
    assert ia["resets"] == N + int(last[:-1].sum())               # env.reset() of all, then one per cut
    assert ia["side_rows"] == int((last & ~absorbing).sum())
    # V(s_{t+1}): the next row's value where the episode goes on
    assert torch.equal(a.next_values[:-1][~last[:-1]], a.values[1:][~last[:-1]])
    # the policy really produced the stored actions / values from the stored states
    with torch.no_grad():
        mu = pi(a.states.reshape(-1, 41)).reshape(T, N, 12)
        assert torch.allclose(a.values, vf(a.states.reshape(-1, 41)).reshape(T, N), rtol=1e-5, atol=1e-6)
    if deterministic:
        assert torch.allclose(a.actions, mu, rtol=1e-5, atol=1e-6)
    else:
        z = (a.actions - mu) / (float(pi.fixed_std) * 0.7)
        assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02


@pytest.mark.parametrize("N,T,max_len,det,K,C", [(512, 27, 10, False, 7, 16), (500, 13, 4, True, 5, 16),
                                                 (37, 40, 100, False, 3, 16), (4096, 24, 8, False, 32, 16),
                                                 (1, 9, 3, False, 4, 16), (300, 11, 5, False, 4, 20),
                                                 (100, 9, 4, False, 3, 5), (20007, 6, 3, False, 4, 16),
                                                 (4096, 400, 400, False, 32, 16),      # BASELINE config 3, whole
                                                 (4096, 400, 150, True, 32, 16)])
def test_persistent_rollout_equals_the_two_kernel_loop(N, T, max_len, det, K, C):
    """K13 (oly_a3_rollout_persistent: ONE launch for the whole rollout, an 8-wave workgroup owns 16 environments through
    all T steps - four waves the forward, four the environment step - observations / task state never leave the CU)
    against T rounds of K11 + K10 from the same start: every
    buffer, the bootstrap side list, the final task state, the next observation, the PD targets, the pool cursors and
    the device counters are BIT-identical (integer, float32 and float64 alike), for full and ragged tiles, with the
    actor's input normalisation on, two rollouts in a row (the second continues the replay row and the pools), and for
    20 contact slots (a second, on-demand pass over the slots) and 5 (fewer slots than lanes), and for more workgroups
    than the chip holds at once (20007 environments = 1251 workgroups), and at BASELINE config 3's own size."""
    from olympic_hip.ppo import MLPCritic, MLPGaussianActor
    torch.manual_seed(5)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    g = torch.Generator().manual_seed(1)
    pi.obs_mean = (0.1 * torch.randn(41, generator=g)).cuda()
    pi.obs_std = (1.0 + 0.2 * torch.rand(41, generator=g)).cuda()
    if N % 2:              # the critic's input normalisation too (critic.py:63-64 applies it in eval mode only)
        vf.obs_mean = (0.05 * torch.randn(41, generator=g)).cuda()
        vf.obs_std = (1.0 + 0.1 * torch.rand(41, generator=g)).cuda()
        vf.eval()
    out = []
    for persistent in (False, True):
        env = _make_env(N, K, seed=3, p_bad=0.03, C=C)
        snaps = []
        for it in range(2):
            torch.manual_seed(11 + it)                            # the action noise block
            env.device_rollout(pi, vf, T, max_len, deterministic=det, anneal=0.7, graph=False, persistent=persistent)
            r = env._dev_rollout
            named = dict(states=r.buf.states, actions=r.buf.actions, means=r.buf.mu, rewards=r.buf.rewards, values=r.buf.values,
                         next_values=r.buf.next_values, flags=r.buf.flags, state_obs=r.state_obs, pd_target=r.pd_target,
                         traj_len=r.traj_len, side_obs=r.side_obs, side_t=r.side_t, side_count=r.side_count,
                         pool_count=r.pool_count, ctr=r.ctr, mu=r._fw.outputs(N)[0], value=r._fw.outputs(N)[1])
            named.update({"st_" + k: v for k, v in env.state.items()})
            snaps.append(({k: v.clone() for k, v in named.items()}, dict(r.last_info), env.physics.k))
        out.append(snaps)
    for (a, ia, ka), (b, ib, kb) in zip(*out):
        for name in a:
            assert a[name].dtype == b[name].dtype and torch.equal(a[name], b[name]), name
        assert ia == ib and ka == kb
    # the stored means are the old policy's forward over the stored observations, to the bit: what the update phase
    # takes as old_mu instead of running the old policy over the buffer again (ppo.KernelUpdate.begin)
    r = env._dev_rollout
    again = torch.empty((T * N, 12), dtype=torch.float32, device="cuda")
    env.eng.mlp_forward2(r.buf.states.reshape(T * N, -1), r._fw.packed_a, 12, again, normalize_a=True)
    assert r.buf.mu_from_fused_forward and torch.equal(r.buf.mu.reshape(T * N, 12), again)
    last = (out[1][0][0]["flags"] & _abi.FLAG_LAST).bool()
    assert bool(last[-1].all()) and (N < 32 or int(last[:-1].sum()) > 0), "the case must exercise device-side resets"
    assert env._dev_rollout._fw.norm_c == bool(N % 2)


@pytest.mark.parametrize("N,T,max_len,det,normalize", [(70, 12, 5, False, True), (16, 9, 3, True, False), (261, 20, 6, False, True)])
def test_persistent_rollout_against_the_oracle_directly(eng, oracle, N, T, max_len, det, normalize):
    """K13 pinned to the oracle itself (not only to the K11 + K10 loop): one launch for all T steps against a loop of
    oly_mlp_forward_cpu x 2 + oly_a3_vec_step_cpu, resets on, stochastic actions from a supplied noise block, the
    actor's input normalisation on (rl/algos/ppo.py:169-196; bars in helpers.check_persistent_rollout_against_oracle)."""
    from helpers import check_persistent_rollout_against_oracle
    info = check_persistent_rollout_against_oracle(eng, oracle, N=N, T=T, max_len=max_len, det=det, normalize=normalize, seed=N + T)
    assert info["resets"] > 0 and info["cuts"] > N and info["bootstrap_rows"] > 0     # the case exercises the cut rules


def test_a_step_past_the_last_row_writes_nothing_and_is_reported(eng, golden, oracle):
    """The step index lives on the device: a launch with t outside [0, T) (one replay too many, counters never
    rewound) must not write beyond the rollout buffers.  K10 and K13 touch nothing, advance nothing and leave a sticky
    mark behind the counters; the oracle refuses the same call; A3DeviceRollout turns the mark into an error."""
    from olympic_hip._ffi import OlyError
    from olympic_hip.ppo import MLPCritic, MLPGaussianActor
    h = lambda t_: t_.cpu().numpy()
    g = golden("a3_task.npz")
    spec = specs.A3Spec(mass=41.5)
    eng.a3_configure(spec, g["clock_lut"])
    eng.contact_configure(*CONTACT)
    N, T = 40, 3
    blocks, state, ro, _ = _host_rollout(spec, N, 4, T, 2, 2, seed=9)
    d_blocks, d_state, d_ro = _to_device(eng, blocks, state, ro)
    launch = eng.a3_vec_prepare(d_blocks, d_state, d_ro)
    launch(_abi.VSTEP_RESET_ALL)
    for _ in range(T):
        launch(0)
    torch.cuda.synchronize()
    snap = {k: v.clone() for k, v in list(d_ro.items()) + list(d_state.items()) if isinstance(v, torch.Tensor)}
    launch(0)                                                        # t == T
    torch.cuda.synchronize()
    c = h(d_ro["ctr"]).reshape(-1, 2)
    assert c[-1, 0] == 1 and (c[:-1, 0] == T).all()
    d_ro["ctr"][-2] = 0
    for k, v in snap.items():
        cur = d_ro[k] if k in d_ro else d_state[k]
        assert torch.equal(cur, v), k
    orc_ro = dict(ro, ctr=np.array([T, 0], np.int32))
    with pytest.raises(Exception):
        oracle.a3_vec_step(spec, g["clock_lut"], CONTACT, blocks, state, orc_ro, 0)
    d_ro["ctr"][0:-2:2] = -1                                          # K13 with a negative counter
    pa = torch.zeros(eng._mlp_floats(41, 12), device="cuda")
    pc = torch.zeros(eng._mlp_floats(41, 1), device="cuda")
    launch.persistent(pa, True, pc, False)
    torch.cuda.synchronize()
    assert int(d_ro["ctr"][-2]) == 1
    # the facade: a rollout whose counters were tampered with raises instead of returning a corrupt buffer
    env = _make_env(64, 4, seed=1)
    torch.manual_seed(0)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    env.device_rollout(pi, vf, 6, 3, graph=False, persistent=False)
    r = env._dev_rollout
    r.launch(0, *r._fw.outputs(64))                                   # a seventh step into a six-row buffer
    with pytest.raises(OlyError, match="outside"):
        r._finalize(r._fw, vf)


def test_torch_forward_graphs_do_not_outlive_their_rollout():
    """with the modules' own torch forward (any policy that is not 2 x 256 relu) the captured rollout graph
    holds torch GEMM nodes; an update phase between two rollouts (synchronize, fresh allocations >= 1 MB, kernel
    writes) is the sequence after which such graphs were seen to replay wrongly (profiles/r02/graph_drift).  The
    graphs are therefore released with their rollout: a second, re-captured rollout after an update-like phase equals
    the eager loop bit for bit, and no graph stays cached."""
    from olympic_hip.ppo import MLPCritic, MLPGaussianActor
    from olympic_hip.vecstep import TorchForward
    N, T, max_len = 256, 19, 7
    torch.manual_seed(0)
    pi, vf = MLPGaussianActor(41, 12, layers=(128, 96)).cuda(), MLPCritic(41, layers=(128, 96)).cuda()
    out = []
    for graph in (False, True):
        env = _make_env(N, 5, seed=6)
        fw = TorchForward(pi, vf)
        for it in range(2):
            torch.manual_seed(21 + it)
            buf = env.device_rollout(pi, vf, T, max_len, anneal=0.9, graph=graph, forward=fw)
            assert not env._dev_rollout._graphs, "torch-forward graphs must be released with their rollout"
            snap = {k: getattr(buf, k).clone() for k in ("states", "actions", "rewards", "values", "next_values", "flags")}
            # an update-like phase: synchronize, new >= 1 MB blocks written by kernels
            torch.cuda.synchronize()
            junk = [torch.empty(1 << 20, device="cuda").normal_() for _ in range(3)]
            (junk[0] @ junk[1][:1 << 20]).item()
            del junk
            torch.cuda.empty_cache()
        out.append(snap)
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k
    # the default policy shape keeps its graphs (they replay only K10 + K11)
    pi2, vf2 = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    env = _make_env(N, 5, seed=6)
    env.device_rollout(pi2, vf2, T, max_len, graph=True, persistent=False)
    assert len(env._dev_rollout._graphs) == 1


def test_persistent_rollout_keeps_the_reward_terms_and_rejects_other_forwards():
    """buf_rew6 through K13, and the error when the forward is not the fused one."""
    from olympic_hip._ffi import OlyError
    from olympic_hip.ppo import MLPCritic, MLPGaussianActor
    from olympic_hip.vecstep import A3DeviceRollout, TorchForward
    torch.manual_seed(2)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    res = []
    for persistent in (False, True):
        env = _make_env(96, 6, seed=8)
        env._dev_rollout = A3DeviceRollout(env, env.physics.blocks, rs=np.random.RandomState(3), keep_rew6=True)
        torch.manual_seed(4)
        env.device_rollout(pi, vf, 12, 5, anneal=1.0, graph=False, persistent=persistent)
        res.append(env._dev_rollout.rew6.clone())
    assert torch.equal(res[0], res[1]) and float(res[1].abs().sum()) > 0
    with pytest.raises(OlyError, match="fused MLP forward"):
        env.device_rollout(pi, vf, 12, 5, graph=False, persistent=True, forward=TorchForward(pi, vf))


def test_ppo_train_on_the_device_rollout(tmp_path):
    """PPO.train end to end with the device rollout (graphs for sampling and updates): finite losses,
    the optimiser steps, float64 rewards reach the return scan, two iterations re-use the graphs."""
    from olympic_hip.ppo import PPO, MLPCritic, MLPGaussianActor
    N = 256
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=1024, epochs=2,
                max_traj_len=16, use_gae=False, num_procs=N, max_grad_norm=0.05, mirror_coeff=0.0, eval_freq=100)
    ppo = PPO(args, str(tmp_path))
    ppo.use_graph_rollout = True
    torch.manual_seed(0)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    w0 = pi.means.weight.detach().clone()
    hist = ppo.train(lambda: _make_env(N, 9, seed=4), pi, vf, n_itr=3, verbose=False)
    assert len(hist) == 3 and all(np.isfinite(h["losses"]).all() for h in hist)
    assert not torch.equal(w0, pi.means.weight) and ppo.total_steps == 3 * 16 * N


# --------------------------------------------------------------------------------- K11
@pytest.mark.parametrize("N", [1, 31, 32, 33, 4096, 12288 + 5])
def test_fused_mlp_forward_vs_oracle_and_torch(eng, oracle, N):
    """oly_mlp_forward2 (actor 41 -> 256 -> 256 -> 12 and critic -> 1 in one launch): bit-exact against
    the oracle's k-ordered fma chains (the matrix cores' f32 arithmetic, restated), and within fp32
    summation-order noise (1e-5 relative to the row scale) of torch's own Linear / relu forward."""
    from olympic_hip.mlp import FusedMLPForward
    from olympic_hip.ppo import MLPCritic, MLPGaussianActor
    torch.manual_seed(N)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    with torch.no_grad():
        for p_ in list(pi.parameters()) + list(vf.parameters()):
            p_.mul_(3.0)                                   # default init is tiny: make the sums non-trivial
        for m_ in (pi.means.bias, vf.network_out.bias, pi.actor_layers[0].bias):
            m_.normal_(0, 0.5)
    pi.obs_mean = torch.randn(41, device="cuda") * 0.3
    pi.obs_std = torch.rand(41, device="cuda") + 0.5
    x = torch.randn(N, 41, device="cuda") * 2
    fw = FusedMLPForward(eng, pi, vf)
    mu, val = fw(x)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref_mu, ref_v = pi(x), vf(x).reshape(N)
    scale = max(1.0, float(ref_mu.abs().max()))
    assert float((mu - ref_mu).abs().max()) <= 2e-5 * scale
    assert float((val - ref_v).abs().max()) <= 2e-5 * max(1.0, float(ref_v.abs().max()))
    assert torch.equal(fw.value(x), val)                   # the one-network launch = the same chains
    n = min(N, 300)                                        # the CPU chain is slow: a slice
    sl = slice(N - n, N)
    h = lambda t_: t_.detach().cpu().numpy()
    par = lambda parts: [h(a) for lin in parts for a in (lin.weight, lin.bias)]
    e_mu = oracle.mlp_forward(h(x[sl]), *par(fw.pa), in_mean=h(pi.obs_mean), in_std=h(pi.obs_std))
    e_v = oracle.mlp_forward(h(x[sl]), *par(fw.pc))
    assert np.array_equal(h(mu[sl]), e_mu)                 # bit-exact: same products, same order
    assert np.array_equal(h(val[sl]), e_v[:, 0])


def test_fused_mlp_identity_with_asymmetric_weights(eng):
    """Layout check with exact integer data (a symmetric weight matrix would hide a row / column swap):
    hidden layers pass the input through (identity blocks), the head is an asymmetric integer matrix."""
    from olympic_hip.mlp import FusedMLPForward
    from olympic_hip.ppo import MLPCritic, MLPGaussianActor
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    with torch.no_grad():
        for net, layers, head in ((pi, pi.actor_layers, pi.means), (vf, vf.critic_layers, vf.network_out)):
            for lin in list(layers) + [head]:
                lin.weight.zero_()
                lin.bias.zero_()
            layers[0].weight[:41, :41] = torch.eye(41)
            layers[1].weight.copy_(torch.eye(256))
            head.weight.copy_(torch.arange(head.out_features * 256, dtype=torch.float32).reshape(head.out_features, 256) % 7 - 2)
    x = torch.randint(0, 5, (77, 41), device="cuda").float()      # non-negative: relu is the identity
    mu, val = FusedMLPForward(eng, pi, vf)(x)
    xp = torch.zeros(77, 256, device="cuda")
    xp[:, :41] = x
    assert torch.equal(mu, xp @ pi.means.weight.t()) and torch.equal(val, (xp @ vf.network_out.weight.t()).reshape(-1))


def test_device_rollout_full_size_properties(oracle):
    """Config 3 at full size (4096 environments x 400 steps, time limit 100): size-independent properties
    of the buffer the device rollout fills, and its hand-over to K6 (the return scan of the buffer equals
    the oracle's scan of the same arrays, bit for bit, with the float64 rewards)."""
    from olympic_hip.ppo import MLPCritic, MLPGaussianActor
    from olympic_hip.rollout import PPORollout
    N, T, max_len = 4096, 400, 100
    torch.manual_seed(1)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    env = _make_env(N, 32, seed=7, p_bad=1.0 / 300)
    buf = env.device_rollout(pi, vf, T, max_len, graph=True)
    info = env._dev_rollout.last_info
    last = (buf.flags & _abi.FLAG_LAST).bool()
    absorbing = (buf.flags & _abi.FLAG_ABSORBING).bool()
    assert bool(last[-1].all()) and not bool((absorbing & ~last).any())
    # no episode segment is longer than the time limit: every window of max_len steps holds a cut
    run = torch.zeros(N, dtype=torch.int32, device="cuda")
    longest = torch.zeros(N, dtype=torch.int32, device="cuda")
    for t in range(T):
        run = torch.where(last[t], torch.zeros_like(run), run + 1)
        longest = torch.maximum(longest, run)
    assert int(longest.max()) <= max_len - 1
    assert info["resets"] == N + int(last[:-1].sum()) and info["side_rows"] == int((last & ~absorbing).sum())
    r = buf.rewards
    assert r.dtype == torch.float64 and bool(torch.isfinite(r).all()) and float(r.min()) >= -0.3 - 1e-9 and float(r.max()) <= 1.0 + 1e-9
    assert torch.equal(buf.next_values[:-1][~last[:-1]], buf.values[1:][~last[:-1]])
    assert bool(torch.isfinite(buf.states).all()) and bool((buf.states[:, :, :4].norm(dim=-1) - 1).abs().max() < 1e-5)   # unit quaternion
    ret, adv = PPORollout(env.eng, gamma=0.99, eps=1e-5).finish(buf, normalize=False)
    sl = slice(100, 164)
    e_ret, e_adv = oracle.return_scan_r64(0.99, buf.rewards[:, sl].cpu().numpy(), buf.values[:, sl].cpu().numpy(),
                                          buf.next_values[:, sl].cpu().numpy(), buf.flags[:, sl].cpu().numpy())
    assert np.array_equal(ret[:, sl].cpu().numpy(), e_ret) and np.array_equal(adv[:, sl].cpu().numpy(), e_adv)


def test_vec_step_and_mlp_argument_errors(eng, golden):
    """Error behaviour at the boundary: wrong shapes / dtypes are refused before anything is launched,
    an unconfigured context says what is missing, unsupported MLP shapes are OLY_ERANGE."""
    from olympic_hip.engine import Engine, OlyError
    spec = specs.A3Spec(mass=41.5)
    blocks, state, ro, _ = _host_rollout(spec, 40, 2, 3, 2, 2, seed=1)
    d_blocks, d_state, d_ro = _to_device(eng, blocks, state, ro)
    eng.a3_vec_prepare(d_blocks, d_state, d_ro)                       # the well-formed case is accepted
    bad = dict(d_ro, buf_rewards=d_ro["buf_rewards"].float())         # rewards travel as float64
    with pytest.raises(OlyError, match="buf_rewards"):
        eng.a3_vec_prepare(d_blocks, d_state, bad)
    bad = dict(d_ro, ctr=d_ro["ctr"][:-2].clone())                    # one counter pair per workgroup
    with pytest.raises(OlyError, match="ctr"):
        eng.a3_vec_prepare(d_blocks, d_state, bad)
    bad = dict(d_ro, eps=None)                                        # stochastic rollout needs its noise
    with pytest.raises(OlyError, match="eps"):
        eng.a3_vec_prepare(d_blocks, d_state, bad)
    bad = dict(d_blocks, geom2=d_blocks["geom2"][:, :, :-1].contiguous())
    with pytest.raises(OlyError, match="geom2"):
        eng.a3_vec_prepare(bad, d_state, d_ro)
    fresh = Engine(0)
    with pytest.raises(OlyError, match="before a3_configure"):
        fresh.a3_vec_prepare(d_blocks, d_state, d_ro)
    fresh.a3_configure(spec, golden("a3_task.npz")["clock_lut"])
    with pytest.raises(OlyError, match="contact_configure"):
        fresh.a3_vec_prepare(d_blocks, d_state, d_ro)
    z = lambda *s: torch.zeros(s, device="cuda")
    with pytest.raises(OlyError, match="unsupported MLP shape"):
        eng.mlp_pack(z(128, 41), z(128), z(128, 128), z(128), z(12, 128), z(12))       # hidden must be 256
    with pytest.raises(OlyError, match="unsupported MLP shape"):
        eng.mlp_pack(z(256, 80), z(256), z(256, 256), z(256), z(12, 256), z(12))       # in_dim > 64
    pk = eng.mlp_pack(z(256, 41), z(256), z(256, 256), z(256), z(12, 256), z(12))
    with pytest.raises(OlyError, match="packed_a"):
        eng.mlp_forward2(z(8, 41), pk[:-4].clone(), 12, z(8, 12))                        # not a whole packed stream
    # the persistent rollout (K13) at the C boundary: weights missing / misaligned, a network that does not read the
    # observation
    import ctypes as C
    from olympic_hip import _ffi
    L, h = _ffi.lib(), eng.ctx.handle
    launch = eng.a3_vec_prepare(d_blocks, d_state, d_ro)
    pc = eng.mlp_pack(z(256, 41), z(256), z(256, 256), z(256), z(1, 256), z(1))
    with pytest.raises(OlyError, match="packed_critic"):
        launch.persistent(pk, True, pc[:-4].clone(), False)                              # not a whole packed stream
    cb, cst, cr = launch.structs
    P = _ffi.ptr

    def call(in_dim=41, pa=pk, pcr=pc):
        return L.oly_a3_rollout_persistent(h, 40, C.byref(cb), C.byref(cst), C.byref(cr), in_dim, P(pa), 1, P(pcr), 0, None,
                                           None, eng._s())
    off = torch.zeros(pk.numel() + 4, dtype=torch.float32, device="cuda")
    assert call(pa=off[1:]) == _abi.OLY_EINVAL and b"aligned" in L.oly_last_error(h)    # not 16-byte aligned
    assert call(pcr=None) == _abi.OLY_EINVAL and b"NULL" in L.oly_last_error(h)
    assert call(in_dim=40) == _abi.OLY_EINVAL and b"observation" in L.oly_last_error(h)
    assert call() == _abi.OLY_OK
    torch.cuda.synchronize()


@pytest.mark.parametrize("seed", range(40))
def test_vec_step_random_configurations(eng, golden, oracle, seed):
    """Differential fuzz over the launch shape: environments not a multiple of the 16-env workgroup, one
    readback row (K = 1), one-step trajectories (every step cuts and resets), a one-record reset ring (records
    re-used), 1 to 40 contact slots, deterministic and stochastic policies.  Integer state, flags, cursors and
    actions bit-exact, float64 values to 1e-11, observations to one float32 ulp."""
    cfg = np.random.default_rng(1000 + seed)
    spec = specs.A3Spec(mass=41.5)
    lut = golden("a3_task.npz")["clock_lut"]
    N = int(cfg.choice([1, 2, 15, 16, 17, 63, 65, 130, 257, 1023]))
    K = int(cfg.choice([1, 2, 7]))
    T = int(cfg.integers(1, 8))
    max_len = int(cfg.choice([1, 2, 3, 50]))
    depth = int(cfg.choice([1, 2, 5]))
    C = int(cfg.choice([1, 3, 16, 17, 40]))
    det = bool(cfg.integers(0, 2))
    blocks, state, ro, rng = _host_rollout(spec, N, K, T, max_len, depth, seed=seed, det=det, p_bad=0.1, p_low=0.05)
    if C != 16:
        blocks = a3_synthetic_blocks(N, K, seed=seed, C=C, p_bad=0.1, p_low=0.05)
        blocks["ncon"] = rng.integers(0, C + 1, blocks["ncon"].shape).astype(np.int32)
    d_blocks, d_state, d_ro = _to_device(eng, blocks, state, ro)
    launch = eng.a3_vec_prepare(d_blocks, d_state, d_ro)
    h = lambda t_: t_.cpu().numpy()
    for step in range(-1, T):
        fl = _abi.VSTEP_RESET_ALL if step < 0 else 0
        if step >= 0:
            ro["mu"][...] = rng.normal(0, 0.3, (N, spec.nu)).astype(np.float32)
            ro["value"][...] = rng.normal(0, 1, N).astype(np.float32)
            d_ro["state"].copy_(dev(ro["state"]))
            launch(0, dev(ro["mu"]), dev(ro["value"]))
        else:
            launch(fl)
        oracle.a3_vec_step(spec, lut, CONTACT, blocks, state, ro, fl)
        what = (seed, step, dict(N=N, K=K, T=T, max_len=max_len, depth=depth, C=C, det=det))
        for k in ("phase", "t1", "t2", "reached_frames", "target_reached", "mode", "seq_len"):
            assert np.array_equal(h(d_state[k]), state[k]), (k, what)
        for k in ("traj_len", "side_count", "side_t", "pool_count", "buf_flags"):
            assert np.array_equal(h(d_ro[k]), ro[k]), (k, what)
        assert np.array_equal(h(d_ro["buf_actions"]), ro["buf_actions"]), what
        assert np.array_equal(h(d_ro["pd_target"]), ro["pd_target"]), what
        np.testing.assert_allclose(h(d_state["sequence"]), state["sequence"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(h(d_state["goal"]), state["goal"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(h(d_ro["buf_rewards"]), ro["buf_rewards"], rtol=1e-11, atol=1e-13)
        for k in ("state", "buf_states", "side_obs"):
            a, b = h(d_ro[k]), ro[k]
            assert np.abs(a - b).max() <= np.spacing(np.float32(1.0)) * max(1.0, np.abs(b).max()), (k, what)
    c = h(d_ro["ctr"]).reshape(-1, 2)
    assert (c[:-1, 0] == ro["ctr"][0]).all() and (c[:-1, 1] == ro["ctr"][1]).all() and c[-1, 0] == 0


@pytest.mark.parametrize("seed", range(int(os.environ.get("OLY_FUZZ_K13", "24"))))
def test_persistent_rollout_random_configurations(seed):
    """Differential fuzz of K13 against the K11 + K10 loop (itself fuzzed against the oracle above) over the launch
    shape: environment counts around the 16-environment tile, one readback row, one-step trajectories (every step
    cuts and resets), a one-record reset ring (records re-used), 1 to 40 contact slots, deterministic / stochastic,
    T = 1.  Everything bit-identical.  OLY_FUZZ_K13=n runs n seeds (400 were run once on the shipped kernel)."""
    from olympic_hip.a3 import ReplayA3Physics, VecA3Env
    from olympic_hip.engine import Engine
    from olympic_hip.ppo import MLPCritic, MLPGaussianActor
    from olympic_hip.vecstep import A3DeviceRollout
    cfg = np.random.default_rng(7000 + seed)
    N = int(cfg.choice([1, 2, 15, 16, 17, 31, 33, 130, 257, 1023]))
    K = int(cfg.choice([1, 2, 7]))
    T = int(cfg.integers(1, 9))
    max_len = int(cfg.choice([1, 2, 3, 50]))
    depth = int(cfg.choice([1, 2, 5]))
    C = int(cfg.choice([1, 3, 16, 17, 40]))
    det = bool(cfg.integers(0, 2))
    torch.manual_seed(seed)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    host = a3_synthetic_blocks(N, K, seed=seed, C=C, p_bad=0.1, p_low=0.05)
    host["ncon"] = cfg.integers(0, C + 1, host["ncon"].shape).astype(np.int32)
    out = []
    for persistent in (False, True):
        blocks = {k: dev(v) for k, v in host.items()}
        env = VecA3Env(specs.A3Spec(mass=41.5), N, Engine(0), ReplayA3Physics(blocks), *CONTACT, rs=np.random.RandomState(seed))
        env._dev_rollout = A3DeviceRollout(env, blocks, pool_depth=depth, rs=np.random.RandomState(seed + 1), keep_rew6=True)
        torch.manual_seed(100 + seed)
        env.device_rollout(pi, vf, T, max_len, deterministic=det, anneal=0.6, graph=False, persistent=persistent)
        r = env._dev_rollout
        named = dict(states=r.buf.states, actions=r.buf.actions, means=r.buf.mu, rewards=r.buf.rewards, values=r.buf.values,
                     next_values=r.buf.next_values, flags=r.buf.flags, rew6=r.rew6, state_obs=r.state_obs,
                     pd_target=r.pd_target, traj_len=r.traj_len, side_obs=r.side_obs, side_t=r.side_t,
                     side_count=r.side_count, ctr=r.ctr)
        named.update({"st_" + k: v for k, v in env.state.items()})
        out.append(({k: v.clone() for k, v in named.items()}, dict(r.last_info)))
    what = dict(seed=seed, N=N, K=K, T=T, max_len=max_len, depth=depth, C=C, det=det)
    for name in out[0][0]:
        assert torch.equal(out[0][0][name], out[1][0][name]), (name, what)
    assert out[0][1] == out[1][1], what


@pytest.mark.parametrize("rows", [16, 32])
def test_fused_mlp_both_tile_heights_on_every_case(rows):
    """oly_mlp_forward2 picks 16-row tiles (v_mfma_f32_16x16x4_f32, 512 four-wave workgroups for 4096 rows x 2 networks)
    for small batches and 32-row tiles (v_mfma_f32_32x32x2_f32) for large ones; OLY_K11_ROWS forces one kernel onto
    every fused-MLP test and the persistent-rollout comparison of this file (the knob is read once per process, hence
    the child): both are bit-exact against the same oracle."""
    import subprocess
    import sys
    env = dict(os.environ, OLY_K11_ROWS=str(rows))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                          "(fused_mlp and not both_tile_heights) or persistent_rollout_equals", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("in_dim,out_a,out_b,N", [(1, 1, 1, 5), (7, 32, 1, 33), (40, 5, 32, 64), (64, 12, 1, 97),
                                                  (63, 31, 2, 200), (33, 1, 17, 31)])
def test_fused_mlp_other_dimensions_vs_oracle(eng, oracle, in_dim, out_a, out_b, N):
    """Every supported width (input up to 64, heads up to 32 columns, zero-padded inside the packed stream),
    two different heads in one launch, with and without input normalisation: bit-exact against the oracle."""
    rng = np.random.default_rng(in_dim * 100 + out_a)
    f = lambda *s: rng.normal(0, 0.4, s).astype(np.float32)
    nets = []
    for out in (out_a, out_b):
        nets.append([f(256, in_dim), f(256), f(256, 256) * 0.2, f(256), f(out, 256) * 0.2, f(out)])
    mean, std = f(in_dim), (rng.uniform(0.5, 1.5, in_dim)).astype(np.float32)
    x = f(N, in_dim) * 3
    pk_a = eng.mlp_pack(*[dev(a) for a in nets[0]], dev(mean), dev(std))
    pk_b = eng.mlp_pack(*[dev(a) for a in nets[1]])
    ya = torch.full((N, out_a), 9.0, device="cuda")
    yb = torch.full((N, out_b), 9.0, device="cuda")
    eng.mlp_forward2(dev(x), pk_a, out_a, ya, pk_b, out_b, yb, normalize_a=True, normalize_b=False)
    assert np.array_equal(ya.cpu().numpy(), oracle.mlp_forward(x, *nets[0], in_mean=mean, in_std=std))
    assert np.array_equal(yb.cpu().numpy(), oracle.mlp_forward(x, *nets[1]))
