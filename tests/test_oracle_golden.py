"""Pins the CPU oracle (oracle/oly_oracle.c) to the golden vectors that were produced by
executing the reference's own functions (tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from olympic_hip import _abi, specs
from helpers import a3_analytic_cases, check_a3_analytic, h1_rows_from_full, a3_fixture_arrays, ulp_diff


# ------------------------------------------------------------------------------ K1/K5
def test_h1_obs_fallen_reward_action(golden, oracle):
    g = golden("h1_step.npz")
    spec = specs.unitree_h1("walk")
    qpos, qvel = h1_rows_from_full(spec, g["full_obs"])
    M = len(qpos)
    prev = g["obs"][:, spec.reward_idx]            # reward(state=obs_i): prev obs := obs_i
    o = oracle.il_step(spec, qpos[None], qvel[None], g["action"][None].astype(np.float32), prev,
                       obs_f64=True, ctrl_f64=True)
    assert np.array_equal(o["obs"][0], g["obs"])                    # bit-exact float64
    assert np.array_equal(o["absorbing"][0].astype(bool), g["absorbing"])
    assert np.array_equal(o["absorbing"][0].astype(bool), g["fallen"])
    assert np.array_equal(o["fall_code"][0], g["msg_code"])
    assert ulp_diff(o["reward_f64"][0], g["reward_walk"]).max() <= 2   # libm exp vs numpy exp
    assert np.array_equal(o["prev"], g["obs"][:, spec.reward_idx])
    # action: golden action is float64; the kernel contract takes float32 actions
    a32 = g["action"].astype(np.float32).astype(np.float64)
    un = a32 * spec.act_delta + spec.act_mean
    exp_ctrl = np.zeros((M, spec.nu))
    exp_ctrl[:, spec.act_to_ctrl] = np.clip(un, spec.ctrl_lo, spec.ctrl_hi)
    assert np.array_equal(o["ctrl"][0], exp_ctrl)
    # and the reference's own un-normalisation on the float64 action
    assert np.array_equal(g["action"] * spec.act_delta + spec.act_mean, g["unnorm_action"])
    # fp32 outputs are the correctly rounded fp64 ones
    o32 = oracle.il_step(spec, qpos[None], qvel[None], g["action"][None].astype(np.float32), prev)
    assert np.array_equal(o32["obs"][0], g["obs"].astype(np.float32))
    assert np.array_equal(o32["ctrl"][0], exp_ctrl.astype(np.float32))


def test_h1_run_reward_and_no_absorbing(golden, oracle):
    g = golden("h1_step.npz")
    spec = specs.unitree_h1("run", use_absorbing_states=False)
    qpos, qvel = h1_rows_from_full(spec, g["full_obs"])
    o = oracle.il_step(spec, qpos[None], qvel[None], None, g["obs"][:, spec.reward_idx], obs_f64=True)
    assert ulp_diff(o["reward_f64"][0], g["reward_run"]).max() <= 2
    assert not o["absorbing"].any()                         # base_humanoid_robot.py:260
    assert np.array_equal(o["fall_code"][0], g["msg_code"])  # _has_fallen itself is unchanged


def test_h1_reward_reads_previous_obs(golden, oracle):
    """reward[t] = f(obs[t-1]); reward[0] = f(carried prev); prev_out = obs[T-1] (utils/reward.py:72)."""
    g = golden("h1_step.npz")
    spec = specs.unitree_h1("walk")
    T, N = 12, 100
    qpos, qvel = h1_rows_from_full(spec, g["full_obs"][:T * N])
    qpos, qvel = qpos.reshape(T, N, -1), qvel.reshape(T, N, -1)
    prev0 = np.linspace(-1, 3, N)
    o = oracle.il_step(spec, qpos, qvel, None, prev0, obs_f64=True)
    rw = g["reward_walk"][:T * N].reshape(T, N)
    assert ulp_diff(o["reward_f64"][1:], rw[:-1]).max() <= 2
    assert ulp_diff(o["reward_f64"][0], np.exp(-np.square(prev0 - 1.25))).max() <= 2
    assert np.array_equal(o["prev"], g["obs"][:T * N].reshape(T, N, -1)[-1, :, spec.reward_idx])


def test_h1_x_pos_reward_negative_index_quirk(golden, oracle):
    """PosReward index is get_obs_idx('q_pelvis_tx') = -2: python wraps to obs[n_obs-2]
    (loco_env_base.py:810,1203)."""
    g = golden("h1_step.npz")
    spec = specs.unitree_h1("walk", reward_type="x_pos")
    assert spec.reward_idx == spec.n_obs - 2
    qpos, qvel = h1_rows_from_full(spec, g["full_obs"][:64])
    o = oracle.il_step(spec, qpos[None], qvel[None], None, g["obs"][:64, -2], obs_f64=True)
    assert np.array_equal(o["reward_f64"][0], g["obs"][:64, -2])
    spec0 = specs.unitree_h1("walk", reward_type=None)
    o0 = oracle.il_step(spec0, qpos[None], qvel[None], None, np.zeros(64))
    assert not o0["reward"].any()


# --------------------------------------------------------------------------------- K6
def _ppo_block(g):
    """Lay the ragged golden episodes out as a [T,1] rollout block with flags."""
    L = g["ep_len"]
    n = int(L.sum())
    flags = np.zeros(n, np.uint8)
    next_val = np.zeros(n, np.float32)
    end = np.cumsum(L) - 1
    for e, dn, lv in zip(end, g["done_tail"], g["last_val"]):
        flags[e] = _abi.FLAG_LAST | (_abi.FLAG_ABSORBING if dn else 0)
        next_val[e] = lv
    return flags, next_val


def test_ppo_finish_path(golden, oracle):
    g = golden("ppo_returns.npz")
    flags, next_val = _ppo_block(g)
    ret, adv = oracle.return_scan(_abi.SCAN_RETURN, float(g["gamma"]), 0.95, g["rewards"][:, None],
                                  g["values"][:, None], next_val[:, None], flags[:, None])
    assert np.array_equal(ret[:, 0], g["returns"])          # bit-exact float32
    assert np.array_equal(adv[:, 0], g["adv"])
    st = oracle.adv_stats(adv)
    assert st[0] == adv.size
    out = oracle.adv_normalize(adv, st, ddof=1, eps=float(g["eps"]))
    np.testing.assert_allclose(out[:, 0], g["adv_norm"], rtol=2e-6, atol=2e-7)


def test_ppo_finish_path_float64_rewards(golden, oracle):
    """Rewards float32 cannot represent (what env.step really returns): the f64-reward scan is
    bit-exact vs PPOBuffer.finish_path, the f32-reward scan of the narrowed rewards is not."""
    g = golden("ppo_returns_f64.npz")
    flags, next_val = _ppo_block(g)
    ret, adv = oracle.return_scan_r64(float(g["gamma"]), g["rewards"][:, None], g["values"][:, None],
                                      next_val[:, None], flags[:, None])
    assert np.array_equal(ret[:, 0], g["returns"])
    assert np.array_equal(adv[:, 0], g["adv"])
    ret32, _ = oracle.return_scan(_abi.SCAN_RETURN, float(g["gamma"]), 0.95, g["rewards"][:, None].astype(np.float32),
                                  g["values"][:, None], next_val[:, None], flags[:, None])
    d = ulp_diff(ret32[:, 0], g["returns"])
    assert d.max() <= 2.0 and (d > 0).any()                  # narrowing the reward first costs last-place bits


def test_adv_normalize_parts_tree(oracle):
    """Rank triples are combined by a balanced pairwise tree in rank order; 1 part = plain normalise."""
    rng = np.random.default_rng(3)
    x = rng.normal(0.1, 1.3, 4096).astype(np.float32)
    shards = np.split(x, 8)
    parts = np.stack([oracle.adv_stats(s) for s in shards])
    tot = ((parts[0] + parts[1]) + (parts[2] + parts[3])) + ((parts[4] + parts[5]) + (parts[6] + parts[7]))
    assert np.array_equal(oracle.adv_normalize_parts(x, parts, 1, 1e-5), oracle.adv_normalize(x, tot, 1, 1e-5))
    assert np.array_equal(oracle.adv_normalize_parts(x, parts[:1], 0, 1e-8), oracle.adv_normalize(x, parts[0], 0, 1e-8))
    p3 = parts[:3]                                            # non power of two: padded with zeros
    assert np.array_equal(oracle.adv_normalize_parts(x, p3, 1, 1e-5),
                          oracle.adv_normalize(x, (p3[0] + p3[1]) + (p3[2] + 0.0), 1, 1e-5))


def test_ppo_reference_smoke_value():
    """r=[1,2,3], gamma=.99, V_last=10 -> [15.62329, 14.771, 12.9]."""
    from oracle import oracle as orc
    ret, _ = orc.return_scan(_abi.SCAN_RETURN, 0.99, 0.0, np.array([[1.], [2.], [3.]]),
                             np.zeros((3, 1)), np.array([[0.], [0.], [10.]]),
                             np.array([[0], [0], [_abi.FLAG_LAST]]))
    np.testing.assert_allclose(ret[:, 0], [15.62329, 14.771, 12.9], rtol=1e-6)


def _gae_py(v, vn, r, absorbing, last, gamma, lam):
    """mushroom_rl.utils.value_functions.compute_gae restated (float32 numpy, as called)."""
    adv = np.empty_like(v)
    for rev_k in range(len(v)):
        k = len(v) - rev_k - 1
        if last[k] or rev_k == 0:
            adv[k] = r[k] - v[k]
            if not absorbing[k]:
                adv[k] += gamma * vn[k]
        else:
            adv[k] = r[k] + gamma * vn[k] - v[k] + gamma * lam * adv[k + 1]
    return adv + v, adv


def test_gae_matches_restated_mushroom(oracle):
    rng = np.random.default_rng(0)
    T, N = 257, 7
    r = rng.uniform(-0.3, 1, (T, N)).astype(np.float32)
    v = rng.normal(0, 1, (T, N)).astype(np.float32)
    vn = rng.normal(0, 1, (T, N)).astype(np.float32)
    last = rng.uniform(size=(T, N)) < 0.02
    absorbing = last & (rng.uniform(size=(T, N)) < 0.5)
    flags = (last * _abi.FLAG_LAST + absorbing * _abi.FLAG_ABSORBING).astype(np.uint8)
    ret, adv = oracle.return_scan(_abi.SCAN_GAE, 0.99, 0.97, r, v, vn, flags)
    for n in range(N):
        e_ret, e_adv = _gae_py(v[:, n:n + 1], vn[:, n:n + 1], r[:, n], absorbing[:, n], last[:, n],
                               0.99, 0.97)
        assert np.array_equal(adv[:, n], e_adv[:, 0])
        assert np.array_equal(ret[:, n], e_ret[:, 0])


def test_gae_equals_its_definition(oracle):
    """Independent of the recursion: A_t = sum_l (gamma*lam)^l delta_{t+l} over the rest of the segment,
    delta_k = r_k + gamma*(1 - absorbing_k)*v_next_k - v_k (Schulman et al. 2016, eq. 16; the segment ends
    at the first `last` flag or with the buffer), evaluated in float64."""
    rng = np.random.default_rng(5)
    T, N, gamma, lam = 120, 6, 0.99, 0.97
    r = rng.uniform(-0.3, 1, (T, N)).astype(np.float32)
    v = rng.normal(0, 1, (T, N)).astype(np.float32)
    vn = rng.normal(0, 1, (T, N)).astype(np.float32)
    last = rng.uniform(size=(T, N)) < 0.05
    absorbing = last & (rng.uniform(size=(T, N)) < 0.5)
    flags = (last * _abi.FLAG_LAST + absorbing * _abi.FLAG_ABSORBING).astype(np.uint8)
    ret, adv = oracle.return_scan(_abi.SCAN_GAE, gamma, lam, r, v, vn, flags)
    delta = r.astype(np.float64) + gamma * np.where(absorbing, 0.0, vn.astype(np.float64)) - v
    want = np.zeros((T, N))
    for n in range(N):
        for t in range(T):
            k, w = t, 1.0
            while True:
                want[t, n] += w * delta[k, n]
                if last[k, n] or k == T - 1:
                    break
                k, w = k + 1, w * gamma * lam
    np.testing.assert_allclose(adv, want, rtol=0, atol=2e-5)
    np.testing.assert_allclose(ret, want + v, rtol=0, atol=2e-5)


def test_gae_hand_computed_toy(oracle):
    # 3 steps, no episode end inside, gamma=.5, lam=.5, v=0 everywhere, v_next=0:
    # adv2 = r2 = 4 ; adv1 = r1 + .25*adv2 = 3 ; adv0 = r0 + .25*adv1 = 1.75
    r = np.array([[1.], [2.], [4.]], np.float32)
    z = np.zeros((3, 1), np.float32)
    ret, adv = oracle.return_scan(_abi.SCAN_GAE, 0.5, 0.5, r, z, z, np.zeros((3, 1), np.uint8))
    assert adv[:, 0].tolist() == [1.75, 3.0, 4.0]
    assert ret[:, 0].tolist() == [1.75, 3.0, 4.0]


def test_gae_lambda1_equals_discounted_return(oracle):
    """GAE(lam=1) + v is the bootstrapped discounted return of finish_path (ppo.py:68-84)
    when v_next[t] = v[t+1] inside an episode."""
    rng = np.random.default_rng(1)
    T, N = 64, 5
    r = rng.uniform(0, 1, (T, N)).astype(np.float32)
    v = rng.normal(0, 1, (T, N)).astype(np.float32)
    vn = np.roll(v, -1, axis=0)
    vn[-1] = rng.normal(0, 1, N)
    flags = np.zeros((T, N), np.uint8)
    flags[20, 1] = _abi.FLAG_LAST | _abi.FLAG_ABSORBING
    flags[40, 3] = _abi.FLAG_LAST
    ret_g, _ = oracle.return_scan(_abi.SCAN_GAE, 0.99, 1.0, r, v, vn, flags)
    ret_r, _ = oracle.return_scan(_abi.SCAN_RETURN, 0.99, 1.0, r, v, vn, flags)
    np.testing.assert_allclose(ret_g, ret_r, rtol=2e-5, atol=2e-5)


# --------------------------------------------------------------------------------- K7
def test_col_stats_vs_standardizer_and_rms(golden, oracle):
    g = golden("running_stats.npz")
    x = g["x"]
    off = np.concatenate([[0], np.cumsum(g["lens"])])
    cs = None
    # Standardizer: _sum=0, _sumsq=1e-2, _count=1e-2 (networks.py:54-56)
    for i in range(len(g["lens"])):
        xb = x[off[i]:off[i + 1]].astype(np.float32)
        cs = oracle.col_stats(xb, cs)
        cnt = cs[0] + 1e-2
        mean = cs[1] / cnt
        std = np.sqrt(np.maximum((cs[2] + 1e-2) / cnt - mean ** 2, 1e-2))
        np.testing.assert_allclose(mean, g["st_mean"][i], rtol=1e-5, atol=1e-6)   # ref sums in f32
        np.testing.assert_allclose(std, g["st_std"][i], rtol=1e-5, atol=1e-6)
    out = oracle.disc_standardize(g["st_fwd_in"], None, g["st_fwd_mean"], g["st_fwd_std"])
    assert np.array_equal(out, g["st_fwd_out"].astype(np.float32))
    # RunningMeanStd (normalize.py:182-208) from the same (count,sum,sumsq) triples
    cs = None
    for i in range(len(g["lens"])):
        cs = oracle.col_stats(x[off[i]:off[i + 1]].astype(np.float32), cs)
    xs32 = x.astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(cs[1] / cs[0], xs32.mean(0), rtol=1e-12)
    np.testing.assert_allclose(cs[2] / cs[0] - (cs[1] / cs[0]) ** 2, xs32.var(0), rtol=1e-10)
    np.testing.assert_allclose(g["rms_mean"][-1], x.mean(0), rtol=1e-4)   # count starts at eps=1e-4


# --------------------------------------------------------------------------------- K4
def test_trajectory_cursor(golden, oracle):
    g = golden("trajectory.npz")
    table = g["table"]
    resets = g["resets"]
    ct, cs, origin, sample = oracle.traj_reset(table, resets[:, 1], resets[:, 0])
    assert np.array_equal(sample, g["reset_samples"])
    assert np.array_equal(ct, resets[:, 1]) and np.array_equal(cs, resets[:, 0])
    # random reset then walk to the end
    sub, tno = g["rnd_reset"]
    ct, cs, origin, sample = oracle.traj_reset(table, [tno], [sub])
    walk = [sample[0].copy()]
    L = table.shape[2]
    while True:
        cs, sample, at_end = oracle.traj_next(table, ct, cs, origin, sample)
        if at_end[0]:
            assert cs[0] == L                                  # reference: step == trajectory_length
            break
        walk.append(sample[0].copy())
    assert np.array_equal(np.array(walk), g["walk"])


def test_trajectory_euler(golden, oracle):
    g = golden("trajectory.npz")
    s = g["walk"][:5].copy()
    cur = np.random.default_rng(0).normal(size=(5, 17))
    out = oracle.traj_euler(17, 0.01, cur, s)
    exp = s.copy()
    for n in range(5):
        exp[n, :17] = [qp + 0.01 * qv for qp, qv in zip(cur[n], s[n, 17:34])]   # loco_env_base.py:517
    assert np.array_equal(out, exp)


# --------------------------------------------------------------------------------- K3
def test_contact_reduce(golden, oracle):
    g = golden("contacts.npz")
    o = oracle.contact_reduce(g["geom_bodyid"], int(g["floor_body"]), int(g["rfoot_body"]),
                              int(g["lfoot_body"]), g["ncon"], g["geom1"], g["geom2"], g["force6"],
                              g["pos"][:, :, 2])
    for k in ("n_r", "n_l", "idx_r", "idx_l"):
        assert np.array_equal(o[k], g[k]), k
    assert np.array_equal(o["bad"].astype(bool), g["bad"])
    np.testing.assert_allclose(o["grf_r"], g["grf_r"], rtol=1e-14)     # norm: BLAS vs plain loop
    np.testing.assert_allclose(o["grf_l"], g["grf_l"], rtol=1e-14)
    assert np.array_equal(o["min_z"], g["min_z"])
    assert np.array_equal((o["n_r"] + o["n_l"]) > 0, g["any_foot"])


# --------------------------------------------------------------------------------- K2
def test_a3_task_step_reward_done_obs(golden, oracle):
    g = golden("a3_task.npz")
    spec = specs.A3Spec(mass=float(g["mass"]))
    assert spec.period == int(g["period"]) and spec.delay_frames == int(g["delay_frames"])
    E, K = g["phase"].shape
    st, _ = a3_fixture_arrays(g, 0)
    for k in range(K):
        _, inp = a3_fixture_arrays(g, k)
        cr = oracle.contact_reduce(g["geom_bodyid"], int(g["floor_body"]), int(g["rfoot_body"]),
                                   int(g["lfoot_body"]), g["ncon"][:, k], g["geom1"][:, k],
                                   g["geom2"][:, k], g["force6"][:, k], g["cpos_z"][:, k])
        np.testing.assert_allclose(cr["grf_l"], g["grf_l"][:, k], rtol=1e-14)
        inp.update(grf_l=cr["grf_l"], grf_r=cr["grf_r"], min_z=cr["min_z"], n_r=cr["n_r"],
                   n_l=cr["n_l"], bad=cr["bad"])
        o = oracle.a3_step(spec, g["clock_lut"], inp, st)
        # integer task state: bit-exact
        assert np.array_equal(st["phase"], g["phase"][:, k]), k
        assert np.array_equal(st["t1"], g["t1"][:, k]) and np.array_equal(st["t2"], g["t2"][:, k])
        assert np.array_equal(st["target_reached"].astype(bool), g["target_reached"][:, k])
        assert np.array_equal(st["reached_frames"], g["reached_frames"][:, k])
        assert np.array_equal(o["done"].astype(bool), g["done"][:, k])
        # float: goal uses LAPACK inverse in the reference, analytic rigid inverse here
        np.testing.assert_allclose(st["goal"], g["goal"][:, k], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(o["rew6_f64"], g["rew6"][:, k], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(o["reward_f64"], g["reward"][:, k], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(o["obs"], g["obs"][:, k], rtol=1e-12, atol=1e-13)


def test_a3_pd_target_and_torque(golden, oracle):
    g = golden("a3_task.npz")
    spec = specs.A3Spec()
    assert np.array_equal(spec.motor_offset, g["motor_offset"])
    assert np.array_equal(spec.kp, g["pd_kp"]) and np.array_equal(spec.kd, g["pd_kd"])
    a32 = g["pd_action"].astype(np.float32)
    tgt = oracle.a3_pd_target(spec, a32)
    assert np.array_equal(tgt, a32.astype(np.float64) + g["motor_offset"])
    np.testing.assert_allclose(tgt, g["pd_target"], rtol=0, atol=1e-7)     # f32 action contract
    B, S, _ = g["pd_q"].shape
    for s in range(S):
        tau = oracle.a3_pd_torque(spec, g["pd_target"], g["pd_q"][:, s], g["pd_qd"][:, s])
        assert np.array_equal(tau, g["pd_tau"][:, s])                      # bit-exact float64


def test_clock_lut_range(golden):
    g = golden("a3_task.npz")
    lut = g["clock_lut"]
    assert lut.shape == (4, 88) and np.abs(lut).max() <= 1.0 + 1e-12


# --------------------------------------------------------------------------------- K8
def test_disc_reward(golden, oracle):
    g = golden("vail_disc.npz")
    d = g["d"].reshape(-1)
    r = oracle.disc_reward(d)
    # 1 - sigmoid(d) cancels: an error of k ulp in p becomes k*2^-24/(1-p+1e-8) in the reward.
    p = 1.0 / (1.0 + np.exp(-d.astype(np.float64)))
    tol = 4 * 2.0 ** -24 / (1 - p + 1e-8) + 4e-7 * np.abs(g["reward"]) + 1e-7
    assert (np.abs(r - g["reward"]) <= tol).all()
    re = oracle.disc_reward(g["d_ext"])
    pe = 1.0 / (1.0 + np.exp(-g["d_ext"].reshape(-1).astype(np.float64)))
    tole = 4 * 2.0 ** -24 / (1 - pe + 1e-8) + 4e-7 * np.abs(g["reward_ext"]) + 1e-7
    assert (np.abs(re - g["reward_ext"]) <= tole).all()
    assert np.isfinite(re).all() and re[-1] == pytest.approx(-np.log(np.float32(1e-8)), rel=1e-6)


def test_disc_standardize_and_reparam(golden, oracle):
    g = golden("vail_disc.npz")
    # the forward updated the statistics with x BEFORE standardising (networks.py:68-74)
    xs = oracle.disc_standardize(g["x"], np.arange(32), g["st_mean"], g["st_std"])
    ref = ((g["x"].astype(np.float64) - g["st_mean"]) / g["st_std"]).astype(np.float32)
    assert np.array_equal(xs, ref)
    z = oracle.disc_reparam(g["mu"], g["logvar"], g["eps"])
    import torch
    zt = (torch.tensor(g["mu"]) + torch.exp(torch.tensor(g["logvar"]) / 2) * torch.tensor(g["eps"])).numpy()
    np.testing.assert_allclose(z, zt, rtol=3e-7, atol=1e-7)


def test_vail_forward_end_to_end(golden, oracle):
    """Pre-amble + torch GEMMs + epilogue reproduce the reference network's logits/reward."""
    import torch
    g = golden("vail_disc.npz")
    cs = oracle.col_stats(g["x"])
    cnt = cs[0] + 1e-2
    mean = cs[1] / cnt
    std = np.sqrt(np.maximum((cs[2] + 1e-2) / cnt - mean ** 2, 1e-2))
    np.testing.assert_allclose(mean, g["st_mean"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(std, g["st_std"], rtol=2e-5, atol=2e-6)
    xs = torch.tensor(oracle.disc_standardize(g["x"], None, g["st_mean"], g["st_std"]))
    t = lambda k: torch.tensor(g[k])
    h = torch.relu(xs @ t("enc_w0").T + t("enc_b0"))
    h = torch.relu(h @ t("enc_w1").T + t("enc_b1"))
    mu, lv = h @ t("mu_w").T + t("mu_b"), h @ t("lv_w").T + t("lv_b")
    np.testing.assert_allclose(mu.numpy(), g["mu"], rtol=1e-4, atol=1e-5)
    z = torch.tensor(oracle.disc_reparam(mu.numpy(), lv.numpy(), g["eps"]))
    d = z @ t("dec_w").T + t("dec_b")
    np.testing.assert_allclose(d.numpy(), g["d"], rtol=1e-3, atol=1e-3)


# --------------------------------------------------------------------------------- K12
def test_exp32_against_fp64_exp(oracle):
    """The fixed float32 exp of the fused discriminator: <= 1 ulp of exp over the whole range, exact
    limits, monotone ends (the error class of the torch.exp that reparameterize calls)."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-104.0, 88.7, 60000), rng.normal(0, 2, 60000), rng.normal(0, 0.05, 20000),
                        [0.0, -0.0, 88.72283, 88.72284, -103.9720, -103.9721, -87.5, 1e-30, -1e-30]]).astype(np.float32)
    y = oracle.exp32(x)
    ref = np.exp(x.astype(np.float64))
    spacing = np.spacing(np.clip(ref, 2.0 ** -149, 3.4e38).astype(np.float32)).astype(np.float64)
    fin = np.isfinite(y)
    assert (np.abs(y[fin].astype(np.float64) - ref[fin]) <= 1.0 * spacing[fin]).all()
    assert (ref[~fin] > 3.4028234e38).all()                            # infinity only where exp overflows float32
    sp = oracle.exp32(np.array([np.nan, np.inf, -np.inf, 100.0, -120.0], np.float32))
    assert np.isnan(sp[0]) and sp[1] == np.inf and sp[2] == 0.0 and sp[3] == np.inf and sp[4] == 0.0
    assert oracle.exp32(np.zeros(1, np.float32))[0] == 1.0


def test_disc_forward_oracle_against_the_reference_network(golden, oracle):
    """oly_disc_forward_cpu (the fused kernel's arithmetic) on the inputs of the reference run:
    VariationalNet + Standardizer forward + make_discrim_reward, executed by gen_golden.py."""
    g = golden("vail_disc.npz")
    o = oracle.disc_forward(g["x"], g, mask=np.arange(32), mean=g["st_mean"], std=g["st_std"], eps=g["eps"])
    # Linear layers: fma chains in k order vs the reference's BLAS summation order
    np.testing.assert_allclose(o["mu"], g["mu"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(o["logvar"], g["logvar"], rtol=1e-4, atol=2e-6)
    d = g["d"].reshape(-1)
    np.testing.assert_allclose(o["logits"], d, rtol=2e-5, atol=2e-6)
    # reward: the difference in d (<= 2e-6 + 2e-5 |d|) passes through dr/dd = sigmoid(d) <= 1, plus the
    # float32 steps of the formula, amplified by the 1 - p cancellation (see test_disc_reward)
    p = 1.0 / (1.0 + np.exp(-d.astype(np.float64)))
    tol = (2e-6 + 2e-5 * np.abs(d)) + 4 * 2.0 ** -24 / (1 - p + 1e-8) + 4e-7 * np.abs(g["reward"]) + 1e-7
    assert (np.abs(o["reward"] - g["reward"]) <= tol).all()
    # the same statistics derived from the running sums (Standardizer.update_mean_std on x)
    cs = oracle.col_stats(g["x"])
    o2 = oracle.disc_forward(g["x"], g, colstats=cs, eps=g["eps"])
    np.testing.assert_allclose(o2["logits"], d, rtol=1e-4, atol=1e-5)
    # composition of the separately pinned pieces: standardise -> the same chains on pre-standardised input
    xs = oracle.disc_standardize(g["x"], None, g["st_mean"], g["st_std"])
    o3 = oracle.disc_forward(xs, g, eps=g["eps"])
    for k in o:
        assert np.array_equal(o[k], o3[k]), k
    # deterministic forward (no noise): z = mu
    o4 = oracle.disc_forward(xs, g)
    assert np.array_equal(o4["mu"], o["mu"]) and not np.array_equal(o4["logits"], o["logits"])
    lo = np.zeros(len(xs), np.float32)
    hi = np.zeros(len(xs), np.float32)
    w = g["dec_w"].reshape(-1)
    for k in range(64):
        lo = np.float32(np.float64(o4["mu"][:, k]) * np.float64(w[k]) + np.float64(lo))       # fma: one rounding
        hi = np.float32(np.float64(o4["mu"][:, 64 + k]) * np.float64(w[64 + k]) + np.float64(hi))
    assert np.array_equal(o4["logits"], (lo + hi) + g["dec_b"][0])


# ------------------------------------------------------------------------------ K9 / normalisers
def test_ppo_loss_terms_match_reference_update_policy(golden, oracle):
    """oly_ppo_loss_cpu / oly_mirror_loss_cpu on the reference's own update_policy outputs
    (tolerance: the reference reduces in fp32, the oracle in fp64)."""
    from helpers import ppo_update_arrays
    g = golden("ppo_update.npz")
    a = ppo_update_arrays(g)
    scal, gmu, gsd, gv = oracle.ppo_loss(a["mu"], a["std"], a["old_mu"], a["std"], a["action"], a["adv"], a["ret"],
                                         a["value"], a["clip"], 0.5)
    for i, n in enumerate(("actor_loss", "entropy_penalty", "critic_loss", "approx_kl_div", "clip_fraction")):
        np.testing.assert_allclose(scal[i], float(g[n]), rtol=3e-5, atol=3e-7, err_msg=n)
    loss, gd, gm = oracle.mirror_loss(a["mu"], a["mir"], a["act_src"], a["act_sign"])
    np.testing.assert_allclose(loss, float(g["mirror_loss"]), rtol=3e-5, atol=1e-9)
    # gradients: finite differences of the oracle's own forward in float64-safe directions
    B, A = a["mu"].shape
    assert np.isfinite(gmu).all() and gmu.shape == (B, A) and gv.shape == (B,)
    np.testing.assert_allclose(gv, 0.5 * 2 * (a["value"] - a["ret"]) / B, rtol=1e-6)
    np.testing.assert_allclose(gd, 2 * (a["mu"] - a["act_sign"] * a["mir"][:, a["act_src"]]) / (B * A), rtol=1e-6)
    back = np.zeros_like(gm)
    back[:, a["act_src"]] = -a["act_sign"] * gd
    assert np.array_equal(back, gm)


def test_signed_perm_is_the_reference_matrix_product(golden, oracle):
    from olympic_hip.wrappers import _signed_perm
    g = golden("symmetry.npz")
    for tab, x, want in ((g["mirrored_obs"], g["obs"], g["obs_mirror"]), (g["mirrored_acts"], g["act"], g["act_mirror"])):
        src, sgn = _signed_perm(tab.tolist())
        out = oracle.signed_perm(x, src, sgn)
        assert np.array_equal(out, want.astype(np.float32))


def test_obs_filter_and_running_mean_std(golden, oracle):
    g = golden("normalize.npz")
    clip, eps = float(g["clipob"]), float(g["epsilon"])
    out = oracle.obs_filter(g["frozen_in"], g["mean"][-1], g["var"][-1], eps, clip)
    assert np.array_equal(out, g["frozen_out"].astype(np.float32))           # frozen statistics: exact
    # online: batch moments from (count, sum, sumsq) + the reference's merge (normalize.py:188-208)
    off = np.concatenate([[0], np.cumsum(g["lens"])])
    mean, var, count = np.zeros(6), np.zeros(6), 1e-4
    for i in range(len(g["lens"])):
        xb = g["x"][off[i]:off[i + 1]]
        cs = oracle.col_stats(xb)
        n = cs[0]
        bm = cs[1] / n
        bv = cs[2] / n - bm * bm
        delta, tot = bm - mean, count + n
        m2 = var * count + bv * n + delta ** 2 * count * n / tot
        mean, var, count = mean + delta * n / tot, m2 / tot, tot
        np.testing.assert_allclose(mean, g["mean"][i], rtol=2e-6, atol=2e-7)   # reference moments are fp32
        np.testing.assert_allclose(var, g["var"][i], rtol=2e-5)
        assert count == pytest.approx(float(g["count"][i]))
        o = oracle.obs_filter(xb, mean, var, eps, clip)
        np.testing.assert_allclose(o, g["out"][off[i]:off[i + 1]], rtol=2e-5, atol=2e-6)


def test_il_ground_forces_hand_cases(oracle):
    """No reference oracle exists for _get_collision_force / RunningAveragedWindow (mushroom-rl,
    absent: parity unpinned); pinned to hand-derived cases of the restated definition."""
    gg = np.array([0, -1, 1, 1, 2], np.int32)                     # floor, -, foot_r x2, foot_l
    pairs = [(0, 1), (0, 2)]
    f = np.zeros((2, 1, 4, 6))
    f[0, 0] = [[1, 2, 3, 9, 9, 9], [4, 5, 6, 9, 9, 9], [7, 8, 9, 9, 9, 9], [10, 11, 12, 9, 9, 9]]
    f[1, 0] = 10 * f[0, 0]
    g1 = np.array([[[1, 0, 3, 0]], [[4, 2, 0, 0]]], np.int32)
    g2 = np.array([[[2, 2, 0, 4]], [[0, 0, 3, 4]]], np.int32)
    ncon = np.array([[4], [3]], np.int32)
    step, mean = oracle.il_ground_forces(gg, pairs, ncon, g1, g2, f)
    # substep 0: contact 0 has an ungrouped geom; contact 1 = (floor, foot_r); contact 2 = (foot_r, floor)
    # comes later; contact 3 = (floor, foot_l)
    assert step[0, 0].tolist() == [4, 5, 6, 10, 11, 12]
    # substep 1: contact 0 = (foot_l, floor) reversed order counts; contact 1 = (foot_r, floor); the 4th is beyond ncon
    assert step[1, 0].tolist() == [40, 50, 60, 10, 20, 30]
    assert mean[0].tolist() == [22, 27.5, 33, 10, 15.5, 21]
    step0, mean0 = oracle.il_ground_forces(gg, pairs, np.zeros((2, 1), np.int32), g1, g2, f)
    assert not step0.any() and not mean0.any()
    # raw ncon beyond the 4 staged slots (the reference scans every data.ncon contact, UnitreeH1.py:113-123):
    #  env 0: both pairs have their first contact among the staged slots -> exact, no flag;
    #  env 1: the foot_l pair has none among them (its contact may be the unseen 5th) -> flagged;
    #  env 2: negative count -> flagged, reduced as no contact
    g1b = np.array([[[0, 0, 1, 1], [0, 0, 1, 1], [0, 0, 0, 0]]], np.int32)
    g2b = np.array([[[2, 4, 1, 1], [2, 3, 1, 1], [2, 4, 2, 4]]], np.int32)
    fb = np.arange(1 * 3 * 4 * 6, dtype=np.float64).reshape(1, 3, 4, 6)
    stepb, meanb, over = oracle.il_ground_forces(gg, pairs, np.array([[6, 5, -2]], np.int32), g1b, g2b, fb, want_overflow=True)
    assert over.tolist() == [0, 1, 1]
    assert stepb[0, 0].tolist() == [0, 1, 2, 6, 7, 8] and stepb[0, 1].tolist() == [24, 25, 26, 0, 0, 0]
    assert not stepb[0, 2].any() and np.array_equal(meanb, stepb[0])
    _, _, over4 = oracle.il_ground_forces(gg, pairs, np.array([[4, 4, 0]], np.int32), g1b, g2b, fb, want_overflow=True)
    assert over4.tolist() == [0, 0, 0]                                # ncon == C is not an overflow


def test_robot_geom_tables():
    from olympic_hip.robot_data import ROBOTS
    h1 = ROBOTS["UnitreeH1"]
    assert dict(h1["collision_groups"]) == {"floor": [0], "foot_r": [22], "foot_l": [12]} and h1["n_geom"] == 42
    assert h1["grf_pairs"] == [("floor", "foot_r"), ("floor", "foot_l")]
    assert ROBOTS["Atlas"]["grf_pairs"] is None and len(ROBOTS["Talos"]["grf_pairs"]) == 2
    sp = specs.unitree_h1("walk").with_foot_forces("UnitreeH1")
    assert sp.grf_pairs == [(0, 1), (0, 2)] and sp.n_obs == 38


def test_rollout_cuts_hand_case(oracle):
    """PPO.sample's cut rules (rl/algos/ppo.py:169-196): done -> LAST|ABSORBING, time limit or block
    end -> LAST only, counters restart after a cut."""
    fl, tl, nc = oracle.rollout_cuts([0, 1, 0, 0], [3, 3, 8, 9], 10, False)
    assert fl.tolist() == [0, 3, 0, 2] and tl.tolist() == [4, 0, 9, 0] and nc == 2
    fl, tl, nc = oracle.rollout_cuts([0, 1, 0, 0], [3, 3, 8, 9], 10, True)
    assert fl.tolist() == [2, 3, 2, 2] and tl.tolist() == [0, 0, 0, 0] and nc == 4


def test_a3_orientation_analytic_cases(golden, oracle):
    """transforms3d boundary (parity unpinned): the oracle's quaternion / euler algebra on
    closed-form cases (identity, pure yaw, +-90 deg, gimbal branch, round trips)."""
    g = golden("a3_task.npz")
    c = a3_analytic_cases()
    st = {k: v.copy() for k, v in c["state"].items()}
    eo = oracle.a3_step(specs.A3Spec(mass=41.5), g["clock_lut"], c["inputs"], st)
    check_a3_analytic(eo["obs"], st["goal"], c)


def test_vec_step_replays_the_reference_sequence(golden, oracle):
    """oly_a3_vec_step_cpu (contacts -> task.step -> reward -> done -> get_obs -> cut rule, one call
    per step) on the reference-generated a3_task fixture: until an environment is done for the
    first time (the fixture keeps stepping, the rollout loop resets) observations, rewards, done
    flags and integer task state are the reference's."""
    from olympic_hip.vecstep import REC
    g = golden("a3_task.npz")
    spec = specs.A3Spec(mass=float(g["mass"]))
    E, K = g["phase"].shape
    names = ("qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel", "rf_vel", "root_pos", "root_quat",
             "head_pos", "ncon", "geom1", "geom2", "force6", "cpos_z")
    blocks = {n: np.ascontiguousarray(np.swapaxes(g[n], 0, 1)) for n in names}
    state, _ = a3_fixture_arrays(g, 0)
    T, nobs, nu, slots = K, spec.n_obs, spec.nu, 2
    pool = np.zeros(E, REC)
    pool["mode"], pool["seq_len"] = _abi.MODE_STANDING, 1
    ro = dict(T=T, max_traj_len=10 ** 6, deterministic=True, side_slots=slots, pool_depth=1,
              mu=np.zeros((E, nu), np.float32), value=np.zeros(E, np.float32), scale=None, eps=None,
              state=np.zeros((E, nobs), np.float32), pd_target=np.zeros((E, nu)),
              buf_states=np.zeros((T, E, nobs), np.float32), buf_actions=np.zeros((T, E, nu), np.float32),
              buf_rewards=np.zeros((T, E)), buf_values=np.zeros((T, E), np.float32),
              buf_flags=np.zeros((T, E), np.uint8), buf_rew6=np.zeros((T, E, 6), np.float32),
              traj_len=np.zeros(E, np.int32), side_obs=np.zeros((E * slots, nobs), np.float32),
              side_t=np.full(E * slots, -1, np.int32), side_count=np.zeros(E, np.int32),
              pool=pool.view(np.uint8).reshape(-1).copy(), pool_count=np.zeros(E, np.int32),
              ctr=np.zeros(2, np.int32))
    contact = (g["geom_bodyid"], int(g["floor_body"]), int(g["rfoot_body"]), int(g["lfoot_body"]))
    fixture_state, _ = a3_fixture_arrays(g, 0)
    for k in range(K - 1):
        # put every environment back on the fixture's trajectory (undoing the resets of step k-1)
        for name in ("mode", "seq_len", "sequence"):
            state[name][...] = fixture_state[name]
        if k > 0:
            for name in ("phase", "t1", "t2", "reached_frames"):
                state[name][...] = g[name][:, k - 1]
            state["target_reached"][...] = g["target_reached"][:, k - 1]
        oracle.a3_vec_step(spec, g["clock_lut"], contact, blocks, state, ro, 0)
        done = g["done"][:, k]
        assert np.array_equal((ro["buf_flags"][k] & _abi.FLAG_ABSORBING) != 0, done)
        assert np.array_equal((ro["buf_flags"][k] & _abi.FLAG_LAST) != 0, done)          # no time limit here
        np.testing.assert_allclose(ro["buf_rewards"][k], g["reward"][:, k], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(ro["buf_rew6"][k], g["rew6"][:, k], rtol=2e-6, atol=1e-7)
        a = ~done                                                    # not reset: the reference's next observation
        assert np.array_equal(ro["state"][a], g["obs"][a, k].astype(np.float32))
        for name in ("phase", "t1", "t2", "reached_frames"):
            assert np.array_equal(state[name][a], g[name][a, k]), name
        # a reset environment restarts un-advanced: robot part unchanged, goal steps zero, counters zero,
        # clock of the drawn phase (walking_task.py:321-344, StickFigureA3.py:205-235)
        assert np.array_equal(ro["state"][done][:, :nobs - 10], g["obs"][done, k].astype(np.float32)[:, :nobs - 10])
        assert not ro["state"][done][:, nobs - 8:].any() and not state["reached_frames"][done].any()
        assert np.array_equal(ro["state"][done][:, nobs - 10:nobs - 8], np.tile(np.float32([0, 1]), (done.sum(), 1)))
    assert g["done"][:, :K - 1].sum() > 100 and (~g["done"][:, :K - 1]).sum() > 100
    assert ro["ctr"].tolist() == [K - 1, K - 1] and ro["pool_count"].sum() == g["done"][:, :K - 1].sum()


def test_vec_step_reset_matches_reference_reset(golden, oracle):
    """env.reset() inside the vec step (OLY_VSTEP_RESET_ALL) on the reference-generated a3_reset fixture:
    WalkingTask.reset's transform_sequence (mid point of the feet, root yaw incl. rolled / pitched roots),
    t1 / t2 after update_target_steps, and get_obs of the un-advanced task (goal steps zero, clock of the
    drawn phase), from the local step sequence captured inside the reference's reset."""
    from helpers import a3_reset_fixture_rollout, check_a3_reset_fixture
    g = golden("a3_reset.npz")
    spec = specs.A3Spec(mass=41.5)
    blocks, state, ro = a3_reset_fixture_rollout(g, spec)
    lut = golden("a3_task.npz")["clock_lut"]
    oracle.a3_vec_step(spec, lut, (np.arange(13, dtype=np.int32), 0, 7, 10), blocks, state, ro, _abi.VSTEP_RESET_ALL)
    check_a3_reset_fixture(g, state, ro["state"])
    assert (ro["pool_count"] == 1).all() and ro["ctr"].tolist() == [0, 0]
    assert set(np.unique(g["mode"])) == {_abi.MODE_STANDING, _abi.MODE_FORWARD} and set(np.unique(g["phase"])) == {0, 44}


def test_rotation_restatement_against_scipy(golden, oracle):
    """transforms3d is absent, so the goldens ran the reference against tests/golden/_ref_stubs.py's
    restatement of it (parity unpinned).  scipy's Rotation is an independent implementation of the same
    conventions (extrinsic x-y-z = transforms3d 'sxyz', scalar-last quaternions): it has to agree with the
    stubs, with the oracle's obs quaternion (quat2euler -> euler2quat with yaw dropped,
    StickFigureA3.py:160-161) and with the host yaw used at reset (a3.yaw_of_quat)."""
    import os
    import sys
    from scipy.spatial.transform import Rotation
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import _ref_stubs as rs
    from olympic_hip.a3 import yaw_of_quat
    rng = np.random.default_rng(21)
    n = 400
    rpy = np.stack([rng.uniform(-np.pi, np.pi, n), rng.uniform(-1.5, 1.5, n), rng.uniform(-np.pi, np.pi, n)], 1)
    R = Rotation.from_euler("xyz", rpy)
    xyzw = R.as_quat()
    wxyz = np.concatenate([xyzw[:, 3:], xyzw[:, :3]], 1)
    for i in range(n):
        m = rs._euler2mat(*rpy[i])
        np.testing.assert_allclose(m, R[i].as_matrix(), atol=1e-14)
        np.testing.assert_allclose(rs._quat2mat(wxyz[i] * rng.uniform(0.5, 2.0)), R[i].as_matrix(), atol=1e-14)
        q = rs._euler2quat(*rpy[i])
        assert min(np.abs(q - wxyz[i]).max(), np.abs(q + wxyz[i]).max()) < 1e-14
        np.testing.assert_allclose(rs._mat2euler(m), rpy[i], atol=1e-12)
        np.testing.assert_allclose(rs._quat2euler(wxyz[i]), R[i].as_euler("xyz"), atol=1e-12)
        assert abs(yaw_of_quat(wxyz[i]) - rpy[i, 2]) < 1e-12
    # the oracle's obs[0:4]: the body quaternion with its yaw removed = scipy's (roll, pitch, 0)
    c = a3_analytic_cases()
    st = {k: v.copy() for k, v in c["state"].items()}
    obs = oracle.a3_step(specs.A3Spec(mass=41.5), golden("a3_task.npz")["clock_lut"], c["inputs"], st)["obs"]
    body = c["inputs"]["qpos"][:, 3:7]
    e = Rotation.from_quat(np.concatenate([body[:, 1:], body[:, :1]], 1)).as_euler("xyz")
    regular = np.abs(np.abs(e[:, 1]) - np.pi / 2) > 1e-6          # scipy warns / differs in the gimbal branch
    flat = Rotation.from_euler("xyz", np.stack([e[:, 0], e[:, 1], np.zeros(len(e))], 1)).as_quat()
    flat = np.concatenate([flat[:, 3:], flat[:, :3]], 1)
    got = obs[:, :4].astype(np.float64)
    err = np.minimum(np.abs(got - flat).max(1), np.abs(got + flat).max(1))
    assert regular.sum() > 490 and err[regular].max() < 1e-6     # obs is float32


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")
def test_fixtures_regenerate_bit_identically(tmp_path):
    """Every committed fixture is reproducible: tests/golden/gen_golden.py (which EXECUTES the reference's own
    functions from /root/reference under inert stubs) is re-run into a temporary directory and each array of each
    .npz must equal the committed one bit for bit (fixed seeds everywhere; no salted hash() seeds)."""
    import glob
    import subprocess
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    env = dict(os.environ, PYTHONHASHSEED="random")
    for k in ("LD_PRELOAD", "OLY_ORACLE_ASAN", "ASAN_OPTIONS", "UBSAN_OPTIONS"):   # the generator never touches the oracle
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(here, "gen_golden.py"), "--out", str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    committed = sorted(os.path.basename(f) for f in glob.glob(os.path.join(here, "*.npz")))
    assert sorted(os.path.basename(f) for f in glob.glob(str(tmp_path / "*.npz"))) == committed
    for name in committed:
        a, b = np.load(os.path.join(here, name), allow_pickle=False), np.load(str(tmp_path / name), allow_pickle=False)
        assert sorted(a.files) == sorted(b.files), name
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (name, k)
            assert a[k].tobytes() == b[k].tobytes(), f"{name}:{k} does not regenerate"
