"""Shared helpers for the parity tests (input reshaping, ulp distance, synthetic batches)."""
import numpy as np

from olympic_hip.synthetic import h1_rows_from_full, h1_synthetic_block  # noqa: F401


def ulp_diff(a, b):
    """Distance in units of the spacing of b (works for float32 or float64 arrays)."""
    a, b = np.asarray(a), np.asarray(b)
    sp = np.spacing(np.maximum(np.abs(b), np.finfo(b.dtype).tiny).astype(b.dtype))
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / sp.astype(np.float64)


def a3_fixture_arrays(g, k):
    """(state at reset, inputs of step k) of tests/golden/a3_task.npz as oracle/kernel dicts."""
    E = g["phase"].shape[0]
    st = dict(phase=g["phase0"].astype(np.int32).copy(), t1=g["t1_0"].astype(np.int32).copy(),
              t2=g["t2_0"].astype(np.int32).copy(), reached_frames=np.zeros(E, np.int32),
              target_reached=np.zeros(E, np.uint8), mode=g["mode"].astype(np.int32).copy(),
              seq_len=g["seq_len"].astype(np.int32).copy(),
              sequence=np.ascontiguousarray(g["sequence"], dtype=np.float64),
              goal=np.zeros((E, 8)))
    inp = {n: np.ascontiguousarray(g[n][:, k]) for n in
           ("qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel", "rf_vel", "root_pos",
            "root_quat", "head_pos")}
    return st, inp




def ppo_update_arrays(g):
    """numpy forward of the fixture's actor / old actor / critic (relu MLPs) on its obs, and the
    mirrored forward: everything oly_ppo_loss / oly_mirror_loss take as inputs."""
    from olympic_hip.wrappers import _signed_perm

    def mlp(x, tag, layers, head):
        for i in range(2):
            x = np.maximum(x @ g[f"{tag}.{layers}.{i}.weight"].T + g[f"{tag}.{layers}.{i}.bias"], 0)
        return (x @ g[f"{tag}.{head}.weight"].T + g[f"{tag}.{head}.bias"]).astype(np.float32)
    obs = g["obs"].astype(np.float32)
    o_src, o_sgn = _signed_perm(g["mirrored_obs"].tolist())
    a_src, a_sgn = _signed_perm(g["mirrored_acts"].tolist())
    mobs = obs[:, o_src] * o_sgn
    for i in (31, 32):                                     # mirror_clock_observation: sin(arcsin(x) + pi)
        mobs[:, i] = np.sin(np.arcsin(mobs[:, i]) + np.float32(np.pi))
    return dict(mu=mlp(obs, "pi", "actor_layers", "means"), old_mu=mlp(obs, "old", "actor_layers", "means"),
                value=mlp(obs, "vf", "critic_layers", "network_out").reshape(-1),
                mir=mlp(mobs.astype(np.float32), "pi", "actor_layers", "means"),
                act_src=a_src.astype(np.int32), act_sign=a_sgn.astype(np.float32),
                obs_src=o_src.astype(np.int32), obs_sign=o_sgn.astype(np.float32),
                std=np.float32(g["fixed_std"]), action=g["act"].astype(np.float32),
                adv=g["adv"].reshape(-1).astype(np.float32), ret=g["ret"].reshape(-1).astype(np.float32),
                clip=float(g["clip"]))


def euler2quat_sxyz(r, p, y):
    """transforms3d.euler.euler2quat (static xyz, w first) written out from its definition."""
    ci, si, cj, sj, ck, sk = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
    return np.stack([cj * ci * ck + sj * si * sk, cj * si * ck - sj * ci * sk, cj * si * sk + sj * ci * ck,
                     cj * ci * sk - sj * si * ck], -1)


def a3_analytic_cases():
    """Closed-form orientation cases for oly_a3_step (identity, 90 deg rotations, gimbal
    branch cy ~ 0, round trips): body quaternions from known (roll, pitch, yaw), a root that is
    rotated about z only, two targets with known yaws.  Returns state / inputs / expectations."""
    from olympic_hip import _abi
    h = np.pi / 2
    rpy = [(0, 0, 0), (0, 0, 1.1), (0, 0, -3.0), (h, 0, 0), (-h, 0, 0), (0, h, 0), (0, -h, 0), (0, 0, h), (0, 0, -h),
           (0.3, h, 0.0), (h, 0.2, -2.0), (3.0, -1.2, 0.4)]
    rng = np.random.default_rng(4)
    rpy += [(rng.uniform(-np.pi, np.pi), rng.uniform(-1.5, 1.5), rng.uniform(-np.pi, np.pi)) for _ in range(500)]
    rpy = np.array(rpy, dtype=np.float64)
    N = len(rpy)
    body_q = euler2quat_sxyz(rpy[:, 0], rpy[:, 1], rpy[:, 2]) * rng.uniform(0.5, 2.0, (N, 1))   # any norm
    psi = rng.uniform(-np.pi, np.pi, N)                       # root yaw for the goal-step frame
    psi[:3] = [0.0, h, -h]
    root_q = euler2quat_sxyz(np.zeros(N), np.zeros(N), psi)
    theta = rng.uniform(-np.pi, np.pi, (N, 2))                 # target yaws
    root_pos = rng.normal(0, 1, (N, 3))
    tgt = rng.normal(0, 1, (N, 2, 3))
    seq = np.zeros((N, _abi.OLY_MAX_SEQ, 4))
    seq[:, 0, :3], seq[:, 1, :3] = tgt[:, 0], tgt[:, 1]
    seq[:, 0, 3], seq[:, 1, 3] = theta[:, 0], theta[:, 1]
    st = dict(phase=np.zeros(N, np.int32), t1=np.zeros(N, np.int32), t2=np.ones(N, np.int32),
              reached_frames=np.zeros(N, np.int32), target_reached=np.zeros(N, np.uint8),
              mode=np.full(N, _abi.MODE_FORWARD, np.int32), seq_len=np.full(N, 2, np.int32), sequence=seq,
              goal=np.zeros((N, 8)))
    far = np.full((N, 3), 50.0)
    qpos = np.zeros((N, 25))
    qpos[:, 3:7] = body_q
    inp = dict(qpos=qpos, qvel=np.zeros((N, 24)), act_len=np.zeros((N, 12)), act_vel=np.zeros((N, 12)), lf_pos=far,
               rf_pos=far.copy(), lf_vel=np.zeros((N, 3)), rf_vel=np.zeros((N, 3)), root_pos=root_pos, root_quat=root_q,
               head_pos=root_pos.copy(), grf_l=np.zeros(N), grf_r=np.zeros(N), min_z=np.zeros(N),
               n_r=np.zeros(N, np.int32), n_l=np.zeros(N, np.int32), bad=np.zeros(N, np.uint8))
    return dict(state=st, inputs=inp, rpy=rpy, psi=psi, theta=theta, root_pos=root_pos, tgt=tgt)


def check_a3_analytic(obs, goal, c):
    rpy, psi, theta, root_pos, tgt = c["rpy"], c["psi"], c["theta"], c["root_pos"], c["tgt"]
    N, h = len(rpy), np.pi / 2
    # pitch = +-90 deg: roll and yaw are not separable; mat2euler puts the whole in-plane rotation
    # into roll (ak = 0), so the expected roll there is roll -+ yaw
    gimbal = np.isclose(np.abs(rpy[:, 1]), h)
    roll = np.where(gimbal, rpy[:, 0] - np.sign(rpy[:, 1]) * rpy[:, 2], rpy[:, 0])
    want = euler2quat_sxyz(roll, rpy[:, 1], np.zeros(N))
    want = want * np.where(want[:, :1] < 0, -1.0, 1.0)        # roll in (-pi, pi], pitch in [-pi/2, pi/2]: w >= 0
    np.testing.assert_allclose(obs[:, :4], want, rtol=0, atol=2e-12)
    assert np.array_equal(obs[:3, :4].round(15), np.array([[1, 0, 0, 0]] * 3, float))       # identity / pure yaw
    co, si = np.cos(psi), np.sin(psi)
    for i in range(2):                                         # targets in the root's yaw frame
        d = tgt[:, i] - root_pos
        np.testing.assert_allclose(goal[:, 0 + i], co * d[:, 0] + si * d[:, 1], rtol=0, atol=1e-12)
        np.testing.assert_allclose(goal[:, 2 + i], -si * d[:, 0] + co * d[:, 1], rtol=0, atol=1e-12)
        np.testing.assert_allclose(goal[:, 4 + i], d[:, 2], rtol=0, atol=1e-12)
        rel = theta[:, i] - psi
        dth = np.arctan2(np.sin(goal[:, 6 + i] - rel), np.cos(goal[:, 6 + i] - rel))
        np.testing.assert_allclose(dth, 0, atol=1e-12)
    assert np.array_equal(obs[:, 41 - 8:], goal)


def a3_reset_fixture_rollout(g, spec):
    """The a3_reset.npz fixture as the inputs of one RESET_ALL vec step: K = 1 readback row holding the
    fixture's foot / root poses and joint state, one pool record per environment = the captured draws."""
    from olympic_hip import _abi
    from olympic_hip.vecstep import REC
    E = len(g["mode"])
    z = lambda *s: np.zeros((1, E) + s)
    blocks = dict(qpos=g["qpos"][None].copy(), qvel=g["qvel"][None].copy(), act_len=g["act_len"][None].copy(),
                  act_vel=g["act_vel"][None].copy(), lf_pos=g["lfoot"][None].copy(), rf_pos=g["rfoot"][None].copy(),
                  lf_vel=z(3), rf_vel=z(3), root_pos=z(3), root_quat=g["root_quat"][None].copy(), head_pos=z(3),
                  ncon=np.zeros((1, E), np.int32), geom1=np.zeros((1, E, 16), np.int32),
                  geom2=np.zeros((1, E, 16), np.int32), force6=z(16, 6), cpos_z=z(16))
    blocks = {k: np.ascontiguousarray(v) for k, v in blocks.items()}
    pool = np.zeros(E, REC)
    pool["mode"], pool["phase"], pool["seq_len"], pool["seq"] = g["mode"], g["phase"], g["seq_len"], g["local_sequence"]
    zi = lambda dt, *s: np.zeros((E,) + s, dt)
    state = dict(phase=zi(np.int32), t1=zi(np.int32), t2=zi(np.int32), reached_frames=zi(np.int32),
                 target_reached=zi(np.uint8), mode=np.full(E, _abi.MODE_STANDING, np.int32), seq_len=np.ones(E, np.int32),
                 sequence=zi(np.float64, _abi.OLY_MAX_SEQ, 4), goal=np.ones((E, 8)))
    nobs, nu, T, slots = spec.n_obs, spec.nu, 2, 2
    ro = dict(T=T, max_traj_len=10, deterministic=True, side_slots=slots, pool_depth=1, mu=zi(np.float32, nu),
              value=zi(np.float32), scale=None, eps=None, state=zi(np.float32, nobs), pd_target=zi(np.float64, nu),
              buf_states=np.zeros((T, E, nobs), np.float32), buf_actions=np.zeros((T, E, nu), np.float32),
              buf_rewards=np.zeros((T, E)), buf_values=np.zeros((T, E), np.float32), buf_flags=np.zeros((T, E), np.uint8),
              buf_rew6=None, traj_len=zi(np.int32), side_obs=np.zeros((E * slots, nobs), np.float32),
              side_t=np.full(E * slots, -1, np.int32), side_count=zi(np.int32),
              pool=pool.view(np.uint8).reshape(-1).copy(), pool_count=zi(np.int32), ctr=np.zeros(2, np.int32))
    return blocks, state, ro


def check_a3_reset_fixture(g, state, next_obs):
    """State and first observation after the reset against the reference's (a3_reset.npz)."""
    for k in ("mode", "phase", "seq_len", "t1", "t2"):
        assert np.array_equal(state[k], g[k]), k
    assert not state["reached_frames"].any() and not state["target_reached"].any() and not state["goal"].any()
    # transform_sequence: cos / sin / atan2 of the root yaw through libm vs numpy: 1e-12
    np.testing.assert_allclose(state["sequence"], g["sequence"], rtol=1e-12, atol=1e-13)
    want = g["obs"].astype(np.float32)
    assert np.abs(next_obs - want).max() <= np.spacing(np.float32(1.0)) * max(1.0, np.abs(want).max())
    assert not next_obs[:, -8:].any() and not g["obs"][:, -8:].any()          # goal steps zero after reset


# ------------------------------------------------------------------------------ K14 (fused PPO update)
def ppo_update_case(seed, n=96, in_dim=41, act_dim=12, mirror=True, normalize=True, scale=1.0):
    """Random actor / old actor / critic (the reference's shapes), a rollout slice of n rows, a mirror table.
    Weights are drawn so that hidden units are active about half the time and ratios stay O(1)."""
    rng = np.random.default_rng(seed)
    f32 = np.float32

    def net(out, head_scale):
        return [rng.normal(0, scale / np.sqrt(in_dim), (256, in_dim)).astype(f32), rng.normal(0, 0.1, 256).astype(f32),
                rng.normal(0, scale / 16.0, (256, 256)).astype(f32), rng.normal(0, 0.1, 256).astype(f32),
                rng.normal(0, head_scale / 16.0, (out, 256)).astype(f32), rng.normal(0, 0.1, out).astype(f32)]
    actor = net(act_dim, 0.3)
    old = [a + rng.normal(0, 2e-3, a.shape).astype(f32) for a in actor]
    critic = net(1, 1.0)
    obs = rng.normal(0, 1, (n, in_dim)).astype(f32)
    sd = np.full(act_dim, np.exp(-1.5), f32)
    c = dict(obs=obs, actor=actor, old=old, critic=critic, sd=sd, log_sd=np.log(sd).astype(f32),
             adv=rng.normal(0, 1, n).astype(f32), ret=rng.normal(0, 1, n).astype(f32),
             a_mean=rng.normal(0, 0.2, in_dim).astype(f32) if normalize else None,
             a_std=rng.uniform(0.7, 1.4, in_dim).astype(f32) if normalize else None)
    mu = torch_mlp(obs, actor, c["a_mean"], c["a_std"])
    c["action"] = (mu + sd * rng.normal(0, 1, mu.shape)).astype(f32)
    c["old_mu"] = torch_mlp(obs, old, c["a_mean"], c["a_std"])
    if mirror:
        perm = rng.permutation(in_dim)
        c["obs_src"], c["obs_sign"] = perm.astype(np.int32), rng.choice([-1.0, 1.0], in_dim).astype(f32)
        c["mir_obs"] = np.ascontiguousarray(obs[:, perm] * c["obs_sign"])
        c["act_src"] = rng.permutation(act_dim).astype(np.int32)
        c["act_sign"] = rng.choice([-1.0, 1.0], act_dim).astype(f32)
    return c


def torch_mlp(x, wb, mean=None, std=None, grad=False):
    """relu MLP with torch's own float32 Linear layers (numpy in / out unless grad)."""
    import torch
    t = [torch.as_tensor(a) for a in wb] if not grad else wb
    h = torch.as_tensor(x) if not torch.is_tensor(x) else x
    if mean is not None:
        h = (h - torch.as_tensor(mean)) / torch.as_tensor(std)
    h = torch.relu(torch.nn.functional.linear(h, t[0], t[1]))
    h = torch.relu(torch.nn.functional.linear(h, t[2], t[3]))
    y = torch.nn.functional.linear(h, t[4], t[5])
    return y if grad else y.numpy()


def torch_ppo_update_grads(c, idx=None, clip=0.2, vf_coeff=0.5, mirror_coeff=0.4, old_mu=None, float64=False):
    """What the reference's update_policy + the two backward() calls leave in .grad (rl/algos/ppo.py:232-282,
    396-410), evaluated with torch ops and autograd on the CPU in float32: flat gradients in parameter order and
    the six scalars.  `old_mu` overrides the old policy's forward (the kernel takes it as an input).  float64: the same
    graph in double precision on the same float32 inputs (the yardstick for large minibatches, where a float32
    reference's own summation error is as large as the kernel's)."""
    import torch
    if float64:
        up = lambda v: v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v
        c = {k: ([up(a) for a in v] if isinstance(v, (list, tuple)) else up(v)) for k, v in c.items()}
        old_mu = None if old_mu is None else up(old_mu)
    sel = slice(None) if idx is None else np.asarray(idx, np.int64)
    obs, act = torch.as_tensor(c["obs"][sel]), torch.as_tensor(c["action"][sel])
    adv, ret = torch.as_tensor(c["adv"][sel]).reshape(-1, 1), torch.as_tensor(c["ret"][sel]).reshape(-1, 1)
    A = [torch.tensor(a, requires_grad=True) for a in c["actor"]]
    Cw = [torch.tensor(a, requires_grad=True) for a in c["critic"]]
    sd = torch.as_tensor(c["sd"])
    mu = torch_mlp(obs, A, c["a_mean"], c["a_std"], grad=True)
    omu = torch.as_tensor((c["old_mu"] if old_mu is None else old_mu)[sel])
    pdf, old = torch.distributions.Normal(mu, sd), torch.distributions.Normal(omu, sd)
    lp, olp = pdf.log_prob(act).sum(-1, keepdim=True), old.log_prob(act).sum(-1, keepdim=True)
    ratio = (lp - olp).exp()
    cpi, cl = ratio * adv, ratio.clamp(1.0 - clip, 1.0 + clip) * adv
    actor_loss = -torch.min(cpi, cl).mean()
    values = torch_mlp(obs, Cw, grad=True)
    critic_loss = vf_coeff * torch.nn.functional.mse_loss(ret, values)
    ent = -pdf.entropy().mean()
    if "mir_obs" in c and mirror_coeff is not None:
        mir = torch_mlp(torch.as_tensor(c["mir_obs"][sel]), A, c["a_mean"], c["a_std"], grad=True)
        mir = mir[:, torch.as_tensor(c["act_src"].astype(np.int64))] * torch.as_tensor(c["act_sign"])
        mirror_loss = (mu - mir).pow(2).mean()
        mc = mirror_coeff
    else:
        mirror_loss, mc = torch.zeros(()), 0.0
    (actor_loss + mc * mirror_loss + 0.0 * ent).backward()
    critic_loss.backward()
    with torch.no_grad():
        lr = lp - olp
        scal = [float(actor_loss), float(ent), float(critic_loss), float(((ratio - 1) - lr).mean()), float(mirror_loss),
                float(((ratio - 1).abs() > clip).float().mean())]
    flat = lambda ps: np.concatenate([p.grad.reshape(-1).numpy() for p in ps])
    return flat(A), flat(Cw), np.array(scal)


# ------------------------------------------------------------------------------ K13 against the oracle, directly
def check_persistent_rollout_against_oracle(eng, orc, N=70, T=12, max_len=5, K=5, seed=3, det=False, normalize=True):
    """oly_a3_rollout_persistent (K13: all T steps of PPO.sample, rl/algos/ppo.py:169-196, in ONE launch) against a loop
    of the oracle's own functions: oly_mlp_forward_cpu (actor, critic) + oly_a3_vec_step_cpu, step by step.

    The kernel runs all T steps on its own.  The oracle then replays them; at step t its forward reads the observation
    row the KERNEL stored (checked to be within one float32 ulp of the oracle's own: device libm vs glibc in the
    observation's trigonometric terms), so that every later comparison of the step is exact where the arithmetic is:
    mu / value / sampled action / stored value / PD target BIT-EXACT (k-ordered fma chains, f32 mul + add), flags,
    counters, integer task state, cursors BIT-EXACT, float64 rewards 1e-11, goal / sequence 1e-11.  Returns a summary."""
    import torch
    from olympic_hip import _abi, specs
    from olympic_hip.a3 import clock_lut
    from olympic_hip.synthetic import A3_FLOOR_BODY, A3_GEOM_BODYID, A3_LFOOT_BODY, A3_RFOOT_BODY, a3_synthetic_blocks
    from olympic_hip.vecstep import draw_reset_records
    spec = specs.A3Spec(mass=41.5)
    lut = clock_lut(spec.swing_duration, spec.stance_duration, 0.1, "grounded", 1 / spec.control_dt, spec.period)
    contact = (A3_GEOM_BODYID, A3_FLOOR_BODY, A3_RFOOT_BODY, A3_LFOOT_BODY)
    eng.a3_configure(spec, lut)
    eng.contact_configure(*contact)
    rng = np.random.default_rng(seed)
    nobs, nu, depth = spec.n_obs, spec.nu, 3
    slots = T // max_len + 2
    blocks = a3_synthetic_blocks(N, K, seed=seed, p_bad=0.05, p_low=0.02)
    pool = draw_reset_records(np.random.RandomState(seed), N * depth, spec, iter_count=6000)
    z = lambda dt, *sh: np.zeros((N,) + sh, dt)
    state = dict(phase=z(np.int32), t1=z(np.int32), t2=z(np.int32), reached_frames=z(np.int32), target_reached=z(np.uint8),
                 mode=np.full(N, _abi.MODE_STANDING, np.int32), seq_len=np.ones(N, np.int32),
                 sequence=z(np.float64, _abi.OLY_MAX_SEQ, 4), goal=z(np.float64, 8))
    ro = dict(T=T, max_traj_len=max_len, deterministic=det, side_slots=slots, pool_depth=depth, mu=z(np.float32, nu),
              value=z(np.float32), scale=None if det else rng.uniform(0.05, 0.4, nu).astype(np.float32),
              eps=None if det else rng.normal(0, 1, (T, N, nu)).astype(np.float32), state=z(np.float32, nobs),
              pd_target=z(np.float64, nu), buf_states=np.zeros((T, N, nobs), np.float32),
              buf_actions=np.zeros((T, N, nu), np.float32), buf_rewards=np.zeros((T, N)),
              buf_values=np.zeros((T, N), np.float32), buf_flags=np.zeros((T, N), np.uint8), buf_rew6=None,
              traj_len=z(np.int32), side_obs=np.zeros((N * slots, nobs), np.float32), side_t=np.full(N * slots, -1, np.int32),
              side_count=z(np.int32), pool=pool.view(np.uint8).reshape(-1).copy(), pool_count=z(np.int32),
              ctr=np.array([0, 2], np.int32), buf_mu=np.full((T, N, nu), np.nan, np.float32))
    wa = [rng.normal(0, s, sh).astype(np.float32) for s, sh in ((0.2, (256, nobs)), (0.1, (256,)), (0.08, (256, 256)), (0.1, (256,)),
                                                                 (0.05, (nu, 256)), (0.1, (nu,)))]
    wc = [rng.normal(0, s, sh).astype(np.float32) for s, sh in ((0.2, (256, nobs)), (0.1, (256,)), (0.08, (256, 256)), (0.1, (256,)),
                                                                 (0.1, (1, 256)), (0.1, (1,)))]
    mean = rng.normal(0, 0.1, nobs).astype(np.float32) if normalize else None
    std = rng.uniform(0.8, 1.3, nobs).astype(np.float32) if normalize else None
    d = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a)).cuda()
    d_blocks = {k: d(v) for k, v in blocks.items()}
    d_state = {k: d(v) for k, v in state.items()}
    d_ro = {k: (d(v) if isinstance(v, np.ndarray) else v) for k, v in ro.items()}
    ctr = torch.zeros(eng.a3_vec_ctr_len(N), dtype=torch.int32, device="cuda")
    ctr[1:-2:2] = int(ro["ctr"][1])
    d_ro["ctr"] = ctr
    launch = eng.a3_vec_prepare(d_blocks, d_state, d_ro)
    pa, pc = eng.mlp_pack(*[d(a) for a in wa], d(mean), d(std)), eng.mlp_pack(*[d(a) for a in wc])
    launch(_abi.VSTEP_RESET_ALL)
    mu_out = torch.zeros((N, nu), dtype=torch.float32, device="cuda")
    value_out = torch.zeros(N, dtype=torch.float32, device="cuda")
    launch.persistent(pa, normalize, pc, False, mu_out, value_out)     # the whole rollout: ONE launch
    torch.cuda.synchronize()
    h = lambda t_: t_.cpu().numpy()
    got = {k: h(v) for k, v in d_ro.items() if torch.is_tensor(v)}
    got_state = {k: h(v) for k, v in d_state.items()}
    ulp = np.spacing(np.float32(1.0))
    orc.a3_vec_step(spec, lut, contact, blocks, state, ro, _abi.VSTEP_RESET_ALL)
    for t in range(T):
        x = got["buf_states"][t]                                 # the row K13's forward read at step t
        assert np.abs(x - ro["state"]).max() <= ulp * max(1.0, np.abs(ro["state"]).max()), ("observation", t)
        ro["state"][...] = x
        mu = orc.mlp_forward(x, *wa, mean, std)
        val = orc.mlp_forward(x, *wc)[:, 0]
        ro["mu"][...], ro["value"][...] = mu, val
        orc.a3_vec_step(spec, lut, contact, blocks, state, ro, 0)
        assert np.array_equal(got["buf_values"][t], val), ("value", t)
        assert np.array_equal(got["buf_actions"][t], ro["buf_actions"][t]), ("action", t)
        assert np.array_equal(got["buf_mu"][t], mu) and np.array_equal(ro["buf_mu"][t], mu), ("mean", t)
        assert np.array_equal(got["buf_flags"][t], ro["buf_flags"][t]), ("flags", t)
        np.testing.assert_allclose(got["buf_rewards"][t], ro["buf_rewards"][t], rtol=1e-11, atol=1e-13)
    assert np.array_equal(h(mu_out), ro["mu"]) and np.array_equal(h(value_out), ro["value"])     # the last step's forward
    assert np.array_equal(got["pd_target"], ro["pd_target"])
    for k in ("traj_len", "side_count", "side_t", "pool_count"):
        assert np.array_equal(got[k], ro[k]), k
    for k in ("phase", "t1", "t2", "reached_frames", "target_reached", "mode", "seq_len"):
        assert np.array_equal(got_state[k], state[k]), k
    np.testing.assert_allclose(got_state["sequence"], state["sequence"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(got_state["goal"], state["goal"], rtol=1e-11, atol=1e-12)
    for k in ("state", "side_obs"):
        assert np.abs(got[k] - ro[k]).max() <= ulp * max(1.0, np.abs(ro[k]).max()), k
    c = got["ctr"].reshape(-1, 2)
    assert (c[:-1, 0] == T).all() and (c[:-1, 1] == 2 + T).all() and c[-1, 0] == 0
    fl = ro["buf_flags"]
    return dict(N=N, T=T, resets=int(ro["pool_count"].sum()) - N, cuts=int(((fl & _abi.FLAG_LAST) != 0).sum()),
                bootstrap_rows=int((ro["side_t"] >= 0).sum()))
