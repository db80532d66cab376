#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by EXECUTING the reference's
own Python functions (imported from /root/reference under the inert stubs of
_ref_stubs.py) on seeded inputs.  Run in the build container only:

    python tests/golden/gen_golden.py

Outputs are small .npz files holding inputs + expected outputs; no reference
source text is stored.  The reference never travels to the GPU box, these
fixtures do.  Fixture -> reference function map (file:line in /root/reference):

  h1_tables.npz   joint / actuator order and ranges read from
                  olympic_mujoco/environments/data/unitree_h1/h1.xml (data file),
                  spec from UnitreeH1._get_observation_specification (UnitreeH1.py:293)
                  and _get_action_specification (:359), arm removal (:134-160)
  h1_step.npz     LocoEnvBase._create_observation (loco_env_base.py:737),
                  UnitreeH1._has_fallen (UnitreeH1.py:162),
                  BaseHumanoidRobot.is_absorbing (base_humanoid_robot.py:246),
                  TargetVelocityReward.__call__ (utils/reward.py:72),
                  LocoEnvBase._preprocess_action (loco_env_base.py:1050)
  ppo_returns.npz PPOBuffer.store/finish_path (rl/algos/ppo.py:56-84) and the
                  advantage normalisation of PPO.train (:335-336)
  ppo_returns_f64.npz the same scan with float64 rewards that float32 cannot represent
  running_stats.npz RunningMeanStd.update (rl/envs/normalize.py:182-208),
                  Standardizer.update_mean_std (imitation_lib/utils/networks.py:76-81)
  normalize.npz   Normalize._obfilt (rl/envs/normalize.py:139-147) online + frozen
  trajectory.npz  Trajectory.__init__/reset_trajectory/get_next_sample/
                  create_dataset (olympic_mujoco/utils/trajectory.py)
  a3_task.npz     create_phase_reward (tasks/rewards.py:270), WalkingTask.reset/
                  step/calc_reward/done (tasks/walking_task.py), the four reward
                  terms of tasks/rewards.py, StickFigureA3.get_obs
                  (StickFigureA3.py:144), JVRC.step (environments/robot.py:88),
                  MujocoRobotInterface.step_pd (mujoco_robot_interface.py:425)
  a3_reset.npz    WalkingTask.reset + transform_sequence + get_obs of the freshly reset task
                  (walking_task.py:113-135,321-397, StickFigureA3.py:229-232) with the local step
                  sequence captured between generate_step_sequence and transform_sequence
  contacts.npz    MujocoRobotInterface.get_{r,l}foot_floor_contacts (:245-273),
                  get_{r,l}foot_grf (:275-297), check_* (:381-413)
  symmetry.npz    _get_symmetry_matrix (rl/envs/wrappers.py:75),
                  SymmetricEnv.mirror_* (:51-72), A3 tables (StickFigureA3.py:118-129)
  ppo_update.npz  PPO.update_policy (rl/algos/ppo.py:232-282) with Gaussian_FF_Actor / FF_V
                  (rl/policies/actor.py:142, critic.py:37) and the A3 mirror functions
  atlas_tables.npz, talos_tables.npz  spec lists / removal / _has_fallen of atlas.py, talos.py
                  and the joint/actuator order of their MJCF data files
  vail_disc.npz   Standardizer/FullyConnectedNetwork/VariationalNet forward
                  (imitation_lib/utils/networks.py), GAIL.make_discrim_reward
                  (imitation_lib/imitation/gail_TRPO.py:320), GailDiscriminatorLoss
                  and VDBLoss (imitation_lib/utils/math.py)

transforms3d and mushroom-rl are absent: what they would compute is restated in
_ref_stubs.py and those boundaries are "parity unpinned" (SURVEY.md 8c).
"""
import os
import sys
import types
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stubs as stubs  # noqa: E402

REF = stubs.REF
ns = stubs.load_reference()
import torch  # noqa: E402

OT = stubs.ObservationType
AlgorithmType = ns.enums.AlgorithmType


# --out DIR (or GOLDEN_OUT=DIR): write the fixtures somewhere else, e.g. a temporary directory that
# tests/test_oracle_golden.py::test_fixtures_regenerate_bit_identically compares with the committed files
OUT_DIR = os.environ.get("GOLDEN_OUT", HERE)
if "--out" in sys.argv:
    _i = sys.argv.index("--out")
    OUT_DIR = sys.argv[_i + 1]
    del sys.argv[_i:_i + 2]
os.makedirs(OUT_DIR, exist_ok=True)

# fixed per-robot input seeds (hash(cls_name) is salted per process: such fixtures could not be regenerated)
ROBOT_SEEDS = {"Atlas": 711, "Talos": 712}


def save(name, **arrays):
    path = os.path.join(OUT_DIR, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB, keys={list(arrays)}")


# --------------------------------------------------------------------- tables
def walk_mjcf(xml_path, removed_joints=(), removed_motors=()):
    """Document-order walk of an MJCF: joints (qpos/dof addresses), ranges, motors."""
    root = ET.parse(xml_path).getroot()
    joints = []  # (name, qposadr, dofadr, nq, nv, lo, hi)
    qadr = vadr = 0

    def rec(body):
        nonlocal qadr, vadr
        for el in body:
            if el.tag == "freejoint" or (el.tag == "joint" and el.get("type") == "free"):
                joints.append((el.get("name"), qadr, vadr, 7, 6, -np.inf, np.inf))
                qadr += 7
                vadr += 6
            elif el.tag == "joint":
                if el.get("name") in removed_joints:
                    continue
                rng = el.get("range")
                lo, hi = (float(x) for x in rng.split()) if rng else (-np.inf, np.inf)
                joints.append((el.get("name"), qadr, vadr, 1, 1, lo, hi))
                qadr += 1
                vadr += 1
            elif el.tag == "body":
                rec(el)

    rec(root.find("worldbody"))
    motors = []
    default_ctrl = None
    for d in root.iter("default"):
        for m in d.findall("motor"):
            if m.get("ctrlrange"):
                default_ctrl = tuple(float(x) for x in m.get("ctrlrange").split())
    for m in root.find("actuator"):
        if m.get("name") in removed_motors:
            continue
        cr = m.get("ctrlrange")
        cr = tuple(float(x) for x in cr.split()) if cr else default_ctrl
        motors.append((m.get("name"), m.get("joint"), float(m.get("gear", "1").split()[0]), cr))
    return joints, motors, qadr, vadr


def make_h1():
    """A UnitreeH1 instance without running mushroom/mujoco constructors."""
    H1 = ns.h1.UnitreeH1
    env = H1.__new__(H1)
    env._disable_arms = True
    env._disable_back_joint = False
    env._algorithm_type = AlgorithmType.IMITATION_LEARNING
    env._use_foot_forces = False
    env._use_absorbing_states = True
    jr, mr, _ = env._get_xml_modifications()
    spec = H1._get_observation_specification()
    rm = ["q_" + j for j in jr] + ["dq_" + j for j in jr]
    spec = [e for e in spec if e[0] not in rm]          # UnitreeH1.py:77-79
    act = [a for a in H1._get_action_specification() if a not in mr]  # :81
    env.obs_helper = stubs.FakeObservationHelper(spec)
    return env, spec, act, jr, mr


def gen_h1_tables():
    env, spec, act, jr, mr = make_h1()
    xml = f"{REF}/olympic_mujoco/environments/data/unitree_h1/h1.xml"
    joints, motors, nq, nv = walk_mjcf(xml, jr, mr)
    jname = [j[0] for j in joints]
    qadr = {j[0]: j[1] for j in joints}
    vadr = {j[0]: j[2] for j in joints}
    keys = [s[0] for s in spec]
    n_pos = sum(1 for s in spec if s[2] == OT.JOINT_POS)
    qpos_perm = [qadr[s[1]] for s in spec if s[2] == OT.JOINT_POS]
    qvel_perm = [vadr[s[1]] for s in spec if s[2] == OT.JOINT_VEL]
    mname = [m[0] for m in motors]
    act_to_ctrl = [mname.index(a) for a in act]
    lo = np.array([next(j[5] for j in joints if j[0] == s[1]) for s in spec if s[2] == OT.JOINT_POS])
    hi = np.array([next(j[6] for j in joints if j[0] == s[1]) for s in spec if s[2] == OT.JOINT_POS])
    ctrl_lo = np.array([motors[i][3][0] for i in act_to_ctrl])
    ctrl_hi = np.array([motors[i][3][1] for i in act_to_ctrl])
    x_vel_idx = env.get_obs_idx("dq_pelvis_tx")[0]      # loco_env_base.py:1195
    save("h1_tables.npz", spec_keys=np.array(keys), joint_names=np.array(jname),
         nq=nq, nv=nv, n_pos=n_pos, qpos_perm=np.array(qpos_perm), qvel_perm=np.array(qvel_perm),
         action_names=np.array(act), motor_names=np.array(mname), act_to_ctrl=np.array(act_to_ctrl),
         joint_lo=lo, joint_hi=hi, ctrl_lo=ctrl_lo, ctrl_hi=ctrl_hi,
         gear=np.array([m[2] for m in motors]), x_vel_idx=x_vel_idx)
    return dict(keys=keys, lo=lo, hi=hi, ctrl_lo=ctrl_lo, ctrl_hi=ctrl_hi, x_vel_idx=x_vel_idx,
                n_pos=n_pos)


# ---------------------------------------------------------------------- G1 H1
MSG_CODE = {"": 0, "pelvis_y_condition violated.\n": 1, "pelvis_tilt_condition violated.\n": 2,
            "pelvis_list_condition violated.\n": 3, "pelvis_rotation_condition violated.\n": 4}


def h1_synthetic_rows(rng, n, lo, hi):
    """SURVEY 8c generator (ii): joint angles U(range), pelvis free-ish, dq ~ N(0,1.5)."""
    q = np.empty((n, 17))
    q[:, 0:2] = rng.uniform(-5, 5, (n, 2))              # pelvis x, (world) y
    q[:, 2] = rng.uniform(-0.4, 0.2, n)                 # pelvis height
    q[:, 3:6] = rng.uniform(-0.6, 0.6, (n, 3))          # tilt, list, rotation
    q[:, 6:] = rng.uniform(lo[6:], hi[6:], (n, 11))
    dq = rng.normal(0.0, 1.5, (n, 17))
    return np.concatenate([q, dq], axis=1)


def gen_h1_step(tab):
    env, spec, act, _, _ = make_h1()
    keys = tab["keys"]
    rng = np.random.default_rng(20240601)
    rows = []
    src = []
    for tag, f in ((1, "vail_unprocessed_0"), (2, "gail_unprocessed_0")):
        d = np.load(f"{REF}/saved_npz/{f}.npz", allow_pickle=False)
        m = np.stack([d[k] for k in keys], axis=1)
        keep = [0] + [i for i in range(1, len(m)) if not np.array_equal(m[i], m[i - 1])]
        rows.append(m[keep])
        src += [tag] * len(keep)
    syn = h1_synthetic_rows(rng, 1024, tab["lo"], tab["hi"])
    rows.append(syn)
    src += [3] * len(syn)
    # boundary rows: each threshold of UnitreeH1._has_fallen exactly, and one ulp either side
    thr = [(2, -0.3), (2, 0.1), (3, -np.pi / 4.5), (3, np.pi / 12), (4, -np.pi / 12),
           (4, np.pi / 8), (5, -np.pi / 8), (5, np.pi / 8)]
    b = []
    for col, t in thr:
        for v in (t, np.nextafter(t, -np.inf), np.nextafter(t, np.inf)):
            r = np.zeros(34)
            r[6:17] = rng.uniform(tab["lo"][6:], tab["hi"][6:])
            r[17:] = rng.normal(0, 1, 17)
            r[col] = v
            b.append(r)
    rows.append(np.array(b))
    src += [4] * len(b)
    full = np.concatenate(rows, axis=0)
    M = len(full)

    env._reward_function = ns.reward.TargetVelocityReward(target_velocity=1.25,
                                                          x_vel_idx=tab["x_vel_idx"])
    env.norm_act_mean = (tab["ctrl_hi"] + tab["ctrl_lo"]) / 2.0     # loco_env_base.py:171-172
    env.norm_act_delta = (tab["ctrl_hi"] - tab["ctrl_lo"]) / 2.0
    action = rng.uniform(-1.5, 1.5, (M, len(act)))
    obs = np.empty((M, 32))
    fallen = np.zeros(M, dtype=bool)
    absorbing = np.zeros(M, dtype=bool)
    code = np.zeros(M, dtype=np.int32)
    rew_state = np.empty(M)
    rew_run = np.empty(M)
    unnorm = np.empty_like(action)
    for i in range(M):
        o = env._create_observation(full[i])
        obs[i] = o
        f, msg = env._has_fallen(o, return_err_msg=True)
        fallen[i] = f
        code[i] = MSG_CODE[msg]
        absorbing[i] = env.is_absorbing(o)
        rew_state[i] = env.reward(o, action[i], o, absorbing[i])
        unnorm[i] = env._preprocess_action(action[i])
    env._reward_function = ns.reward.TargetVelocityReward(target_velocity=2.5,
                                                          x_vel_idx=tab["x_vel_idx"])
    for i in range(M):
        rew_run[i] = env.reward(obs[i], action[i], obs[i], absorbing[i])
    env._use_absorbing_states = False
    assert not any(env.is_absorbing(o) for o in obs[:50])
    save("h1_step.npz", full_obs=full, source=np.array(src, dtype=np.int8), obs=obs, fallen=fallen,
         absorbing=absorbing, msg_code=code, reward_walk=rew_state, reward_run=rew_run,
         action=action, unnorm_action=unnorm)


# ---------------------------------------------------------------------- G2 PPO
def gen_ppo():
    rng = np.random.default_rng(7)
    gamma = 0.99
    buf = ns.ppo.PPOBuffer(gamma, 0.95)
    ep_len = [400, 1, 2, 37, 400, 5, 123, 400, 64, 3]
    done_tail = [False, True, True, True, False, True, False, False, True, False]
    rewards, values, last_vals = [], [], []
    for L, dn in zip(ep_len, done_tail):
        for _ in range(L):
            r = np.float32(rng.uniform(-0.3, 1.0))
            v = np.float32(rng.normal())
            # shapes exactly as PPO.sample stores them (ppo.py:184-186)
            buf.store(np.zeros((1, 4), np.float32), np.zeros((1, 2), np.float32),
                      np.array([np.float64(r)]), np.array([[v]], dtype=np.float32))
            rewards.append(r)
            values.append(v)
        lv = np.array([[np.float32(rng.normal())]], dtype=np.float32)
        last_vals.append(lv[0, 0])
        buf.finish_path(last_val=(not dn) * lv)
    _, _, returns, vals = buf.get()
    returns_t = torch.Tensor(np.array(returns))
    values_t = torch.Tensor(np.array(vals))
    adv = returns_t - values_t
    eps = 1e-5
    adv_n = (adv - adv.mean()) / (adv.std() + eps)          # ppo.py:335-336
    save("ppo_returns.npz", gamma=gamma, eps=eps, ep_len=np.array(ep_len),
         done_tail=np.array(done_tail), rewards=np.array(rewards, np.float32),
         values=np.array(values, np.float32), last_val=np.array(last_vals, np.float32),
         returns=returns_t.numpy().reshape(-1), adv=adv.numpy().reshape(-1),
         adv_norm=adv_n.numpy().reshape(-1), ep_returns=np.array(buf.ep_returns, np.float64),
         ep_lens=np.array(buf.ep_lens), traj_idx=np.array(buf.traj_idx))


def gen_ppo_f64():
    """finish_path with rewards that are NOT float32-representable: WrapEnv.step returns
    np.array([reward]) (float64, rl/envs/wrappers.py:14) and PPOBuffer keeps it float64, so the
    scan adds the un-narrowed reward (rl/algos/ppo.py:74-76)."""
    rng = np.random.default_rng(71)
    gamma = 0.99
    buf = ns.ppo.PPOBuffer(gamma, 0.95)
    ep_len = [400, 3, 1, 55, 400, 17, 230, 9]
    done_tail = [False, True, False, True, True, False, False, True]
    rewards, values, last_vals = [], [], []
    for L, dn in zip(ep_len, done_tail):
        for _ in range(L):
            r = float(rng.uniform(-0.3, 1.0))                  # a python float, as sum([float(i) ...]) gives
            v = np.float32(rng.normal())
            buf.store(np.zeros((1, 4), np.float32), np.zeros((1, 2), np.float32),
                      np.array([r]), np.array([[v]], dtype=np.float32))
            rewards.append(r)
            values.append(v)
        lv = np.array([[np.float32(rng.normal())]], dtype=np.float32)
        last_vals.append(lv[0, 0])
        buf.finish_path(last_val=(not dn) * lv)
    _, _, returns, vals = buf.get()
    returns_t = torch.Tensor(np.array(returns))
    adv = returns_t - torch.Tensor(np.array(vals))
    assert np.any(np.array(rewards, np.float32).astype(np.float64) != np.array(rewards))
    save("ppo_returns_f64.npz", gamma=gamma, ep_len=np.array(ep_len), done_tail=np.array(done_tail),
         rewards=np.array(rewards, np.float64), values=np.array(values, np.float32),
         last_val=np.array(last_vals, np.float32), returns=returns_t.numpy().reshape(-1),
         adv=adv.numpy().reshape(-1), ep_returns=np.array(buf.ep_returns, np.float64))


# ------------------------------------------------------------ G3 running stats
def gen_running_stats():
    rng = np.random.default_rng(11)
    xs = [rng.normal(0.3, 2.0, (n, 5)) for n in (3, 4, 50, 1, 128)]
    rms = ns.normalize.RunningMeanStd(shape=(5,))
    rms_mean, rms_var, rms_count = [], [], []
    for x in xs:
        rms.update(x)
        rms_mean.append(rms.mean.copy())
        rms_var.append(rms.var.copy())
        rms_count.append(rms.count)
    st = ns.networks.Standardizer()
    st_mean, st_std = [], []
    xs32 = [x.astype(np.float32) for x in xs]
    for x in xs32:
        st.update_mean_std(x)
        st_mean.append(np.array(st.mean, dtype=np.float64).copy())
        st_std.append(np.array(st.std, dtype=np.float64).copy())
    out = st.forward(torch.tensor(xs32[2]))
    save("running_stats.npz", lens=np.array([len(x) for x in xs]), x=np.concatenate(xs),
         rms_mean=np.array(rms_mean), rms_var=np.array(rms_var), rms_count=np.array(rms_count),
         st_mean=np.array(st_mean), st_std=np.array(st_std), st_fwd_in=xs32[2],
         st_fwd_out=out.numpy(), st_fwd_mean=np.array(st.mean), st_fwd_std=np.array(st.std))


def gen_normalize():
    """Normalize._obfilt (rl/envs/normalize.py:139-147) online and frozen, on float32 batches
    (the vectorised envs hand float32 observations to the filter)."""
    rng = np.random.default_rng(12)

    class Venv:
        observation_space = np.zeros(6)
        action_space = np.zeros(2)
        num_envs = 1
    nz = ns.normalize.Normalize(Venv(), clipob=2.5)
    xs = [rng.normal(0.5, 3.0, (n, 6)).astype(np.float32) for n in (7, 33, 64)]
    outs, means, vars_, counts = [], [], [], []
    for x in xs:
        outs.append(nz._obfilt(x))
        means.append(nz.ob_rms.mean.copy())
        vars_.append(nz.ob_rms.var.copy())
        counts.append(nz.ob_rms.count)
    nz.online = False
    frozen_in = rng.normal(0.0, 6.0, (40, 6)).astype(np.float32)
    frozen_out = nz._obfilt(frozen_in)
    save("normalize.npz", lens=np.array([len(x) for x in xs]), x=np.concatenate(xs), out=np.concatenate(outs),
         mean=np.array(means), var=np.array(vars_), count=np.array(counts), clipob=2.5, epsilon=1e-8,
         frozen_in=frozen_in, frozen_out=frozen_out)


# --------------------------------------------------------------- G4 trajectory
def gen_trajectory(tab):
    keys = list(tab["keys"])
    rng = np.random.default_rng(3)
    n_traj, L = 2, 250
    t = np.arange(n_traj * L) * 0.002
    data = {}
    lo = np.concatenate([[-np.inf, -np.inf], tab["lo"][2:]])
    hi = np.concatenate([[np.inf, np.inf], tab["hi"][2:]])
    for i, k in enumerate(keys[:17]):
        amp = rng.uniform(0.1, 0.5)
        w = rng.uniform(2.0, 9.0)
        ph = rng.uniform(0, 2 * np.pi)
        if i == 0:
            q = 1.25 * t + 0.02 * np.sin(w * t)
            dq = 1.25 + 0.02 * w * np.cos(w * t)
        elif i == 2:
            q = -0.05 + 0.03 * np.sin(w * t + ph)
            dq = 0.03 * w * np.cos(w * t + ph)
        elif i in (3, 4, 5):
            q = 0.1 * np.sin(w * t + ph)
            dq = 0.1 * w * np.cos(w * t + ph)
        else:
            q = amp * np.sin(w * t + ph)
            dq = amp * w * np.cos(w * t + ph)
        data[k] = q
        data["d" + k] = dq
    # make two joints violate their range so that clipping is exercised
    data["q_knee_angle_r"] = data["q_knee_angle_r"] * 3.0 + 1.2
    data["q_hip_rotation_l"] = data["q_hip_rotation_l"] * 2.0
    data["split_points"] = np.array([0, L, 2 * L])
    path = os.path.join(HERE, "_tmp_traj_in.npz")
    np.savez(path, **{k: data[k] for k in keys + ["split_points"]})

    joint_pos_idx = np.arange(17)
    low32 = np.concatenate([lo[2:], -np.inf * np.ones(17)])     # observation_space.low (x,y dropped)
    high32 = np.concatenate([hi[2:], np.inf * np.ones(17)])
    LB = ns.loco.LocoEnvBase
    np.random.seed(123)
    tr = ns.trajectory.Trajectory(keys=list(keys), low=low32, high=high32,
                                  joint_pos_idx=joint_pos_idx,
                                  interpolate_map=LB._interpolate_map,
                                  interpolate_remap=LB._interpolate_remap,
                                  traj_path=path, traj_dt=1 / 500, control_dt=0.01,
                                  clip_trajectory_to_joint_ranges=True, warn=False)
    table = np.array(tr.trajectories)            # [34, n_traj, L100]
    sp = np.array(tr.split_points)
    # fixed resets
    resets = [(0, 0), (7, 1), (49, 0), (25, 1)]
    reset_samples = []
    for sub, tno in resets:
        s = tr.reset_trajectory(sub, tno)
        reset_samples.append(np.array(s, dtype=np.float64).ravel())
    # random reset (np.random stream) then walk to the end
    np.random.seed(99)
    s = tr.reset_trajectory()
    rnd = (tr.subtraj_step_no, tr.traj_no)
    walk = [np.array(s, dtype=np.float64).ravel()]
    cur = np.concatenate(tr.get_current_sample())
    assert np.array_equal(cur, walk[0])
    n_none = 0
    while True:
        s = tr.get_next_sample()
        if s is None:
            n_none += 1
            break
        walk.append(np.concatenate(s))
    ds = tr.create_dataset(ignore_keys=["q_pelvis_tx", "q_pelvis_tz"])
    raw = np.stack([data[k] for k in keys])
    os.remove(path)
    save("trajectory.npz", keys=np.array(keys), raw=raw, raw_split_points=data["split_points"],
         low=low32, high=high32, traj_dt=1 / 500, control_dt=0.01,
         table=table, split_points=sp, resets=np.array(resets), reset_samples=np.array(reset_samples),
         rnd_reset=np.array(rnd), walk=np.array(walk), ds_states=ds["states"],
         ds_next_states=ds["next_states"], ds_absorbing=ds["absorbing"], ds_last=ds["last"])


# ------------------------------------------------------------------ G5/6 A3 RL
class Contact:
    def __init__(self, g1, g2, pos):
        self.geom1, self.geom2, self.pos = g1, g2, pos


class FakeA3Client:
    """Stands in for MujocoRobotInterface readback: returns the synthetic per-step
    quantities MuJoCo would have produced.  The contact filters/GRF/predicates are
    the reference's own methods, bound to this object below."""

    def __init__(self, mass, geom_bodyid, body_ids):
        self._mass = mass
        self.model = types.SimpleNamespace(geom_bodyid=geom_bodyid, nu=12,
                                           geom=lambda n: types.SimpleNamespace(pos=np.zeros(3)))
        self.data = types.SimpleNamespace(ncon=0, contact=[])
        self.body_ids = body_ids
        self.rfoot_body_name, self.lfoot_body_name, self.floor_body_name = \
            "right_foot", "left_foot", "world"
        self.force6 = np.zeros((0, 6))
        self.state = {}

    def get_robot_mass(self):
        return self._mass

    def get_object_xpos_by_name(self, name, typ):
        return self.state["xpos_" + name]

    def get_object_xquat_by_name(self, name, typ):
        return self.state["xquat_" + name]

    def get_lfoot_body_pos(self):
        return self.state["xpos_left_foot"].copy()

    def get_rfoot_body_pos(self):
        return self.state["xpos_right_foot"].copy()

    def get_lfoot_body_vel(self):
        return [self.state["lvel"], np.zeros(3)]

    def get_rfoot_body_vel(self):
        return [self.state["rvel"], np.zeros(3)]


def install_mujoco_lookups(client):
    mj = sys.modules["mujoco"]
    mj.mjtObj = types.SimpleNamespace(mjOBJ_BODY=1)
    mj.mj_name2id = lambda model, typ, name: client.body_ids[name]

    def mj_contactForce(model, data, i, out):
        out[:] = client.force6[i]
    mj.mj_contactForce = mj_contactForce


A3_GEOMS = ["floor", "torso", "head", "lower_waist", "butt", "right_thigh", "right_shin",
            "right_foot", "right_foot_sole", "left_thigh", "left_shin", "left_foot",
            "left_foot_sole"]
A3_BODIES = {"world": 0, "torso": 1, "head": 2, "lower_waist": 3, "pelvis": 4, "right_thigh": 5,
             "right_shin": 6, "right_foot": 7, "left_thigh": 8, "left_shin": 9, "left_foot": 10}
A3_GEOM_BODY = np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], dtype=np.int32)


def random_contacts(rng, C=16):
    """ncon ~ Poisson(4) clipped to C, ~70 % foot-floor in (floor, foot) order."""
    ncon = int(min(C, rng.poisson(4)))
    g1 = np.zeros(C, np.int32)
    g2 = np.zeros(C, np.int32)
    f6 = np.zeros((C, 6))
    pos = np.zeros((C, 3))
    for i in range(ncon):
        u = rng.uniform()
        if u < 0.35:
            g1[i], g2[i] = 0, 8            # floor, right sole
        elif u < 0.70:
            g1[i], g2[i] = 0, 12           # floor, left sole
        elif u < 0.80:
            g1[i], g2[i] = 8, 0            # wrong order: not a foot-floor contact for the filter
        elif u < 0.90:
            g1[i], g2[i] = 0, int(rng.integers(1, 7))   # floor vs other body
        else:
            g1[i], g2[i] = int(rng.integers(1, 13)), int(rng.integers(1, 13))
        f6[i] = rng.normal(0, 120, 6)
        pos[i] = [rng.normal(), rng.normal(), rng.uniform(-0.02, 0.05)]
    return ncon, g1, g2, f6, pos


def bind_contacts(client):
    M = ns.mri.MujocoRobotInterface
    for n in ("get_rfoot_floor_contacts", "get_lfoot_floor_contacts", "get_rfoot_grf",
              "get_lfoot_grf", "check_rfoot_floor_collision", "check_lfoot_floor_collision",
              "check_bad_collisions", "check_self_collisions"):
        setattr(client, n, types.MethodType(getattr(M, n), client))


def set_contacts(client, ncon, g1, g2, f6, pos):
    client.data.ncon = ncon
    client.data.contact = [Contact(int(g1[i]), int(g2[i]), pos[i]) for i in range(ncon)]
    client.force6 = f6


def gen_contacts():
    rng = np.random.default_rng(5)
    client = FakeA3Client(40.0, A3_GEOM_BODY, A3_BODIES)
    install_mujoco_lookups(client)
    bind_contacts(client)
    N, C = 512, 16
    out = dict(ncon=np.zeros(N, np.int32), geom1=np.zeros((N, C), np.int32),
               geom2=np.zeros((N, C), np.int32), force6=np.zeros((N, C, 6)),
               pos=np.zeros((N, C, 3)), n_r=np.zeros(N, np.int32), n_l=np.zeros(N, np.int32),
               idx_r=-np.ones((N, C), np.int32), idx_l=-np.ones((N, C), np.int32),
               grf_r=np.zeros(N), grf_l=np.zeros(N), bad=np.zeros(N, bool),
               self_col=np.zeros(N, bool), min_z=np.zeros(N), any_foot=np.zeros(N, bool))
    for n in range(N):
        if n == 0:
            ncon, g1, g2, f6, pos = 0, np.zeros(C, np.int32), np.zeros(C, np.int32), \
                np.zeros((C, 6)), np.zeros((C, 3))
        elif n == 1:                                     # maximum: all C slots, all right foot
            ncon, g1, g2 = C, np.zeros(C, np.int32), np.full(C, 8, np.int32)
            f6, pos = rng.normal(0, 50, (C, 6)), rng.normal(0, 1, (C, 3))
        else:
            ncon, g1, g2, f6, pos = random_contacts(rng, C)
        set_contacts(client, ncon, g1, g2, f6, pos)
        rc = client.get_rfoot_floor_contacts()
        lc = client.get_lfoot_floor_contacts()
        out["ncon"][n] = ncon
        out["geom1"][n], out["geom2"][n], out["force6"][n], out["pos"][n] = g1, g2, f6, pos
        out["n_r"][n], out["n_l"][n] = len(rc), len(lc)
        out["idx_r"][n, :len(rc)] = [i for i, _ in rc]
        out["idx_l"][n, :len(lc)] = [i for i, _ in lc]
        out["grf_r"][n] = client.get_rfoot_grf()
        out["grf_l"][n] = client.get_lfoot_grf()
        out["bad"][n] = client.check_bad_collisions()
        out["self_col"][n] = client.check_self_collisions()
        # tasks/rewards.py:29-33 contact point used by the height reward
        if client.check_rfoot_floor_collision() or client.check_lfoot_floor_collision():
            out["min_z"][n] = min(c.pos[2] for _, c in rc + lc)
            out["any_foot"][n] = True
    save("contacts.npz", geom_bodyid=A3_GEOM_BODY, floor_body=0, rfoot_body=7, lfoot_body=10, **out)


def quat_from_rpy(r, p, y):
    return stubs._euler2quat(r, p, y)


def gen_a3_task():
    rng = np.random.default_rng(17)
    mass = 41.5
    client = FakeA3Client(mass, A3_GEOM_BODY, A3_BODIES)
    install_mujoco_lookups(client)
    bind_contacts(client)
    cwd = os.getcwd()
    os.chdir(REF)                                        # walking_task.py:42 opens a CWD-relative file
    try:
        WT = ns.walking_task.WalkingTask
    finally:
        pass
    control_dt = 0.025
    # clock LUT, exactly the call of WalkingTask.reset (walking_task.py:346-350)
    rc, lc = ns.rewards.create_phase_reward(0.75, 0.35, 0.1, "grounded", 1 / control_dt)
    period = int(np.floor(2 * 1.1 * (1 / control_dt)))
    ph = np.arange(period)
    lut = np.stack([rc[0](ph), rc[1](ph), lc[0](ph), lc[1](ph)])   # r_frc, r_vel, l_frc, l_vel

    E, K, C = 16, 100, 16          # envs, steps per env, contact slots
    rec = dict(
        reset_lfoot=np.zeros((E, 3)), reset_rfoot=np.zeros((E, 3)), reset_root_quat=np.zeros((E, 4)),
        iter_count=np.zeros(E, np.int64),
        mode=np.zeros(E, np.int32), phase0=np.zeros(E, np.int32), seq_len=np.zeros(E, np.int32),
        sequence=np.zeros((E, 20, 4)), t1_0=np.zeros(E, np.int32), t2_0=np.zeros(E, np.int32),
        root_pos=np.zeros((E, K, 3)), root_quat=np.zeros((E, K, 4)), head_pos=np.zeros((E, K, 3)),
        lf_pos=np.zeros((E, K, 3)), rf_pos=np.zeros((E, K, 3)), lf_vel=np.zeros((E, K, 3)),
        rf_vel=np.zeros((E, K, 3)), ncon=np.zeros((E, K), np.int32),
        geom1=np.zeros((E, K, C), np.int32), geom2=np.zeros((E, K, C), np.int32),
        force6=np.zeros((E, K, C, 6)), cpos_z=np.zeros((E, K, C)),
        qpos=np.zeros((E, K, 25)), qvel=np.zeros((E, K, 24)), act_len=np.zeros((E, K, 12)),
        act_vel=np.zeros((E, K, 12)),
        # outputs
        phase=np.zeros((E, K), np.int32), t1=np.zeros((E, K), np.int32), t2=np.zeros((E, K), np.int32),
        target_reached=np.zeros((E, K), bool), reached_frames=np.zeros((E, K), np.int32),
        goal=np.zeros((E, K, 8)), rew6=np.zeros((E, K, 6)), reward=np.zeros((E, K)),
        done=np.zeros((E, K), bool), obs=np.zeros((E, K, 41)), grf_l=np.zeros((E, K)),
        grf_r=np.zeros((E, K)),
    )
    A3 = ns.a3.StickFigureA3
    for e in range(E):
        task = WT(client=client, dt=control_dt, neutral_foot_orient=np.array([1, 0, 0, 0]),
                  root_body="torso", lfoot_body="left_foot", rfoot_body="right_foot",
                  head_body="head")
        task._goal_height_ref = 0.80                      # StickFigureA3.py:110-113
        task._total_duration = 1.1
        task._swing_duration = 0.75
        task._stance_duration = 0.35
        yaw0 = rng.uniform(-np.pi, np.pi)
        base = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), 0.0])
        client.state = {
            "xpos_left_foot": base + [0, 0.1, 0.03], "xpos_right_foot": base + [0, -0.1, 0.03],
            "xquat_torso": quat_from_rpy(0, 0, yaw0),
        }
        np.random.seed(1000 + e)
        task.reset(iter_count=5000 + 400 * e)
        rec["reset_lfoot"][e] = client.state["xpos_left_foot"]
        rec["reset_rfoot"][e] = client.state["xpos_right_foot"]
        rec["reset_root_quat"][e] = client.state["xquat_torso"]
        rec["iter_count"][e] = 5000 + 400 * e
        rec["mode"][e] = task.mode.value
        rec["phase0"][e] = task._phase
        rec["seq_len"][e] = len(task.sequence)
        rec["sequence"][e, :len(task.sequence)] = np.array(task.sequence)
        rec["t1_0"][e], rec["t2_0"][e] = task.t1, task.t2

        env = A3.__new__(A3)
        env._algorithm_type = AlgorithmType.REINFORCEMENT_LEARNING
        env.task = task
        env.actuators = list(range(12))
        env.base_obs_len = 41
        iface = types.SimpleNamespace()
        env.interface = iface

        seq = np.array(task.sequence)
        cy, sy = np.cos(yaw0), np.sin(yaw0)
        for k in range(K):
            # a walker that roughly follows the step sequence so targets get reached
            tgt = seq[min(task.t1, len(seq) - 1)]
            prog = min(1.0, (k % 45) / 30.0)
            foot_on = tgt[:3] + rng.normal(0, 0.03, 3) * (1.5 - prog)
            other = tgt[:3] + np.array([-0.3 * cy, -0.3 * sy, 0.0]) + rng.normal(0, 0.05, 3)
            if (k // 45) % 2 == 0:
                lf, rf = foot_on, other
            else:
                lf, rf = other, foot_on
            lf[2] = abs(lf[2]) * 0.3
            rf[2] = abs(rf[2]) * 0.3
            root = (lf + rf) / 2 + np.array([0, 0, rng.uniform(0.55, 0.95)])
            rq = quat_from_rpy(rng.normal(0, 0.1), rng.normal(0, 0.1), yaw0 + rng.normal(0, 0.2))
            rq = rq * rng.choice([-1.0, 1.0])
            head = root + np.array([rng.normal(0, 0.05), rng.normal(0, 0.05), 0.45])
            lv, rv = rng.normal(0, 0.25, 3), rng.normal(0, 0.25, 3)
            if k % 7 == 0:
                lv *= 0.05
            ncon, g1, g2, f6, pos = random_contacts(rng, C)
            if k % 5 == 0:       # keep some steps free of "bad" contacts so episodes are not all done
                keep = [(0, 8), (0, 12)]
                for i in range(ncon):
                    if (g1[i], g2[i]) not in keep:
                        g1[i], g2[i] = keep[i % 2]
            qpos = np.concatenate([root, rq, rng.uniform(-1, 1, 18)])
            qvel = rng.normal(0, 1, 24)
            alen, avel = rng.uniform(-1, 1, 12), rng.normal(0, 2, 12)

            client.state.update({
                "xpos_torso": root, "xquat_torso": rq, "xpos_head": head,
                "xpos_lf_force": lf, "xpos_rf_force": rf,
                "xquat_lf_force": np.array([1.0, 0, 0, 0]), "xquat_rf_force": np.array([1.0, 0, 0, 0]),
                "lvel": lv, "rvel": rv,
            })
            set_contacts(client, ncon, g1, g2, f6, pos)
            iface.get_qpos = lambda qpos=qpos: qpos
            iface.get_qvel = lambda qvel=qvel: qvel
            iface.get_act_joint_positions = lambda alen=alen: list(alen)   # gear = 1 (a3.xml:132-143)
            iface.get_act_joint_velocities = lambda avel=avel: list(avel)

            task.step()
            rewards = task.calc_reward(None, None, None)
            total = sum(float(i) for i in rewards.values())          # StickFigureA3.py:194
            done = task.done()
            obs = env.get_obs()

            for name, v in (("root_pos", root), ("root_quat", rq), ("head_pos", head), ("lf_pos", lf),
                            ("rf_pos", rf), ("lf_vel", lv), ("rf_vel", rv), ("qpos", qpos),
                            ("qvel", qvel), ("act_len", alen), ("act_vel", avel)):
                rec[name][e, k] = v
            rec["ncon"][e, k] = ncon
            rec["geom1"][e, k], rec["geom2"][e, k] = g1, g2
            rec["force6"][e, k], rec["cpos_z"][e, k] = f6, pos[:, 2]
            rec["phase"][e, k] = task._phase
            rec["t1"][e, k], rec["t2"][e, k] = task.t1, task.t2
            rec["target_reached"][e, k] = task.target_reached
            rec["reached_frames"][e, k] = task.target_reached_frames
            rec["goal"][e, k] = np.concatenate([task._goal_steps_x, task._goal_steps_y,
                                                task._goal_steps_z, task._goal_steps_theta])
            rec["rew6"][e, k] = [float(rewards[n]) for n in
                                 ("foot_frc_score", "foot_vel_score", "orient_cost", "height_error",
                                  "step_reward", "upper_body_reward")]
            rec["reward"][e, k] = total
            rec["done"][e, k] = done
            rec["obs"][e, k] = obs
            rec["grf_l"][e, k], rec["grf_r"][e, k] = task.l_foot_frc, task.r_foot_frc
    os.chdir(cwd)
    print("a3_task: modes", np.bincount(rec["mode"]), "done frac", rec["done"].mean(),
          "max t1", rec["t1"].max(), "reached", rec["target_reached"].mean())

    # JVRC.step target + PD torque (environments/robot.py:88-115, mujoco_robot_interface.py:425-443)
    kp = 0.5 * np.array([200, 200, 200, 250, 80, 80] * 2, dtype=np.float64)   # StickFigureA3.py:78-85
    kd = 0.5 * np.array([20, 20, 20, 25, 8, 8] * 2, dtype=np.float64)
    pdc = types.SimpleNamespace()
    pdc.nu = lambda: 12
    pdc.nq = lambda: 25
    pdc.nv = lambda: 24
    pdc.sim_dt = lambda: 0.0025
    pdc.model = types.SimpleNamespace(nu=12)
    M = ns.mri.MujocoRobotInterface
    pdc.set_pd_gains = types.MethodType(M.set_pd_gains, pdc)
    pdc.step_pd = types.MethodType(M.step_pd, pdc)
    pdc.get_motor_qposadr = lambda: list(range(7, 19))                      # a3.xml joint order
    pdc.get_gear_ratios = lambda: np.ones(12)
    pdc.get_act_joint_torques = lambda: [0.0] * 12
    sub = dict(i=0)
    B, S = 64, 10
    q_seq = rng.uniform(-1, 1, (B, S, 12))
    qd_seq = rng.normal(0, 2, (B, S, 12))
    taus = np.zeros((B, S, 12))
    pdc.get_act_joint_positions = lambda: list(q_seq[sub["b"], sub["i"]])
    pdc.get_act_joint_velocities = lambda: list(qd_seq[sub["b"], sub["i"]])

    def set_motor_torque(t):
        taus[sub["b"], sub["i"]] = np.asarray(t)
    pdc.set_motor_torque = set_motor_torque

    def step():
        sub["i"] += 1
    pdc.step = step
    jv = ns.robot.JVRC(np.stack([kp, kd]), control_dt, list(range(12)), pdc)
    acts = rng.uniform(-1, 1, (B, 12))
    targets = np.zeros((B, 12))
    for b in range(B):
        sub["b"], sub["i"] = b, 0
        targets[b] = jv.step(acts[b])
    save("a3_task.npz", clock_lut=lut, period=period, mass=mass, control_dt=control_dt,
         goal_height_ref=0.80, target_radius=0.20, delay_frames=int(np.floor(0.75 / control_dt)),
         geom_bodyid=A3_GEOM_BODY, floor_body=0, rfoot_body=7, lfoot_body=10,
         pd_kp=kp, pd_kd=kd, pd_action=acts, pd_target=targets, motor_offset=np.array(jv.motor_offset),
         pd_q=q_seq, pd_qd=qd_seq, pd_tau=taus, **rec)


def gen_a3_reset():
    """StickFigureA3.reset_model's task half: WalkingTask.reset (walking_task.py:321-397: mode / phase draws,
    generate_step_sequence, transform_sequence about the feet's mid point and the root yaw) followed by
    get_obs of the un-advanced task (StickFigureA3.py:229-232): goal steps zero, clock of the drawn phase.
    The local step sequence (generate_step_sequence's return value) is captured on its way into
    transform_sequence: it is what the device-side reset takes as its pre-drawn record."""
    rng = np.random.default_rng(23)
    client = FakeA3Client(41.5, A3_GEOM_BODY, A3_BODIES)
    install_mujoco_lookups(client)
    cwd = os.getcwd()
    os.chdir(REF)                                        # walking_task.py:42 opens a CWD-relative file
    WT = ns.walking_task.WalkingTask
    A3 = ns.a3.StickFigureA3
    E = 48
    rec = dict(lfoot=np.zeros((E, 3)), rfoot=np.zeros((E, 3)), root_quat=np.zeros((E, 4)), iter_count=np.zeros(E, np.int64),
               qpos=np.zeros((E, 25)), qvel=np.zeros((E, 24)), act_len=np.zeros((E, 12)), act_vel=np.zeros((E, 12)),
               mode=np.zeros(E, np.int32), phase=np.zeros(E, np.int32), seq_len=np.zeros(E, np.int32),
               local_sequence=np.zeros((E, 20, 4)), sequence=np.zeros((E, 20, 4)), t1=np.zeros(E, np.int32),
               t2=np.zeros(E, np.int32), obs=np.zeros((E, 41)), seed=np.zeros(E, np.int64))
    for e in range(E):
        task = WT(client=client, dt=0.025, neutral_foot_orient=np.array([1, 0, 0, 0]), root_body="torso",
                  lfoot_body="left_foot", rfoot_body="right_foot", head_body="head")
        task._goal_height_ref, task._total_duration = 0.80, 1.1
        task._swing_duration, task._stance_duration = 0.75, 0.35
        base = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), 0.0])
        rq = quat_from_rpy(rng.normal(0, 0.3), rng.normal(0, 0.3), rng.uniform(-np.pi, np.pi)) * rng.choice([-1.0, 1.0])
        client.state = {"xpos_left_foot": base + [rng.normal(0, 0.05), 0.1, 0.03],
                        "xpos_right_foot": base + [rng.normal(0, 0.05), -0.1, 0.02], "xquat_torso": rq}
        captured = {}
        orig = task.generate_step_sequence

        def capture(orig=orig, captured=captured, **kw):
            out = orig(**kw)
            captured["local"] = np.array(out)
            return out
        task.generate_step_sequence = capture
        seed = 2000 + e
        np.random.seed(seed)
        it = int(rng.choice([0, 4000, 9000, 20000]))
        task.reset(iter_count=it)
        env = A3.__new__(A3)
        env._algorithm_type = AlgorithmType.REINFORCEMENT_LEARNING
        env.task, env.actuators, env.base_obs_len = task, list(range(12)), 41
        qpos = np.concatenate([rng.normal(0, 1, 3), quat_from_rpy(rng.normal(0, 0.4), rng.normal(0, 0.4), rng.uniform(-3, 3)),
                               rng.uniform(-1, 1, 18)])
        qvel, alen, avel = rng.normal(0, 1, 24), rng.uniform(-1, 1, 12), rng.normal(0, 2, 12)
        env.interface = types.SimpleNamespace(get_qpos=lambda qpos=qpos: qpos, get_qvel=lambda qvel=qvel: qvel,
                                              get_act_joint_positions=lambda alen=alen: list(alen),
                                              get_act_joint_velocities=lambda avel=avel: list(avel))
        obs = env.get_obs()
        n = len(task.sequence)
        rec["lfoot"][e], rec["rfoot"][e] = client.state["xpos_left_foot"], client.state["xpos_right_foot"]
        rec["root_quat"][e], rec["iter_count"][e], rec["seed"][e] = rq, it, seed
        rec["qpos"][e], rec["qvel"][e], rec["act_len"][e], rec["act_vel"][e] = qpos, qvel, alen, avel
        rec["mode"][e], rec["phase"][e], rec["seq_len"][e] = task.mode.value, task._phase, n
        rec["local_sequence"][e, :n] = captured["local"]
        rec["sequence"][e, :n] = np.array(task.sequence)
        rec["t1"][e], rec["t2"][e], rec["obs"][e] = task.t1, task.t2, obs
    os.chdir(cwd)
    print("a3_reset: modes", np.bincount(rec["mode"]), "phases", np.unique(rec["phase"]))
    save("a3_reset.npz", **rec)


# --------------------------------------------------------------- G7 symmetry
def gen_symmetry():
    base_mir_obs = [0.1, -1, 2, -3, -4, 5, -6, 13, -14, -15, 16, -17, 18, 7, -8, -9, 10, -11, 12,
                    25, -26, -27, 28, -29, 30, 19, -20, -21, 22, -23, 24]
    # the tables exactly as StickFigureA3._initialize_observation_space builds them (:118-129)
    append_obs = [(len(base_mir_obs) + i) for i in range(10)]
    mirrored_obs = np.array(base_mir_obs + append_obs, copy=True).tolist()
    mirrored_acts = [6, -7, -8, 9, -10, 11, 0.1, -1, -2, 3, -4, 5]
    clock_inds = append_obs[0:2]
    # cross-check against the reference source text rather than trusting the transcription
    src = open(f"{REF}/olympic_mujoco/environments/real_humanoid_robots/StickFigureA3.py").read()
    assert "13, -14, -15, 16, -17, 18" in src and "6, -7, -8, 9, -10, 11" in src
    SE = ns.wrappers.SymmetricEnv
    env = SE(lambda: types.SimpleNamespace(base_obs_len=41), mirrored_obs=mirrored_obs,
             mirrored_act=mirrored_acts, clock_inds=clock_inds)
    rng = np.random.default_rng(2)
    obs = torch.tensor(rng.normal(0, 1, (64, 41)), dtype=torch.float32)
    obs[:, 31:33] = torch.tensor(rng.uniform(-1, 1, (64, 2)), dtype=torch.float32)
    act = torch.tensor(rng.normal(0, 1, (64, 12)), dtype=torch.float32)
    save("symmetry.npz", mirrored_obs=np.array(mirrored_obs), mirrored_acts=np.array(mirrored_acts),
         clock_inds=np.array(clock_inds), obs_matrix=env.obs_mirror_matrix.numpy(),
         act_matrix=env.act_mirror_matrix.numpy(), obs=obs.numpy(), act=act.numpy(),
         obs_mirror=env.mirror_observation(obs).numpy(),
         obs_mirror_clock=env.mirror_clock_observation(obs).numpy(),
         act_mirror=env.mirror_action(act).numpy())


# ------------------------------------------------------------------ G8 VAIL D
def gen_vail():
    nw = ns.networks
    torch.manual_seed(0)
    np.random.seed(0)
    # construction exactly as examples/imitation_learning/utils.py:151-161 for UnitreeH1
    enc = nw.FullyConnectedNetwork(input_shape=(32,), output_shape=(128,), n_features=[256],
                                   activations=["relu", "relu"], standardizer=None, squeeze_out=False)
    dec = nw.FullyConnectedNetwork(input_shape=(128,), output_shape=(1,), n_features=[],
                                   activations=["identity"], standardizer=None,
                                   initializers=[nw.NormcInitializer(std=0.1)], squeeze_out=False)
    stand = nw.Standardizer()
    net = nw.VariationalNet(input_shape=(32,), output_shape=(1,), z_size=128, encoder_net=enc,
                            decoder_net=dec, standardizer=stand, use_actions=False,
                            use_next_states=False)
    # spread the logits so that the reward formula is exercised away from 0 as well
    with torch.no_grad():
        dec._linears[0].weight.mul_(40.0)
        net.logvar_out.weight.mul_(30.0)
    rng = np.random.default_rng(8)
    B = 512
    x = (rng.normal(0, 1, (B, 32)) * rng.uniform(0.2, 3.0, 32) + rng.normal(0, 1, 32)).astype(np.float32)
    eps = torch.tensor(rng.normal(0, 1, (B, 128)).astype(np.float32))
    orig = torch.randn_like
    torch.randn_like = lambda t: eps
    try:
        with torch.no_grad():
            d, mu, logvar = net(torch.tensor(x))
    finally:
        torch.randn_like = orig
    G = ns.gail.GAIL
    fake = types.SimpleNamespace(_use_next_state=False,
                                 discrim_output=lambda s, a, apply_mask=True: d.numpy())
    reward = G.make_discrim_reward(fake, x, None, None)
    # extreme logits through the same reference formula
    d_ext = np.array([[-40.0], [-20.0], [-5.0], [0.0], [5.0], [15.0], [17.5], [20.0], [40.0], [90.0]],
                     dtype=np.float32)
    fake2 = types.SimpleNamespace(_use_next_state=False,
                                  discrim_output=lambda s, a, apply_mask=True: d_ext)
    with np.errstate(over="ignore"):
        reward_ext = G.make_discrim_reward(fake2, None, None, None)
    target = np.concatenate([np.zeros((B // 2, 1)), np.ones((B // 2, 1))]).astype(np.float32)
    gl = ns.ilmath.GailDiscriminatorLoss()
    gloss = float(gl(d, torch.tensor(target)))
    vl = ns.ilmath.VDBLoss(info_constraint=0.1, lr_beta=1e-5)
    vloss = float(vl((d, mu, logvar), torch.tensor(target)))
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    save("vail_disc.npz", x=x, eps=eps.numpy(), d=d.numpy(), mu=mu.numpy(), logvar=logvar.numpy(),
         reward=reward, d_ext=d_ext, reward_ext=reward_ext, st_mean=np.array(stand.mean),
         st_std=np.array(stand.std), target=target, gail_loss=gloss, vdb_loss=vloss,
         vdb_beta_after=float(vl._beta),
         enc_w0=sd["encoder_net._linears.0.weight"], enc_b0=sd["encoder_net._linears.0.bias"],
         enc_w1=sd["encoder_net._linears.1.weight"], enc_b1=sd["encoder_net._linears.1.bias"],
         mu_w=sd["mu_out.weight"], mu_b=sd["mu_out.bias"], lv_w=sd["logvar_out.weight"],
         lv_b=sd["logvar_out.bias"], dec_w=sd["decoder_net._linears.0.weight"],
         dec_b=sd["decoder_net._linears.0.bias"])


# -------------------------------------------------------------- G10 PPO update_policy
def gen_ppo_update():
    import importlib
    base = importlib.import_module("rl.policies.base")
    sys.modules["rl.policies"].base = base
    actor_m = importlib.import_module("rl.policies.actor")
    critic_m = importlib.import_module("rl.policies.critic")
    torch.manual_seed(3)
    rng = np.random.default_rng(4)
    policy = actor_m.Gaussian_FF_Actor(41, 12, fixed_std=torch.exp(torch.tensor(-1.5)), bounded=False)
    critic = critic_m.FF_V(41)
    old_policy = actor_m.Gaussian_FF_Actor(41, 12, fixed_std=torch.exp(torch.tensor(-1.5)), bounded=False)
    old_policy.load_state_dict(policy.state_dict())
    with torch.no_grad():                       # make old != new so that ratio != 1
        for prm in policy.parameters():
            prm.add_(0.02 * torch.randn_like(prm))
    B = 64
    obs = torch.tensor(rng.normal(0, 1, (B, 41)), dtype=torch.float32)
    obs[:, 31:33] = torch.tensor(rng.uniform(-1, 1, (B, 2)), dtype=torch.float32)
    act = torch.tensor(rng.normal(0, 0.3, (B, 12)), dtype=torch.float32)
    ret = torch.tensor(rng.normal(1, 1, (B, 1)), dtype=torch.float32)
    adv = torch.tensor(rng.normal(0, 1, (B, 1)), dtype=torch.float32)
    base_mir_obs = [0.1, -1, 2, -3, -4, 5, -6, 13, -14, -15, 16, -17, 18, 7, -8, -9, 10, -11, 12,
                    25, -26, -27, 28, -29, 30, 19, -20, -21, 22, -23, 24]
    mirrored_obs = base_mir_obs + [len(base_mir_obs) + i for i in range(10)]
    mirrored_acts = [6, -7, -8, 9, -10, 11, 0.1, -1, -2, 3, -4, 5]
    sym = ns.wrappers.SymmetricEnv(lambda: types.SimpleNamespace(base_obs_len=41), mirrored_obs=mirrored_obs,
                                   mirrored_act=mirrored_acts, clock_inds=[31, 32])
    fake = types.SimpleNamespace(policy=policy, critic=critic, old_policy=old_policy, clip=0.2, vf_coeff=0.5)
    out = ns.ppo.PPO.update_policy(fake, obs, act, ret, adv, 1, mirror_observation=sym.mirror_clock_observation,
                                   mirror_action=sym.mirror_action)
    names = ("actor_loss", "entropy_penalty", "critic_loss", "approx_kl_div", "mirror_loss", "clip_fraction")
    vals = {n: np.float64(float(v)) for n, v in zip(names, out)}
    out2 = ns.ppo.PPO.update_policy(fake, obs, act, ret, adv, 1)
    vals["mirror_loss_none"] = np.float64(float(out2[4]))
    arrs = {}
    for tag, mod in (("pi", policy), ("old", old_policy), ("vf", critic)):
        for k, v in mod.state_dict().items():
            arrs[f"{tag}.{k}"] = v.numpy()
    save("ppo_update.npz", obs=obs.numpy(), act=act.numpy(), ret=ret.numpy(), adv=adv.numpy(), clip=0.2,
         fixed_std=float(torch.exp(torch.tensor(-1.5))), mirrored_obs=np.array(mirrored_obs),
         mirrored_acts=np.array(mirrored_acts), **vals, **arrs)


# -------------------------------------------------------- G11 Atlas / Talos (table-driven robots)
def gen_il_robot(cls_name, mod, xml, defaults):
    import importlib
    m = importlib.import_module(f"olympic_mujoco.environments.real_humanoid_robots.{mod}")
    cls = getattr(m, cls_name)
    rng = np.random.default_rng(ROBOT_SEEDS[cls_name])
    out = {}
    for tag, (arms, back) in (("default", defaults), ("all_joints", (False, False))):
        env = cls.__new__(cls)
        env._disable_arms, env._disable_back_joint = arms, back
        env._algorithm_type = AlgorithmType.IMITATION_LEARNING
        env._use_foot_forces, env._use_absorbing_states = False, True
        jr, mr, _ = env._get_xml_modifications()
        spec = [e for e in cls._get_observation_specification()
                if e[0] not in ["q_" + j for j in jr] + ["dq_" + j for j in jr]]
        act = [a for a in cls._get_action_specification() if a not in mr]
        env.obs_helper = stubs.FakeObservationHelper(spec)
        joints, motors, nq, nv = walk_mjcf(f"{REF}/olympic_mujoco/environments/data/{xml}", jr, mr)
        qadr = {j[0]: j[1] for j in joints}
        mname = [mm[0] for mm in motors]
        n_pos = sum(1 for e in spec if e[2] == OT.JOINT_POS)
        M = 512
        obs = rng.uniform(-0.9, 0.9, (M, len(spec) - 2))
        obs[:, 0] = rng.uniform(-0.4, 0.2, M)
        fallen = np.zeros(M, bool)
        code = np.zeros(M, np.int32)
        import inspect
        src = inspect.getsource(cls._has_fallen)
        # condition names in the order of the reference's if/elif message chain
        names = [ln.split('"')[1].split(" violated")[0] for ln in src.splitlines() if "error_msg +=" in ln]
        for i in range(M):
            f, msg = env._has_fallen(obs[i], return_err_msg=True)
            fallen[i] = f
            code[i] = 0 if not msg else names.index(msg.split(" violated")[0]) + 1
        out[tag] = dict(keys=np.array([e[0] for e in spec]), qpos_perm=np.array([qadr[e[1]] for e in spec[:n_pos]]),
                        act_to_ctrl=np.array([mname.index(a) for a in act]), nq=nq, n_pos=n_pos,
                        obs=obs, fallen=fallen, code=code, cond_names=np.array(names))
    save(f"{cls_name.lower()}_tables.npz", **{f"{t}.{k}": v for t, d in out.items() for k, v in d.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["tables", "h1", "ppo", "stats", "traj", "contacts", "a3", "sym", "vail", "ppoupd", "robots", "norm", "ppo64", "a3reset"]
    tab = gen_h1_tables() if any(w in which for w in ("tables", "h1", "traj")) else None
    if "h1" in which:
        gen_h1_step(tab)
    if "ppo" in which:
        gen_ppo()
    if "ppo64" in which:
        gen_ppo_f64()
    if "stats" in which:
        gen_running_stats()
    if "norm" in which:
        gen_normalize()
    if "traj" in which:
        gen_trajectory(tab)
    if "contacts" in which:
        gen_contacts()
    if "a3" in which:
        gen_a3_task()
    if "a3reset" in which:
        gen_a3_reset()
    if "sym" in which:
        gen_symmetry()
    if "vail" in which:
        gen_vail()
    if "ppoupd" in which:
        gen_ppo_update()
    if "robots" in which:
        gen_il_robot("Atlas", "atlas", "atlas/atlas.xml", (True, True))
        gen_il_robot("Talos", "talos", "talos/talos.xml", (True, False))
