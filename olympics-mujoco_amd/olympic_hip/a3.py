"""StickFigureA3 reinforcement-learning mode: host-side episode setup + the vectorised env.

Reference (file:line under the reference tree):
  create_phase_reward                    tasks/rewards.py:270-366   -> clock_lut() (host, once)
  WalkingTask.reset / generate_step_sequence / transform_sequence
                                         tasks/walking_task.py:113-182,321-397 -> WalkingTaskReset
  StickFigureA3.step / get_obs           real_humanoid_robots/StickFigureA3.py:144-200 -> VecA3Env.step
                                         (oly_a3_pd_target -> physics -> oly_contact_reduce -> oly_a3_step)

Episode setup draws random numbers and runs once per episode, so it stays on the host and
writes the per-env state arrays the kernel updates afterwards.  The per-step work of all N
environments is two kernel launches.
"""
import numpy as np
import torch
from scipy.interpolate import PchipInterpolator

from . import _abi


def clock_lut(swing_duration=0.75, stance_duration=0.35, strict_relaxer=0.1, stance_mode="grounded",
              freq=40.0, period=None):
    """[4, period] table (r_frc, r_vel, l_frc, l_vel) of the reference's phase clocks at the
    integer phases the task evaluates them at."""
    s, d = swing_duration, stance_duration
    spans = np.array([[0.0, s], [s, s + d], [s + d, 2 * s + d], [2 * s + d, 2 * (s + d)]]) * freq
    knots = []
    for a, b in spans:
        off = (b - a) * strict_relaxer
        knots += [a + off, b - off]
    x = np.array(knots)
    dbl = {"aerial": (-1.0, 1.0), "zero": (0.0, 0.0)}.get(stance_mode, (1.0, -1.0))   # (frc, vel) in double stance
    # per span: right swing, double stance, left swing, double stance
    r_frc = [-1.0, dbl[0], 1.0, dbl[0]]
    r_vel = [1.0, dbl[1], -1.0, dbl[1]]
    l_frc = [1.0, dbl[0], -1.0, dbl[0]]
    l_vel = [-1.0, dbl[1], 1.0, dbl[1]]
    last_off = (spans[3, 1] - spans[3, 0]) * strict_relaxer
    xs = np.concatenate([x - x[-1] - last_off, x, x + x[-1] + last_off])   # previous, this, next cycle
    if period is None:
        period = int(np.floor(2 * (s + d) * freq))
    ph = np.arange(period)
    rows = []
    for vals in (r_frc, r_vel, l_frc, l_vel):
        y = np.repeat(np.array(vals), 2)
        rows.append(PchipInterpolator(xs, np.tile(y, 3))(ph))
    return np.array(rows)


class WalkingTaskReset:
    """WalkingTask.reset for one environment; `rs` is a numpy RandomState-like source
    (default: the global numpy stream, as in the reference)."""

    def __init__(self, spec, rs=None):
        self.spec = spec
        self.rs = rs if rs is not None else np.random

    def _sequence(self, phase, period, step_size, step_gap, step_height, num_steps, lateral):
        rs = self.rs
        if lateral:
            seq, y = [], 0.0
            c = rs.choice([-1, 1])
            for i in range(1, num_steps):
                y = y + step_size if i % 2 else y - (2 / 3) * step_size
                seq.append(np.array([0, c * y, 0, 0]))
            return seq
        if phase == 0.5 * period:
            first, y = np.array([0, -1 * rs.uniform(0.095, 0.105), 0, 0]), -step_gap
        else:
            first, y = np.array([0, 1 * rs.uniform(0.095, 0.105), 0, 0]), step_gap
        seq = [first]
        x = z = 0
        c = rs.randint(2, 4)
        for i in range(1, num_steps):
            x += step_size
            y *= -1
            if i > c:
                z += step_height
            seq.append(np.array([x, y, z, 0]))
        return seq

    def __call__(self, lfoot_pos, rfoot_pos, root_yaw, iter_count=0):
        """Returns dict(mode, phase, sequence [<=20,4] world frame, seq_len, t1, t2)."""
        rs, sp = self.rs, self.spec
        period = np.floor(2 * sp.total_duration * (1 / sp.control_dt))
        phase = int(rs.choice([0, period / 2]))
        mode = rs.choice([_abi.MODE_STANDING, _abi.MODE_BACKWARD, _abi.MODE_LATERAL, _abi.MODE_FORWARD],
                         p=[0.2, 0, 0, 0.8])
        d = dict(step_size=0.3, step_gap=0.15, step_height=0, num_steps=20, lateral=False)
        if mode == _abi.MODE_STANDING:
            d["num_steps"] = 1
        elif mode == _abi.MODE_BACKWARD:
            d["step_size"] = -0.1
        elif mode == _abi.MODE_LATERAL:
            d["step_size"], d["lateral"] = 0.4, True
        else:
            h = np.clip((iter_count - 3000) / 8000, 0, 1) * 0.1
            d["step_height"] = rs.choice([-h, h])
        seq = self._sequence(phase, period, **d)
        mid = (np.asarray(lfoot_pos) + np.asarray(rfoot_pos)) / 2
        cy, sy = np.cos(root_yaw), np.sin(root_yaw)
        world = [np.array([mid[0] + x * cy - y * sy, mid[1] + x * sy + y * cy, z, root_yaw + th])
                 for x, y, z, th in seq]
        t1, t2 = 0, 1                                        # reset sets 0,0 then update_target_steps
        if t2 == len(world):
            t2 = len(world) - 1
        return dict(mode=int(mode), phase=phase, sequence=np.array(world), seq_len=len(world), t1=t1, t2=t2)


def yaw_of_quat(q):
    """transforms3d quat2euler(q)[2] (sxyz) of a (w,x,y,z) quaternion = mat2euler(quat2mat(q))[2]:
    the same expressions, in the same order, as the device-side transform_sequence (k10_vec_step.hip)."""
    w, x, y, z = (float(v) for v in q)
    n = w * w + x * x + y * y + z * z
    if n < np.finfo(np.float64).eps:
        return 0.0
    s = 2.0 / n
    X, Y, Z = x * s, y * s, z * s
    r00 = 1.0 - (y * Y + z * Z)
    r10 = x * Y + w * Z
    if np.sqrt(r00 * r00 + r10 * r10) > 4.0 * np.finfo(np.float64).eps:
        return float(np.arctan2(r10, r00))
    return 0.0


class VecA3Env:
    """N StickFigureA3 environments in RL mode.  `physics` supplies, per step, the dict of
    device tensors named in oly_a3_inputs (minus the K3 outputs) plus the contact slots."""

    def __init__(self, spec, num_envs, engine, physics, geom_bodyid, floor_body, rfoot_body, lfoot_body,
                 lut=None, obs_f64=False, rs=None):
        self.spec, self.num_envs, self.eng, self.physics = spec, int(num_envs), engine, physics
        self.lut = clock_lut(spec.swing_duration, spec.stance_duration, 0.1, "grounded", 1 / spec.control_dt,
                             spec.period) if lut is None else lut
        engine.a3_configure(spec, self.lut)
        engine.contact_configure(geom_bodyid, floor_body, rfoot_body, lfoot_body)
        self.obs_f64 = obs_f64
        self._reset_one = WalkingTaskReset(spec, rs)
        N, dev = self.num_envs, engine.device
        z = lambda dt, *shape: torch.zeros((N,) + shape, dtype=dt, device=dev)
        self.state = dict(phase=z(torch.int32), t1=z(torch.int32), t2=z(torch.int32),
                          reached_frames=z(torch.int32), target_reached=z(torch.uint8), mode=z(torch.int32),
                          seq_len=z(torch.int32), sequence=z(torch.float64, _abi.OLY_MAX_SEQ, 4),
                          goal=z(torch.float64, 8))
        self.observation_space = np.zeros(spec.n_obs)
        self.action_space = np.zeros(spec.nu)
        self.base_obs_len = spec.n_obs
        self.iteration_count = 0

    def reset_task(self, env_ids, lfoot_pos, rfoot_pos, root_quat):
        """Host episode setup for the listed envs (arrays indexed like env_ids)."""
        host = {k: [] for k in ("mode", "phase", "seq_len", "t1", "t2")}
        seqs = np.zeros((len(env_ids), _abi.OLY_MAX_SEQ, 4))
        for i in range(len(env_ids)):
            r = self._reset_one(lfoot_pos[i], rfoot_pos[i], yaw_of_quat(root_quat[i]), self.iteration_count)
            for k in host:
                host[k].append(r[k])
            seqs[i, :r["seq_len"]] = r["sequence"]
        idx = torch.as_tensor(np.asarray(env_ids, dtype=np.int64), device=self.eng.device)
        for k in host:
            self.state[k][idx] = torch.as_tensor(np.asarray(host[k], dtype=np.int32), device=self.eng.device)
        self.state["sequence"][idx] = torch.as_tensor(seqs, device=self.eng.device)
        self.state["reached_frames"][idx] = 0
        self.state["target_reached"][idx] = 0
        self.state["goal"][idx] = 0.0

    def state_dict(self):
        """Task state carried between steps (phase counters, target indices, sequences, goals)."""
        return dict(iteration_count=self.iteration_count, **{k: v.clone() for k, v in self.state.items()})

    def load_state_dict(self, d):
        self.iteration_count = d["iteration_count"]
        for k, v in self.state.items():
            v.copy_(d[k])

    @property
    def has_device_physics(self):
        """True when the physics readback is a set of device-resident [K,N,...] blocks: the whole
        rollout loop can then stay on the device (vecstep.A3DeviceRollout)."""
        return isinstance(self.physics, ReplayA3Physics)

    def device_rollout(self, policy, critic, T, max_traj_len, deterministic=False, anneal=1.0, graph=True, **kw):
        """PPO.sample for all N environments with one fused launch per vec step (K10)."""
        if getattr(self, "_dev_rollout", None) is None:
            from .vecstep import A3DeviceRollout
            self._dev_rollout = A3DeviceRollout(self, self.physics.blocks, rs=self._reset_one.rs)
        return self._dev_rollout.rollout(policy, critic, T, max_traj_len, deterministic, anneal, graph, **kw)

    def step(self, actions):
        """(obs [N,41], total_reward [N], done [N] bool, rewards [N,6])  StickFigureA3.py:187-200."""
        actions = actions.to(torch.float32).contiguous()
        target = self.eng.a3_pd_target(actions)
        o = self._evaluate(self.physics.step(target))
        return o["obs"], o["reward"], o["done"].bool(), o["rew6"]

    def _evaluate(self, inp):
        """K3 + K2 on one physics readback (task.step -> calc_reward -> done -> get_obs)."""
        cr = self.eng.contact_reduce(inp["ncon"], inp["geom1"], inp["geom2"], inp["force6"], inp["cpos_z"],
                                     want_idx=False)
        kin = {k: inp[k] for k in ("qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel", "rf_vel",
                                   "root_pos", "root_quat", "head_pos")}
        kin.update(grf_l=cr["grf_l"], grf_r=cr["grf_r"], min_z=cr["min_z"], n_r=cr["n_r"], n_l=cr["n_l"],
                   bad=cr["bad"])
        return self.eng.a3_step(kin, self.state, obs_f64=self.obs_f64)


class ReplayA3Physics:
    """Replays recorded / synthetic per-step input dicts ([K,N,...] device tensors)."""

    def __init__(self, blocks, mass=41.5):
        self.blocks, self.k, self.mass = blocks, 0, mass
        self._cur = {n: v[0] for n, v in blocks.items()}

    def step(self, target):
        self._cur = {n: v[self.k] for n, v in self.blocks.items()}
        self.k = (self.k + 1) % next(iter(self.blocks.values())).shape[0]
        return self._cur

    def readback(self):
        return self._cur

    def set_state(self, env_ids, qpos, qvel):
        """Replay has no dynamics: reports the recorded foot / root poses of the listed envs."""
        idx = torch.as_tensor(np.asarray(env_ids, dtype=np.int64), device=self._cur["lf_pos"].device)
        g = lambda k: self._cur[k][idx].cpu().numpy()
        return dict(lfoot_pos=g("lf_pos"), rfoot_pos=g("rf_pos"), root_quat=g("root_quat"))


class AlgorithmType:
    """olympic_mujoco/enums/enums.py"""
    REINFORCEMENT_LEARNING = "reinforcement_learning"
    IMITATION_LEARNING = "imitation_learning"


class StickFigureA3:
    """Reinforcement-learning StickFigureA3 with the reference's interface
    (real_humanoid_robots/StickFigureA3.py:60-260; used as `partial(StickFigureA3,
    algorithm_type=REINFORCEMENT_LEARNING)` by examples/reinforcement_learning_ppo/a3):

        env.robot.{mirrored_obs, mirrored_acts, clock_inds, iteration_count, actuators}
        env.observation_space / action_space (np.zeros(41) / np.zeros(12)), env.base_obs_len
        env.reset() / reset_model() -> obs;  env.step(a) -> (obs, total_reward, done, rewards dict)

    num_envs = 1 gives exactly that scalar API (numpy in / out); larger values step N envs
    (`.vec` is the VecA3Env).  The physics object owns MuJoCo:
        physics.set_state(env_ids, qpos [n,25], qvel [n,24]) -> dict(lfoot_pos, rfoot_pos, root_quat) [n,..] (numpy)
        physics.step(target [N,12]) -> readback dict (oly_a3_inputs fields + contact slots), device tensors
        physics.readback() -> the same dict for the current state
        physics.mass  (mj_getTotalmass)
    """

    REWARD_NAMES = ("foot_frc_score", "foot_vel_score", "orient_cost", "height_error", "step_reward",
                    "upper_body_reward")                                   # walking_task.py:96-103

    def __init__(self, algorithm_type=AlgorithmType.REINFORCEMENT_LEARNING, num_envs=1, physics=None, mass=None,
                 device=0, engine=None, rs=None, **unused):
        from types import SimpleNamespace
        from .engine import Engine
        from .robot_data import ROBOTS
        from .specs import A3Spec
        name = getattr(algorithm_type, "name", str(algorithm_type)).lower()
        if "reinforcement" not in name:
            raise NotImplementedError("StickFigureA3: only the reinforcement-learning mode is on the accelerated path")
        if physics is None:
            raise ValueError("StickFigureA3 needs a physics object (MuJoCo stays on the host)")
        d = ROBOTS["StickFigureA3"]
        spec = A3Spec(mass=float(mass if mass is not None else physics.mass))
        self.spec, self.physics = spec, physics
        self.vec = VecA3Env(spec, num_envs, engine or Engine(device), physics, np.asarray(d["geom_bodyid"], np.int32),
                            d["bodies"]["world"], d["bodies"]["right_foot"], d["bodies"]["left_foot"], obs_f64=True, rs=rs)
        self.rs = rs if rs is not None else np.random
        self.robot = SimpleNamespace(mirrored_obs=list(spec.mirrored_obs), mirrored_acts=list(spec.mirrored_acts),
                                     clock_inds=list(spec.clock_inds), iteration_count=np.inf,      # robot.py:55
                                     actuators=list(range(spec.nu)))
        self.observation_space = np.zeros(spec.n_obs)
        self.action_space = np.zeros(spec.nu)
        self.base_obs_len = spec.n_obs
        init = np.zeros(spec.nq)
        init[:3], init[3:7] = [0, 0, 0.81], [1, 0, 0, 0]                   # robot.py:60-86
        init[7:] = np.asarray(spec.half_sitting_pose_deg) * np.pi / 180.0
        self._init_qpos, self._init_qvel = init, np.zeros(spec.nv)

    @property
    def num_envs(self):
        return self.vec.num_envs

    @property
    def eng(self):
        return self.vec.eng

    @property
    def device(self):
        return self.vec.eng.device

    @property
    def has_device_physics(self):
        return self.vec.has_device_physics

    def device_rollout(self, *args, **kw):
        """PPO.sample on the device (vecstep.A3DeviceRollout) when the physics readback lives there."""
        self.vec.iteration_count = self.robot.iteration_count if np.isfinite(self.robot.iteration_count) else 10 ** 9
        return self.vec.device_rollout(*args, **kw)

    def _draw_init_state(self):
        """reset_model's draws, in the reference's order (StickFigureA3.py:213-228)."""
        rs, sp, c = self.rs, self.spec, 0.02
        qpos = self._init_qpos + rs.uniform(low=-c, high=c, size=sp.nq)
        qvel = self._init_qvel + rs.uniform(low=-c, high=c, size=sp.nv)
        qpos[0], qpos[1], qpos[2] = rs.uniform(-1, 1), rs.uniform(-1, 1), 1.34
        pitch, yaw = rs.uniform(-5, 5) * np.pi / 180, rs.uniform(-np.pi, np.pi)
        cj, sj, ck, sk = np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw / 2), np.sin(yaw / 2)
        qpos[3:7] = [cj * ck, -sj * sk, sj * ck, cj * sk]                  # euler2quat(0, pitch, yaw), sxyz
        return qpos, qvel

    def reset_model(self, env_ids=None):
        """Random initial state + WalkingTask.reset for the listed envs.  For one env the draws come
        in the reference's order (state, then task); for several the state draws of all envs come
        first (the reference runs one env per process, so there is no cross-env order to keep)."""
        ids = list(range(self.num_envs)) if env_ids is None else list(env_ids)
        if not ids:
            return self._get_obs()
        states = [self._draw_init_state() for _ in ids]
        kin = self.physics.set_state(ids, np.stack([q for q, _ in states]), np.stack([v for _, v in states]))
        self.vec.iteration_count = self.robot.iteration_count if np.isfinite(self.robot.iteration_count) else 10 ** 9
        self.vec.reset_task(ids, kin["lfoot_pos"], kin["rfoot_pos"], kin["root_quat"])
        return self._get_obs()

    def _get_obs(self):
        """get_obs of the current state without advancing the task: phase is stepped back by one
        around the kernel's phase increment (the kernel evaluates WalkingTask.step + get_obs)."""
        inp = self.physics.readback()
        keep = {k: v.clone() for k, v in self.vec.state.items()}
        self.vec.state["phase"].sub_(1)
        obs = self.vec._evaluate(inp)["obs"]
        for k, v in keep.items():
            self.vec.state[k].copy_(v)
        # get_obs reads the task's STORED goal steps (StickFigureA3.py:150-154): after a reset those are the
        # zeros WalkingTask.reset wrote (walking_task.py:325-328), not what update_goal_steps would derive
        obs[:, -8:] = keep["goal"].to(obs.dtype)
        return obs if self.num_envs > 1 else obs[0].cpu().numpy()

    def reset(self, env_mask=None):
        if env_mask is None:
            return self.reset_model()
        ids = torch.nonzero(env_mask).flatten().tolist()
        return self.reset_model(ids)

    def set_algorithm_type(self, algorithm_type):            # loco_env_base.py:203-204
        self._algorithm_type = algorithm_type

    def render(self, record=False):
        raise NotImplementedError("rendering / recording is out of scope of the accelerated path")

    def stop(self):
        pass

    def close(self):
        pass

    def step(self, a):
        if self.num_envs > 1:
            return self.vec.step(a)
        act = torch.as_tensor(np.asarray(a, dtype=np.float32)[None], device=self.eng.device)
        obs, rew, done, rew6 = self.vec.step(act)
        r6 = rew6[0].cpu().numpy()
        return (obs[0].cpu().numpy(), float(sum(float(x) for x in r6)), bool(done[0]),
                {n: float(v) for n, v in zip(self.REWARD_NAMES, r6)})
