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
    """transforms3d quat2euler(q)[2] (sxyz) of a (w,x,y,z) quaternion."""
    w, x, y, z = q
    n = w * w + x * x + y * y + z * z
    s = 2.0 / n
    r00 = 1.0 - (y * y + z * z) * s
    r10 = (x * y + w * z) * s
    return np.arctan2(r10, r00)


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

    def step(self, actions):
        """(obs [N,41], total_reward [N], done [N] bool, rewards [N,6])  StickFigureA3.py:187-200."""
        actions = actions.to(torch.float32).contiguous()
        target = self.eng.a3_pd_target(actions)
        inp = self.physics.step(target)
        cr = self.eng.contact_reduce(inp["ncon"], inp["geom1"], inp["geom2"], inp["force6"], inp["cpos_z"],
                                     want_idx=False)
        kin = {k: inp[k] for k in ("qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel", "rf_vel",
                                   "root_pos", "root_quat", "head_pos")}
        kin.update(grf_l=cr["grf_l"], grf_r=cr["grf_r"], min_z=cr["min_z"], n_r=cr["n_r"], n_l=cr["n_l"],
                   bad=cr["bad"])
        o = self.eng.a3_step(kin, self.state, obs_f64=self.obs_f64)
        return o["obs"], o["reward"], o["done"].bool(), o["rew6"]


class ReplayA3Physics:
    """Replays recorded / synthetic per-step input dicts ([K,N,...] device tensors)."""

    def __init__(self, blocks):
        self.blocks, self.k = blocks, 0

    def step(self, target):
        out = {n: v[self.k] for n, v in self.blocks.items()}
        self.k = (self.k + 1) % next(iter(self.blocks.values())).shape[0]
        return out
