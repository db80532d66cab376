"""Environment facade with the reference's API on top of the HIP engine.

Reference surface mirrored here (file:line under the reference tree):
  LocoEnvBase.make / register / get_all_task_names   loco_env_base.py:1337-1375 (+ mushroom Environment.make)
  ValidTaskConf                                       loco_env_base.py:1381-1455
  UnitreeH1.generate / valid_task_confs               real_humanoid_robots/UnitreeH1.py:34-36,205-242
  step (mushroom MuJoCo.step order)        -> oly_il_step
  reset / setup / set_sim_state                       loco_env_base.py:568-705
  play_trajectory_from_velocity                       loco_env_base.py:444-560
  create_dataset / get_kinematic_obs_mask / get_obs_idx   :926-968, :870-886, :1195-1205

Physics (mj_step / mj_forward) is a HOST concern and pluggable: `KinematicPhysics` keeps the
state it is given (what mj_forward does to qpos/qvel) and `ReplayPhysics` replays synthetic
batches; a MuJoCo-backed batcher plugs into the same three methods.  Everything after
physics - observation build, has-fallen, reward, action scaling, trajectory lookup - runs in
the HIP kernels for all N environments at once; `UnitreeH1` is the N = 1 view with the
reference's scalar return types.
"""
from copy import deepcopy
from itertools import product
from types import SimpleNamespace

import numpy as np
import torch

from . import specs as _specs
from ._ffi import OlyError
from .trajectory import Trajectory, synthetic_h1_trajectory_files


class ValidTaskConf:
    """Holds all valid configurations of an environment (loco_env_base.py:1381-1455)."""

    def __init__(self, tasks=None, modes=None, data_types=None, non_combinable=None):
        self.tasks, self.modes, self.data_types, self.non_combinable = tasks, modes, data_types, non_combinable
        for nc in non_combinable or []:
            assert len(nc) == 3

    def get_all(self):
        return deepcopy(self.tasks), deepcopy(self.modes), deepcopy(self.data_types), deepcopy(self.non_combinable)

    def get_all_combinations(self):
        confs = []
        for t, m, dt in product(self.tasks or [None], self.modes or [None], self.data_types or [None]):
            conf = {}
            if t is not None:
                conf["task"] = t
            if m is not None:
                conf["mode"] = m
            if dt is not None:
                conf["data_type"] = dt
            if self.non_combinable is not None:
                # the reference appends once per non-matching rule (:1443-1451); kept as is
                for bt, bm, bdt in self.non_combinable:
                    if not ((t == bt or bt is None) and (m == bm or bm is None) and (dt == bdt or bdt is None)):
                        confs.append(conf)
            else:
                confs.append(conf)
        return confs


def check_validity_task_mode_dataset(env_name, task=None, mode=None, dataset_type=None, valid_tasks=None,
                                     valid_modes=None, valid_dataset_types=None, non_combineable=None):
    """Same acceptance rules and exception type as olympic_mujoco/utils/checks.py:3-76."""
    if task is not None and task not in valid_tasks:
        raise ValueError(f"Task \"{task}\" does not exit in the environment {env_name}. Please, choose from "
                         f"{valid_tasks}.")
    if mode is not None and mode not in valid_modes:
        raise ValueError(f"Mode \"{mode}\" does not exit in the environment {env_name}. Please, choose from "
                         f"{valid_modes}.")
    if dataset_type is not None and dataset_type not in valid_dataset_types:
        raise ValueError(f"Dataset type \"{dataset_type}\" does not exit in the environment {env_name}. "
                         f"Please, choose from {valid_dataset_types}.")
    for bt, bm, bdt in non_combineable or []:
        if (task == bt or bt is None) and (mode == bm or bm is None) and (dataset_type == bdt or bdt is None):
            raise ValueError(f"Task \"{task}\", mode \"{mode}\" and dataset type \"{dataset_type}\" are not "
                             f"combineable for the environment {env_name}.")


# ------------------------------------------------------------------------------ physics
class KinematicPhysics:
    """Holds qpos/qvel as written (mj_forward leaves them unchanged); `step` integrates
    nothing.  Used by play_trajectory_from_velocity and as the default stand-in."""

    needs_ctrl = False   # a MuJoCo-backed batcher sets this and consumes ctrl in step()

    def __init__(self, spec, num_envs, device):
        self.qpos = torch.zeros((num_envs, spec.nq), dtype=torch.float64, device=device)
        self.qvel = torch.zeros((num_envs, spec.nv), dtype=torch.float64, device=device)

    def reset(self, env_mask=None):
        if env_mask is None:
            self.qpos.zero_()
            self.qvel.zero_()
        else:
            self.qpos[env_mask] = 0.0
            self.qvel[env_mask] = 0.0

    def set_state(self, qpos, qvel, env_mask=None):
        if env_mask is None:
            self.qpos.copy_(qpos)
            self.qvel.copy_(qvel)
        else:
            self.qpos[env_mask] = qpos[env_mask]
            self.qvel[env_mask] = qvel[env_mask]

    def step(self, ctrl):
        return self.qpos, self.qvel

    def substep_contacts(self):
        """use_foot_forces: dict(ncon [W,N], geom1/geom2 [W,N,C], force6 [W,N,C,6]) of the W
        intermediate steps of the control step just taken, or None (no contacts: zeros)."""
        return None


class ReplayPhysics(KinematicPhysics):
    """Replays pre-generated [T,N,nq]/[T,N,nv] device blocks, one row per step();
    `contacts` = dict of [T,W,N,...] blocks replays the per-substep contact readback too."""

    def __init__(self, spec, qpos_block, qvel_block, contacts=None):
        super().__init__(spec, qpos_block.shape[1], qpos_block.device)
        self.qb, self.vb, self.t = qpos_block, qvel_block, 0
        self.cb, self._last = contacts, None

    def step(self, ctrl):
        self.qpos, self.qvel = self.qb[self.t], self.vb[self.t]
        self._last = self.t
        self.t = (self.t + 1) % self.qb.shape[0]
        return self.qpos, self.qvel

    def substep_contacts(self):
        if self.cb is None or self._last is None:
            return None
        return {k: v[self._last] for k, v in self.cb.items()}


# ------------------------------------------------------------------------------ vec env
class VecLocoEnv:
    """N imitation-learning environments stepped together on one GPU.

    step(actions [N,n_act]) -> (obs [N,n_obs], reward [N], absorbing [N] bool, info) with
    per-env semantics equal to the reference's single-env step."""

    def __init__(self, spec, num_envs, device=0, engine=None, trajectory=None, physics=None,
                 random_start=True, init_step_no=None, obs_f64=False, seed=None):
        from .engine import Engine
        self.spec = spec
        self.num_envs = int(num_envs)
        self.eng = engine or Engine(device)
        self.eng.il_configure(spec)
        if spec.n_grf:
            self.eng.grf_configure(spec.geom_group, spec.grf_pairs)
        self.device = self.eng.device
        self.obs_f64 = obs_f64
        self.physics = physics or KinematicPhysics(spec, self.num_envs, self.device)
        self.trajectories = trajectory
        self._random_start, self._init_step_no = random_start, init_step_no
        self._rng = np.random.default_rng(seed)
        N = self.num_envs
        self._prev = torch.zeros(N, dtype=torch.float64, device=self.device)
        self._obs = None
        self.episode_steps = torch.zeros(N, dtype=torch.int32, device=self.device)
        if trajectory is not None:
            self.eng.traj_upload(trajectory.table)
            self._cur_traj = torch.zeros(N, dtype=torch.int32, device=self.device)
            self._cur_step = torch.zeros(N, dtype=torch.int32, device=self.device)
            self._origin = torch.zeros((N, 2), dtype=torch.float64, device=self.device)
            self._sample = torch.zeros((N, len(trajectory.keys)), dtype=torch.float64, device=self.device)
        self._qadr = torch.as_tensor(spec.qpos_adr.astype(np.int64), device=self.device)
        self._vadr = torch.as_tensor(spec.qvel_adr.astype(np.int64), device=self.device)
        self.info = SimpleNamespace(
            observation_space=SimpleNamespace(shape=(spec.n_obs,), low=self._obs_low(), high=self._obs_high()),
            action_space=SimpleNamespace(shape=(spec.n_act,), low=-np.ones(spec.n_act), high=np.ones(spec.n_act)),
            gamma=spec.gamma, horizon=spec.horizon)

    # ----- spaces (loco_env_base.py:715-735: x,y dropped)
    def _obs_low(self):
        lo = np.concatenate([self.spec.joint_lo, -np.inf * np.ones(self.spec.n_vel)])
        lo[:6] = -np.inf                       # pelvis joints are limited="false" (h1.xml:88-93)
        return np.concatenate([lo[self.spec.n_drop:], -np.inf * np.ones(self.spec.n_grf)])   # grf: :727-728

    def _obs_high(self):
        hi = np.concatenate([self.spec.joint_hi, np.inf * np.ones(self.spec.n_vel)])
        hi[:6] = np.inf
        return np.concatenate([hi[self.spec.n_drop:], np.inf * np.ones(self.spec.n_grf)])

    @property
    def dt(self):
        return self.spec.dt

    # ----- state setting (loco_env_base.py:659-684): spec-ordered sample -> qpos/qvel
    def set_sim_state(self, sample, env_mask=None):
        sp = self.spec
        if sample.shape[-1] < sp.n_pos + sp.n_vel:
            raise AssertionError("sample shorter than the observation spec")
        qpos = torch.zeros((self.num_envs, sp.nq), dtype=torch.float64, device=self.device)
        qvel = torch.zeros((self.num_envs, sp.nv), dtype=torch.float64, device=self.device)
        qpos[:, self._qadr] = sample[:, :sp.n_pos]
        qvel[:, self._vadr] = sample[:, sp.n_pos:sp.n_pos + sp.n_vel]
        self.physics.set_state(qpos, qvel, env_mask)

    def _observe(self, action=None, fresh=False):
        """Post-physics half of step(): one oly_il_step over [1,N]."""
        o = self.eng.il_step(self.physics.qpos.unsqueeze(0).contiguous(), self.physics.qvel.unsqueeze(0).contiguous(),
                             None if action is None else action.unsqueeze(0).contiguous(), self._prev,
                             grf_mean=self._grf_mean(fresh), obs_f64=self.obs_f64, ctrl_f64=False)
        return o

    def _grf_mean(self, fresh):
        """mean_grf.mean of the control step (loco_env_base.py:1072-1084); zeros right after a
        reset (mean_grf.reset(), :584) and when the physics reports no contacts."""
        if not self.spec.n_grf:
            return None
        c = None if fresh else self.physics.substep_contacts()
        if c is None:
            return torch.zeros((1, self.num_envs, self.spec.n_grf), dtype=torch.float64, device=self.device)
        o = self.eng.il_ground_forces(c["ncon"], c["geom1"], c["geom2"], c["force6"])
        # sticky per-environment overflow flags, no host round trip per step: read in reset() / raise_if_contact_overflow()
        self._grf_overflow = o["overflow"] if getattr(self, "_grf_overflow", None) is None else torch.maximum(self._grf_overflow, o["overflow"])
        return o["mean"].unsqueeze(0)

    def raise_if_contact_overflow(self):
        """One device-to-host read of the sticky overflow flags of every il_ground_forces call since the last check: an
        environment whose contact count exceeded the staged slots in a substep where a sensor pair found none among them
        (the foot-force columns of that step cannot match the reference, UnitreeH1.py:113-123)."""
        over = getattr(self, "_grf_overflow", None)
        if over is not None and self.num_envs and bool(over.any().item()):
            bad = torch.nonzero(over).flatten()[:8].tolist()
            self._grf_overflow = None
            raise OlyError(f"il_ground_forces: more contacts than the staged slots and a sensor pair without a contact "
                           f"among them (or a negative count) in environments {bad} since the last check: stage more slots")
        self._grf_overflow = None

    # ----- reset (loco_env_base.py:568-657)
    def reset(self, env_mask=None, obs=None):
        N = self.num_envs
        if env_mask is None:
            self.raise_if_contact_overflow()          # a full reset is a host-synchronous point of every rollout loop
        if obs is not None:
            full = torch.cat([torch.zeros((N, self.spec.n_drop), dtype=torch.float64, device=self.device),
                              torch.as_tensor(obs, dtype=torch.float64, device=self.device).reshape(N, -1)], dim=1)
            self.set_sim_state(full[:, :self.spec.n_pos + self.spec.n_vel], env_mask)
        elif self.trajectories is None:
            if self._random_start:
                raise ValueError("Random start not possible without trajectory data.")
            if self._init_step_no is not None:
                raise ValueError("Setting an initial step is not possible without trajectory data.")
            self.physics.reset(env_mask)
        else:
            if self._init_step_no is not None and self._random_start:
                raise ValueError("Either use a random start or set an initial step, not both.")
            L, J = self.trajectories.trajectory_length, self.trajectories.number_of_trajectories
            if self._random_start:
                tn, st = self._rng.integers(0, J, N), self._rng.integers(0, L, N)
            elif self._init_step_no:
                assert self._init_step_no <= L * J
                tn = np.full(N, int(self._init_step_no / L))
                st = np.full(N, int(self._init_step_no % L))
            else:
                tn, st = self._rng.integers(0, J, N), np.zeros(N, dtype=np.int64)
            tn_d = torch.as_tensor(tn.astype(np.int32), device=self.device)
            st_d = torch.as_tensor(st.astype(np.int32), device=self.device)
            if env_mask is None:
                self.eng.traj_reset(tn_d, st_d, self._cur_traj, self._cur_step, self._origin, self._sample)
            else:
                ct, cs, org, smp = self.eng.traj_reset(tn_d, st_d)
                m = env_mask
                self._cur_traj[m], self._cur_step[m] = ct[m], cs[m]
                self._origin[m], self._sample[m] = org[m], smp[m]
            self.set_sim_state(self._sample, env_mask)
        # self._obs = _create_observation(_build_obs(data)); the reward's "previous obs" is this one
        prev_keep = self._prev.clone()
        o = self._observe(fresh=True)
        new_prev = o["prev"]
        if env_mask is not None:
            new_prev = torch.where(env_mask, new_prev, prev_keep)
            self.episode_steps[env_mask] = 0
        else:
            self.episode_steps.zero_()
        self._prev = new_prev.contiguous()
        self._obs = o["obs"][0]
        return self._obs

    # ----- step
    def step(self, actions):
        actions = torch.as_tensor(actions, device=self.device).to(torch.float32).reshape(self.num_envs, self.spec.n_act).contiguous()
        ctrl = None
        if getattr(self.physics, "needs_ctrl", False):      # un-normalised, clamped, actuator-ordered
            pre = self.eng.il_step(self.physics.qpos.unsqueeze(0).contiguous(),
                                   self.physics.qvel.unsqueeze(0).contiguous(), actions.unsqueeze(0),
                                   self._prev.clone(), grf_mean=self._grf_mean(True), want_fall_code=False)
            ctrl = pre["ctrl"][0]
        self.physics.step(ctrl)
        o = self._observe(actions)
        self._prev = o["prev"]
        self._obs = o["obs"][0]
        self.episode_steps += 1
        absorbing = o["absorbing"][0].bool()
        last = absorbing | (self.episode_steps >= self.spec.horizon)
        info = dict(fall_code=o["fall_code"][0], last=last, ctrl=o["ctrl"][0] if o["ctrl"] is not None else None)
        return self._obs, o["reward"][0], absorbing, info

    # ----- block regime: T pre-computed physics states in ONE launch (what bench.py times)
    def evaluate_block(self, qpos, qvel, actions=None, prev=None):
        """qpos/qvel [T,N,*] f64 (+ actions [T,N,n_act] f32) -> dict(obs, reward, absorbing,
        fall_code, ctrl, prev) with the per-env reward chain carried through the block; the
        env's own carried state is not touched."""
        if prev is None:
            prev = self._prev.clone()
        return self.eng.il_step(qpos, qvel, actions, prev, obs_f64=self.obs_f64)

    # ----- trajectory replay (loco_env_base.py:444-560)
    def play_trajectory_from_velocity(self, n_steps, record=True):
        """Advance all N envs along their reference trajectories by explicit Euler on the
        trajectory velocities; returns (obs [n_steps,N,n_obs], fallen [n_steps,N])."""
        if self.trajectories is None:
            raise AssertionError("no trajectory loaded")
        sp = self.spec
        self.reset()
        obs_rec, fallen_rec = [], []
        curr_qpos = self._sample[:, :sp.n_pos].clone()
        for _ in range(n_steps):
            self.eng.traj_euler(sp.n_pos, self.dt, curr_qpos, self._sample)
            self.set_sim_state(self._sample)
            curr_qpos = self.physics.qpos[:, self._qadr].contiguous()      # _get_joint_pos after mj_forward
            at_end = self.eng.traj_next(self._cur_traj, self._cur_step, self._origin, self._sample)
            if bool(at_end.any()):
                m = at_end.bool()
                self.reset(env_mask=m)
                curr_qpos = torch.where(m.unsqueeze(1), self._sample[:, :sp.n_pos], curr_qpos)
            # obs of the SAMPLE (loco_env_base.py:539): route the sample through the same kernel
            keep_q, keep_v = self.physics.qpos.clone(), self.physics.qvel.clone()
            self.set_sim_state(self._sample)
            o = self._observe()
            self.physics.set_state(keep_q, keep_v)
            if record:
                obs_rec.append(o["obs"][0].clone())
                fallen_rec.append(o["fall_code"][0] > 0)
        if record:
            return torch.stack(obs_rec), torch.stack(fallen_rec)
        return None

    def create_dataset(self, ignore_keys=None):
        if self.trajectories is None:
            raise ValueError("No trajectory was passed to the environment. To create a dataset pass a trajectory first.")
        ds = self.trajectories.create_dataset(ignore_keys=ignore_keys)
        # dataset states must not be terminal (loco_env_base.py:950-957): checked on the GPU
        st = torch.as_tensor(ds["states"], dtype=torch.float64, device=self.device)
        full = torch.cat([torch.zeros((len(st), self.spec.n_drop), dtype=torch.float64, device=self.device), st], 1)
        qpos = torch.zeros((1, len(st), self.spec.nq), dtype=torch.float64, device=self.device)
        qvel = torch.zeros((1, len(st), self.spec.nv), dtype=torch.float64, device=self.device)
        qpos[0][:, self._qadr] = full[:, :self.spec.n_pos]
        qvel[0][:, self._vadr] = full[:, self.spec.n_pos:]
        o = self.eng.il_step(qpos, qvel, None, torch.zeros(len(st), dtype=torch.float64, device=self.device))
        bad = (o["fall_code"][0] > 0)
        if bool(bad.any()):
            k = int(o["fall_code"][0][bad][0].item()) - 1
            raise ValueError("Some of the states in the created dataset are terminal states. This should not "
                             "happen.\n\nViolations:\n" + self.spec.fall_names[k] + " violated.\n")
        return ds

    # ----- checkpoint / resume: everything an iteration carries besides the policy
    def state_dict(self):
        d = dict(prev=self._prev.clone(), episode_steps=self.episode_steps.clone(),
                 qpos=self.physics.qpos.clone(), qvel=self.physics.qvel.clone(),
                 rng=self._rng.bit_generator.state, obs=None if self._obs is None else self._obs.clone())
        if self.trajectories is not None:
            d.update(cur_traj=self._cur_traj.clone(), cur_step=self._cur_step.clone(), origin=self._origin.clone(),
                     sample=self._sample.clone())
        return d

    def load_state_dict(self, d):
        self._prev = d["prev"].to(self.device).clone()
        self.episode_steps.copy_(d["episode_steps"])
        self.physics.set_state(d["qpos"].to(self.device), d["qvel"].to(self.device))
        self._rng.bit_generator.state = d["rng"]
        self._obs = None if d["obs"] is None else d["obs"].to(self.device).clone()
        if self.trajectories is not None:
            for k, t in (("cur_traj", self._cur_traj), ("cur_step", self._cur_step), ("origin", self._origin),
                         ("sample", self._sample)):
                t.copy_(d[k])

    def get_kinematic_obs_mask(self):
        return np.arange(self.spec.n_pos + self.spec.n_vel - 2)       # loco_env_base.py:886

    def get_obs_idx(self, key):
        return [self.spec.obs_idx(key)]

    def get_all_observation_keys(self):
        return list(self.spec.obs_keys)


# ------------------------------------------------------------------------------ registry
class LocoEnvBase:
    """Registry + factory with the reference's `make("Robot.task.dataset")` entry point."""

    _registered_envs = {}

    @classmethod
    def register(cls):
        if cls.__name__ not in LocoEnvBase._registered_envs:
            LocoEnvBase._registered_envs[cls.__name__] = cls

    @staticmethod
    def list_registered_loco_mujoco():
        return list(LocoEnvBase._registered_envs.keys())

    @classmethod
    def get_all_task_names(cls):
        names = []
        for e in cls.list_registered_loco_mujoco():
            env = cls._registered_envs[e]
            for conf in env.valid_task_confs.get_all_combinations():
                names.append(".".join([env.__name__] + list(conf.values())))
        return names

    @staticmethod
    def make(env_name, *args, **kwargs):
        """mushroom Environment.make semantics: 'Name.a.b' -> registered['Name'].generate('a','b')."""
        if "." in env_name:
            parts = env_name.split(".")
            env_name, args = parts[0], list(parts[1:]) + list(args)
        if env_name not in LocoEnvBase._registered_envs:
            raise KeyError(f"environment {env_name} is not registered; known: {LocoEnvBase.list_registered_loco_mujoco()}")
        env = LocoEnvBase._registered_envs[env_name]
        return env.generate(*args, **kwargs) if hasattr(env, "generate") else env(*args, **kwargs)


class UnitreeH1(LocoEnvBase):
    """UnitreeH1 imitation-learning environment (reference: real_humanoid_robots/UnitreeH1.py).
    `num_envs` = 1 gives the reference's scalar API; larger values expose `.vec`."""

    valid_task_confs = ValidTaskConf(tasks=["walk", "run", "carry"], data_types=["real", "perfect"],
                                     non_combinable=[("carry", None, "perfect")])
    _spec_fn = staticmethod(_specs.unitree_h1)
    _default_back = False

    def __init__(self, task="walk", disable_arms=True, disable_back_joint=None, use_foot_forces=False,
                 use_absorbing_states=True, random_start=True, init_step_no=None, reward_type="target_velocity",
                 num_envs=1, device=0, traj_params=None, physics=None, seed=None, **unused):
        if disable_back_joint is None:
            disable_back_joint = self._default_back
        self.spec = self._spec_fn(task, disable_arms=disable_arms, disable_back_joint=disable_back_joint,
                                  use_absorbing_states=use_absorbing_states, reward_type=reward_type)
        if use_foot_forces:                      # the physics object supplies substep_contacts()
            self.spec.with_foot_forces(type(self).__name__)
        traj = None
        if traj_params:
            traj = self.load_trajectory(traj_params)
        self.vec = VecLocoEnv(self.spec, num_envs, device=device, trajectory=traj, physics=physics,
                              random_start=random_start, init_step_no=init_step_no, obs_f64=True, seed=seed)
        self.info = self.vec.info
        self._dataset = None

    def load_trajectory(self, traj_params, warn=True):
        sp = self.spec
        low = np.concatenate([sp.joint_lo, -np.inf * np.ones(sp.n_vel)])
        high = np.concatenate([sp.joint_hi, np.inf * np.ones(sp.n_vel)])
        low[:6], high[:6] = -np.inf, np.inf
        return Trajectory(keys=list(sp.obs_keys), low=low[2:], high=high[2:], joint_pos_idx=np.arange(sp.n_pos),
                          warn=warn, **traj_params)

    @classmethod
    def generate(cls, task="walk", dataset_type="real", traj_path=None, **kwargs):
        check_validity_task_mode_dataset(cls.__name__, task, None, dataset_type, *cls.valid_task_confs.get_all())
        traj_dt = 1 / 500 if dataset_type == "real" else 1 / 100       # base_humanoid_robot.py:164,188
        if traj_path is not None:
            tp = dict(traj_path=traj_path, traj_dt=traj_dt, control_dt=0.01, clip_trajectory_to_joint_ranges=True)
        else:
            # the reference's datasets are not distributed with it (README: external download);
            # fall back to the seeded synthetic trajectory of the same wire format
            sk = {k: kwargs[k] for k in ("disable_arms", "disable_back_joint") if kwargs.get(k) is not None}
            sk.setdefault("disable_back_joint", cls._default_back)
            sp = cls._spec_fn(task, **sk)
            files = synthetic_h1_trajectory_files(sp, traj_dt=traj_dt)
            tp = dict(traj_files=files, traj_dt=traj_dt, control_dt=0.01, clip_trajectory_to_joint_ranges=True)
        return cls(task=task, traj_params=tp, **kwargs)

    # ----- reference single-env API (N = 1 view)
    def _one(self):
        if self.vec.num_envs != 1:
            raise OlyError("scalar API needs num_envs == 1; use .vec for batched stepping")

    def reset(self, obs=None):
        self._one()
        return self.vec.reset(obs=None if obs is None else np.asarray(obs)[None]).cpu().numpy()[0]

    def step(self, action):
        self._one()
        o, r, a, info = self.vec.step(np.asarray(action, dtype=np.float32)[None])
        return o.cpu().numpy()[0], float(r[0].item()), bool(a[0].item()), {}

    def play_trajectory_from_velocity(self, n_episodes=None, n_steps_per_episode=None, render=False, record=False,
                                      recorder_params=None):
        if render or record:
            raise NotImplementedError("rendering is out of scope of the hot path")
        steps = (n_episodes or 1) * (n_steps_per_episode or self.vec.trajectories.trajectory_length)
        return self.vec.play_trajectory_from_velocity(steps)

    def create_dataset(self, ignore_keys=None):
        if ignore_keys is None:
            ignore_keys = ["q_pelvis_tx", "q_pelvis_tz"]            # base_humanoid_robot.py:35-36
        if self._dataset is None:
            self._dataset = self.vec.create_dataset(ignore_keys)
        return deepcopy(self._dataset)

    def get_kinematic_obs_mask(self):
        return self.vec.get_kinematic_obs_mask()

    def get_obs_idx(self, key):
        return self.vec.get_obs_idx(key)

    def get_all_observation_keys(self):
        return self.vec.get_all_observation_keys()

    @property
    def dt(self):
        return self.spec.dt

    def stop(self):
        pass

    def close(self):
        pass

    def set_algorithm_type(self, algorithm_type):            # loco_env_base.py:203-204
        self._algorithm_type = algorithm_type

    def render(self, record=False):
        raise NotImplementedError("rendering / recording is out of scope of the accelerated path")


class Atlas(UnitreeH1):
    """Atlas (reference: real_humanoid_robots/atlas.py; back joints disabled by default :26)."""
    valid_task_confs = ValidTaskConf(tasks=["walk"], data_types=["real", "perfect"])
    _spec_fn = staticmethod(_specs.atlas)
    _default_back = True


class Talos(UnitreeH1):
    """Talos (reference: real_humanoid_robots/talos.py)."""
    valid_task_confs = ValidTaskConf(tasks=["walk"], data_types=["real", "perfect"])
    _spec_fn = staticmethod(_specs.talos)
    _default_back = False




class StickFigureA3(LocoEnvBase):
    """Registry entry for the RL-mode StickFigureA3 (`LocoEnvBase.make("StickFigureA3.run.real",
    algorithm_type=REINFORCEMENT_LEARNING, physics=...)`, show_a3_walk.py:77).  The trajectory
    datasets only matter in imitation mode, which is not on the accelerated path."""
    valid_task_confs = ValidTaskConf(tasks=["walk", "run", "test"], data_types=["real", "perfect"])   # StickFigureA3.py:23-25

    @classmethod
    def generate(cls, task="walk", dataset_type="real", **kwargs):
        from .a3 import StickFigureA3 as _A3
        check_validity_task_mode_dataset(cls.__name__, task, None, dataset_type, *cls.valid_task_confs.get_all())
        return _A3(**kwargs)


UnitreeH1.register()
Atlas.register()
Talos.register()
StickFigureA3.register()
