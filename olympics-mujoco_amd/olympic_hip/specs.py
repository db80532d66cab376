"""Robot tables for the hot path: what mushroom-rl's ObservationHelper / MuJoCo ctor and
the robot classes derive from the MJCF and the spec lists, as plain data.

Reference sources (file:line under the reference tree):
  observation spec   environments/real_humanoid_robots/UnitreeH1.py:293-356
  action spec        UnitreeH1.py:359-376
  arm removal        UnitreeH1.py:70-84,145-154
  fall thresholds    UnitreeH1.py:176-187
  reward parameters  environments/base_robot/base_humanoid_robot.py:149-154
  joint / actuator order, ranges, ctrlrange: data/unitree_h1/h1.xml:48,88-93,99-169,235-245
  A3: environments/real_humanoid_robots/StickFigureA3.py:69-141, data/stickFigure_A3/a3.xml

The H1 joint and motor lists are transcribed from the MJCF (a data file);
``mjcf_tables.tables_from_mjcf`` re-derives them from an MJCF path and
tests/test_specs.py checks both against tests/golden/h1_tables.npz.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _abi

# ---------------------------------------------------------------------------- UnitreeH1

_H1_OBS_JOINTS = ["pelvis_tx", "pelvis_tz", "pelvis_ty", "pelvis_tilt", "pelvis_list",
                  "pelvis_rotation", "back_bkz", "l_arm_shy", "l_arm_shx", "l_arm_shz", "left_elbow",
                  "r_arm_shy", "r_arm_shx", "r_arm_shz", "right_elbow", "hip_flexion_r",
                  "hip_adduction_r", "hip_rotation_r", "knee_angle_r", "ankle_angle_r",
                  "hip_flexion_l", "hip_adduction_l", "hip_rotation_l", "knee_angle_l",
                  "ankle_angle_l"]
_H1_ARM_JOINTS = ["l_arm_shy", "l_arm_shx", "l_arm_shz", "left_elbow", "r_arm_shy", "r_arm_shx",
                  "r_arm_shz", "right_elbow"]
_H1_ACTIONS = ["back_bkz", "l_arm_shy", "l_arm_shx", "l_arm_shz", "left_elbow", "r_arm_shy",
               "r_arm_shx", "r_arm_shz", "right_elbow", "hip_flexion_r", "hip_adduction_r",
               "hip_rotation_r", "knee_angle_r", "ankle_angle_r", "hip_flexion_l", "hip_adduction_l",
               "hip_rotation_l", "knee_angle_l", "ankle_angle_l"]

# (name, lo, hi) in MJCF document order = qpos/qvel address order (all 1-dof)
_H1_JOINTS = [
    ("pelvis_tx", -500.0, 500.0), ("pelvis_tz", -300.0, 300.0), ("pelvis_ty", -100.0, 200.0),
    ("pelvis_tilt", -1.5708, 1.5708), ("pelvis_list", -1.5708, 1.5708),
    ("pelvis_rotation", -1.5708, 1.5708),
    ("hip_rotation_l", -0.43, 0.43), ("hip_adduction_l", -0.43, 0.43), ("hip_flexion_l", -1.57, 1.57),
    ("knee_angle_l", -0.26, 2.05), ("ankle_angle_l", -0.87, 0.52),
    ("hip_rotation_r", -0.43, 0.43), ("hip_adduction_r", -0.43, 0.43), ("hip_flexion_r", -1.57, 1.57),
    ("knee_angle_r", -0.26, 2.05), ("ankle_angle_r", -0.87, 0.52),
    ("back_bkz", -2.35, 2.35),
    ("l_arm_shy", -2.87, 2.87), ("l_arm_shx", -0.34, 3.11), ("l_arm_shz", -1.3, 4.45),
    ("left_elbow", -1.25, 2.61),
    ("r_arm_shy", -2.87, 2.87), ("r_arm_shx", -3.11, 0.34), ("r_arm_shz", -4.45, 1.3),
    ("right_elbow", -1.25, 2.61),
]
# (joint the motor drives, gear) in <actuator> order; every motor has ctrlrange +-0.95
_H1_MOTORS = [
    ("hip_rotation_l", 200.0), ("hip_adduction_l", 200.0), ("hip_flexion_l", 200.0),
    ("knee_angle_l", 300.0), ("ankle_angle_l", 40.0),
    ("hip_rotation_r", 200.0), ("hip_adduction_r", 200.0), ("hip_flexion_r", 200.0),
    ("knee_angle_r", 300.0), ("ankle_angle_r", 40.0), ("back_bkz", 200.0),
    ("l_arm_shy", 40.0), ("l_arm_shx", 40.0), ("l_arm_shz", 18.0), ("left_elbow", 18.0),
    ("r_arm_shy", 40.0), ("r_arm_shx", 40.0), ("r_arm_shz", 18.0), ("right_elbow", 18.0),
]
_H1_CTRLRANGE = (-0.95, 0.95)


@dataclass
class ILRobotSpec:
    """Everything K1/K5 need, as host arrays (see _abi.IlModel / oly_il_model)."""
    name: str
    obs_keys: List[str]                 # full spec keys, q_* then dq_*
    joint_names: List[str]              # qpos address order
    nq: int
    nv: int
    n_pos: int
    n_vel: int
    qpos_adr: np.ndarray
    qvel_adr: np.ndarray
    joint_lo: np.ndarray                # per spec JOINT_POS entry
    joint_hi: np.ndarray
    action_names: List[str]
    nu: int
    act_to_ctrl: np.ndarray
    ctrl_lo: np.ndarray
    ctrl_hi: np.ndarray
    fall_tests: List[Tuple[str, float, float]]   # (obs key, lo, hi), ordered
    fall_names: List[str]
    n_drop: int = 2
    n_grf: int = 0
    dt: float = 0.01                    # timestep 0.001 * n_substeps 10, loco_env_base.py:46,52
    gamma: float = 0.99
    horizon: int = 1000
    reward_type: int = _abi.REWARD_TARGET_VELOCITY
    target_velocity: float = 1.25
    use_absorbing_states: bool = True
    geom_group: Optional[np.ndarray] = None     # collision-group index per geom id (use_foot_forces)
    grf_pairs: Optional[list] = None            # [(group_a, group_b)] sensor pairs, 3 columns each
    _keep: list = field(default_factory=list, repr=False)

    def with_foot_forces(self, robot_name):
        """use_foot_forces=True (loco_env_base.py:119-124,750-757): appends mean_grf / 1000 columns
        (3 per sensor pair) and switches to n_substeps = 1, n_intermediate_steps = 10."""
        from .robot_data import ROBOTS
        d = ROBOTS[robot_name]
        if d.get("grf_pairs") is None:
            raise NotImplementedError(f"{robot_name}: the reference's ground-force vector and its declared size "
                                      "disagree (atlas.py:336-342); use_foot_forces is unusable there")
        gi = {g: i for i, (g, _) in enumerate(d["collision_groups"])}
        gg = np.full(d["n_geom"], -1, np.int32)
        for g, ids in d["collision_groups"]:
            gg[ids] = gi[g]
        self.geom_group, self.grf_pairs = gg, [(gi[a], gi[b]) for a, b in d["grf_pairs"]]
        self.n_grf = 3 * len(self.grf_pairs)
        return self

    @property
    def n_obs(self) -> int:
        return self.n_pos + self.n_vel - self.n_drop + self.n_grf

    @property
    def n_act(self) -> int:
        return len(self.action_names)

    def obs_idx(self, key: str) -> int:
        """LocoEnvBase.get_obs_idx (loco_env_base.py:1195-1205): index in the created
        observation, i.e. shifted by the two deleted entries (may be negative!)."""
        return self.obs_keys.index(key) - self.n_drop

    @property
    def act_mean(self) -> np.ndarray:      # loco_env_base.py:171
        return (self.ctrl_hi + self.ctrl_lo) / 2.0

    @property
    def act_delta(self) -> np.ndarray:     # loco_env_base.py:172
        return (self.ctrl_hi - self.ctrl_lo) / 2.0

    @property
    def reward_idx(self) -> int:
        if self.reward_type == _abi.REWARD_TARGET_VELOCITY:
            i = self.obs_idx("dq_pelvis_tx")      # loco_env_base.py:801
        elif self.reward_type == _abi.REWARD_X_POS:
            i = self.obs_idx("q_pelvis_tx")       # loco_env_base.py:810: -2, python wraps it
        else:
            return 0
        return i if i >= 0 else i + self.n_obs

    def to_c(self) -> _abi.IlModel:
        """Build the C struct; the backing arrays are kept alive on the spec."""
        import ctypes as C

        def i32(a):
            a = np.ascontiguousarray(a, dtype=np.int32)
            self._keep.append(a)
            return a.ctypes.data_as(_abi.i32p)

        def f64(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            self._keep.append(a)
            return a.ctypes.data_as(_abi.f64p)

        m = _abi.IlModel()
        m.nq, m.nv, m.n_pos, m.n_vel = self.nq, self.nv, self.n_pos, self.n_vel
        m.n_drop, m.n_grf, m.n_act, m.nu = self.n_drop, self.n_grf, self.n_act, self.nu
        m.qpos_adr, m.qvel_adr, m.act_to_ctrl = i32(self.qpos_adr), i32(self.qvel_adr), i32(self.act_to_ctrl)
        m.act_mean, m.act_delta = f64(self.act_mean), f64(self.act_delta)
        m.ctrl_lo, m.ctrl_hi = f64(self.ctrl_lo), f64(self.ctrl_hi)
        m.n_fall = len(self.fall_tests)
        m.fall_idx = i32([self.obs_idx(k) for k, _, _ in self.fall_tests])
        m.fall_lo = f64([lo for _, lo, _ in self.fall_tests])
        m.fall_hi = f64([hi for _, _, hi in self.fall_tests])
        m.use_absorbing_states = int(self.use_absorbing_states)
        m.reward_type, m.reward_idx = int(self.reward_type), int(self.reward_idx)
        m.target_velocity = float(self.target_velocity)
        del C
        return m


def build_il_spec(name, obs_joints, removed_joints, actions, joints, motors, ctrlrange, fall_tests,
                  fall_names, **kw) -> ILRobotSpec:
    """Generic builder: `joints` [(name, lo, hi)] and `motors` [(joint, gear)] in MJCF order
    BEFORE removal; `obs_joints`/`actions` are the spec lists of the robot class."""
    keep_j = [j for j in joints if j[0] not in removed_joints]
    adr = {j[0]: i for i, j in enumerate(keep_j)}
    rng = {j[0]: (j[1], j[2]) for j in keep_j}
    oj = [j for j in obs_joints if j not in removed_joints]
    keep_m = [m for m in motors if m[0] not in removed_joints]
    mid = {m[0]: i for i, m in enumerate(keep_m)}
    act = [a for a in actions if a not in removed_joints]
    n = len(oj)
    return ILRobotSpec(
        name=name, obs_keys=["q_" + j for j in oj] + ["dq_" + j for j in oj],
        joint_names=[j[0] for j in keep_j], nq=len(keep_j), nv=len(keep_j), n_pos=n, n_vel=n,
        qpos_adr=np.array([adr[j] for j in oj], np.int32),
        qvel_adr=np.array([adr[j] for j in oj], np.int32),
        joint_lo=np.array([rng[j][0] for j in oj]), joint_hi=np.array([rng[j][1] for j in oj]),
        action_names=[a + "_actuator" for a in act], nu=len(keep_m),
        act_to_ctrl=np.array([mid[a] for a in act], np.int32),
        ctrl_lo=np.full(len(act), ctrlrange[0]), ctrl_hi=np.full(len(act), ctrlrange[1]),
        fall_tests=fall_tests, fall_names=fall_names, **kw)


def unitree_h1(task: str = "walk", disable_arms: bool = True, disable_back_joint: bool = False,
               use_absorbing_states: bool = True, reward_type: Optional[str] = "target_velocity"
               ) -> ILRobotSpec:
    """UnitreeH1 tables for `task` in {"walk", "run"} (base_humanoid_robot.py:149-154)."""
    if task not in ("walk", "run", "carry"):
        raise ValueError(f"Task \"{task}\" does not exit in the environment UnitreeH1.")
    removed = []
    if disable_arms:
        removed += _H1_ARM_JOINTS
    if disable_back_joint:
        removed += ["back_bkz"]
    pi = np.pi
    fall = [("q_pelvis_ty", -0.3, 0.1),                    # UnitreeH1.py:179
            ("q_pelvis_tilt", -pi / 4.5, pi / 12),         # :180
            ("q_pelvis_list", -pi / 12, pi / 8),           # :181
            ("q_pelvis_rotation", -pi / 8, pi / 8)]        # :182
    names = ["pelvis_y_condition", "pelvis_tilt_condition", "pelvis_list_condition",
             "pelvis_rotation_condition"]
    rt = {"target_velocity": _abi.REWARD_TARGET_VELOCITY, "x_pos": _abi.REWARD_X_POS,
          None: _abi.REWARD_NONE}
    if reward_type not in rt:
        raise NotImplementedError("The specified reward has not been implemented: %s" % reward_type)
    return build_il_spec("UnitreeH1", _H1_OBS_JOINTS, removed, _H1_ACTIONS, _H1_JOINTS, _H1_MOTORS,
                         _H1_CTRLRANGE, fall, names, reward_type=rt[reward_type],
                         target_velocity=2.5 if task == "run" else 1.25,
                         use_absorbing_states=use_absorbing_states)


# ------------------------------------------------------------------------- Atlas / Talos
def _il_robot(name, removed, fall, names, task, reward_type, use_absorbing_states):
    from .robot_data import ROBOTS
    if task not in ("walk",):
        raise ValueError(f"Task \"{task}\" does not exit in the environment {name}.")
    rt = {"target_velocity": _abi.REWARD_TARGET_VELOCITY, "x_pos": _abi.REWARD_X_POS, None: _abi.REWARD_NONE}
    if reward_type not in rt:
        raise NotImplementedError("The specified reward has not been implemented: %s" % reward_type)
    d = ROBOTS[name]
    return build_il_spec(name, d["obs_joints"], removed, d["actions"], d["joints"], d["motors"], d["ctrlrange"],
                         fall, names, reward_type=rt[reward_type], target_velocity=1.25,
                         use_absorbing_states=use_absorbing_states)


_PELVIS_NAMES = ["pelvis_y_condition", "pelvis_tilt_condition", "pelvis_list_condition", "pelvis_rotation_condition"]


def _pelvis_tests(rot):
    pi = np.pi
    return [("q_pelvis_ty", -0.3, 0.1), ("q_pelvis_tilt", -pi / 4.5, pi / 12), ("q_pelvis_list", -pi / 12, pi / 8),
            ("q_pelvis_rotation", -rot, rot)]


def atlas(task="walk", disable_arms=True, disable_back_joint=True, use_absorbing_states=True,
          reward_type="target_velocity") -> ILRobotSpec:
    """Atlas tables (real_humanoid_robots/atlas.py: spec lists, removal :86-112, _has_fallen :118-170)."""
    from .robot_data import ROBOTS
    pi = np.pi
    removed = (ROBOTS["Atlas"]["arm_joints"] if disable_arms else []) + \
              (ROBOTS["Atlas"]["back_joints"] if disable_back_joint else [])
    fall, names = _pelvis_tests(pi / 10), list(_PELVIS_NAMES)
    if not disable_back_joint:
        fall += [("q_back_bky", -pi / 4, pi / 10), ("q_back_bkx", -pi / 10, pi / 10),
                 ("q_back_bkz", -pi / 4.5, pi / 4.5)]
        names += ["back_extension_condition", "back_bending_condition", "back_rotation_condition"]
    return _il_robot("Atlas", removed, fall, names, task, reward_type, use_absorbing_states)


def talos(task="walk", disable_arms=True, disable_back_joint=False, use_absorbing_states=True,
          reward_type="target_velocity") -> ILRobotSpec:
    """Talos tables (real_humanoid_robots/talos.py: removal :86-112, _has_fallen :114-160)."""
    from .robot_data import ROBOTS
    pi = np.pi
    removed = (ROBOTS["Talos"]["arm_joints"] if disable_arms else []) + \
              (ROBOTS["Talos"]["back_joints"] if disable_back_joint else [])
    fall, names = _pelvis_tests(pi / 10), list(_PELVIS_NAMES)
    if not disable_back_joint:
        fall += [("q_back_bky", -pi / 4, pi / 10), ("q_back_bkz", -pi / 10, pi / 10)]
        names += ["back_extension_condition", "back_rotation_condition"]
    return _il_robot("Talos", removed, fall, names, task, reward_type, use_absorbing_states)


# ------------------------------------------------------------------------ StickFigureA3

@dataclass
class A3Spec:
    """StickFigureA3 RL-mode constants (StickFigureA3.py:69-141, walking_task.py:321-353)."""
    nq: int = 25
    nv: int = 24
    nu: int = 12
    n_obs: int = 41
    sim_dt: float = 0.0025
    control_dt: float = 0.025
    frame_skip: int = 10
    swing_duration: float = 0.75
    stance_duration: float = 0.35
    total_duration: float = 1.1
    goal_height_ref: float = 0.80
    goal_speed_ref: float = 0.0
    target_radius: float = 0.20
    mass: float = 0.0                     # mj_getTotalmass(model): supplied by the physics host
    kp: np.ndarray = field(default_factory=lambda: 0.5 * np.array(
        [200, 200, 200, 250, 80, 80] * 2, dtype=np.float64))     # StickFigureA3.py:78-85
    kd: np.ndarray = field(default_factory=lambda: 0.5 * np.array(
        [20, 20, 20, 25, 8, 8] * 2, dtype=np.float64))
    gear: np.ndarray = field(default_factory=lambda: np.ones(12))   # a3.xml:132-143 (no gear)
    # robot.JVRC nominal pose (environments/robot.py:60-86): motors drive qpos[7:19]
    half_sitting_pose_deg: Sequence[float] = (-30, 0, 0, 50, 0, -24, -30, 0, 0, 50, 0, -24,
                                              -3, -9.74, -30, -3, 9.74, -30)
    mirrored_obs: Sequence[float] = tuple(
        [0.1, -1, 2, -3, -4, 5, -6, 13, -14, -15, 16, -17, 18, 7, -8, -9, 10, -11, 12,
         25, -26, -27, 28, -29, 30, 19, -20, -21, 22, -23, 24] + list(range(31, 41)))
    mirrored_acts: Sequence[float] = (6, -7, -8, 9, -10, 11, 0.1, -1, -2, 3, -4, 5)
    clock_inds: Sequence[int] = (31, 32)
    _keep: list = field(default_factory=list, repr=False)

    @property
    def period(self) -> int:              # walking_task.py:353
        return int(np.floor(2 * self.total_duration * (1 / self.control_dt)))

    @property
    def delay_frames(self) -> int:        # walking_task.py:335
        return int(np.floor(self.swing_duration / self.control_dt))

    @property
    def motor_offset(self) -> np.ndarray:  # robot.py:75-86
        nominal = [q * np.pi / 180.0 for q in self.half_sitting_pose_deg]
        return np.array(nominal[:self.nu])

    def to_c(self, clock_lut: np.ndarray) -> _abi.A3Model:
        def f64(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            self._keep.append(a)
            return a.ctypes.data_as(_abi.f64p)
        assert clock_lut.shape == (4, self.period)
        m = _abi.A3Model()
        m.nq, m.nv, m.nu, m.period, m.delay_frames = self.nq, self.nv, self.nu, self.period, self.delay_frames
        m.target_radius, m.mass = self.target_radius, self.mass
        m.goal_height_ref, m.goal_speed_ref = self.goal_height_ref, self.goal_speed_ref
        m.clock_lut, m.motor_offset, m.gear = f64(clock_lut), f64(self.motor_offset), f64(self.gear)
        return m
