"""Inert stand-ins for third-party packages the reference imports but which are
absent from this container (mushroom_rl, mujoco, dm_control, transforms3d, ray,
gymnasium, mujoco_viewer).  Used ONLY by gen_golden.py, which runs in the build
container where /root/reference exists, to import individual reference modules
by path so that their own functions can be executed on seeded inputs.

Nothing here computes a result that ends up in a golden vector, with two
labelled exceptions, both restated from the packages' published definitions
because the packages are not installed ("parity unpinned"
at those boundaries):

* ``FakeObservationHelper`` - name->index bookkeeping of mushroom-rl's
  ObservationHelper for 1-dof joints (obs_idx_map / get_from_obs).
* ``transforms3d`` euler/quaternion helpers (static-xyz eulers, w-first quats).
"""
import enum
import sys
import types
from collections import deque

import numpy as np

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


class ObservationType(enum.Enum):
    BODY_POS = 0
    BODY_ROT = 1
    BODY_VEL = 2
    JOINT_POS = 3
    JOINT_VEL = 4
    SITE_POS = 5
    SITE_ROT = 6


class RunningAveragedWindow:
    """mushroom_rl.utils.running_stats.RunningAveragedWindow, restated."""

    def __init__(self, shape=(1,), window_size=50, init_value=None):
        self._shape = shape
        self._window_size = window_size
        self.reset(init_value)

    def reset(self, init_value=None):
        if init_value is None:
            self._avg_buffer = deque(np.zeros((1, *self._shape)), maxlen=self._window_size)
        else:
            self._avg_buffer = deque([init_value], maxlen=self._window_size)
        self._avg_value = self._avg_buffer[0]

    def update_stats(self, sample):
        self._avg_buffer.append(sample)
        self._avg_value = np.mean(self._avg_buffer, axis=0)

    @property
    def mean(self):
        return self._avg_value


class FakeObservationHelper:
    """Index bookkeeping only: every spec entry is a 1-dof joint pos/vel."""

    def __init__(self, observation_spec):
        self.observation_spec = observation_spec
        self.obs_idx_map = {}
        self.joint_pos_idx = []
        self.joint_vel_idx = []
        for i, (key, _name, ot) in enumerate(observation_spec):
            self.obs_idx_map[key] = [i]
            if ot == ObservationType.JOINT_POS:
                self.joint_pos_idx.append(i)
            elif ot == ObservationType.JOINT_VEL:
                self.joint_vel_idx.append(i)

    def get_from_obs(self, obs, key):
        return obs[self.obs_idx_map[key]]

    def get_joint_pos_from_obs(self, obs):
        return obs[self.joint_pos_idx]

    def get_joint_vel_from_obs(self, obs):
        return obs[self.joint_vel_idx]


# ---------------------------------------------------------------- transforms3d
# Restated from the transforms3d documentation: quaternions are (w, x, y, z);
# euler functions default to axes='sxyz' (static frame, rotate about x, then y,
# then z), i.e. R = Rz(ak) @ Ry(aj) @ Rx(ai).

def _euler2mat(ai, aj, ak):
    ci, si = np.cos(ai), np.sin(ai)
    cj, sj = np.cos(aj), np.sin(aj)
    ck, sk = np.cos(ak), np.sin(ak)
    rx = np.array([[1, 0, 0], [0, ci, -si], [0, si, ci]])
    ry = np.array([[cj, 0, sj], [0, 1, 0], [-sj, 0, cj]])
    rz = np.array([[ck, -sk, 0], [sk, ck, 0], [0, 0, 1]])
    return rz @ ry @ rx


def _mat2euler(m):
    m = np.asarray(m, dtype=np.float64)
    cy = np.sqrt(m[0, 0] * m[0, 0] + m[1, 0] * m[1, 0])
    if cy > np.finfo(np.float64).eps * 4.0:
        ax = np.arctan2(m[2, 1], m[2, 2])
        ay = np.arctan2(-m[2, 0], cy)
        az = np.arctan2(m[1, 0], m[0, 0])
    else:
        ax = np.arctan2(-m[1, 2], m[1, 1])
        ay = np.arctan2(-m[2, 0], cy)
        az = 0.0
    return ax, ay, az


def _quat2mat(q):
    w, x, y, z = q
    nq = w * w + x * x + y * y + z * z
    if nq < np.finfo(np.float64).eps:
        return np.eye(3)
    s = 2.0 / nq
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY],
                     [xY + wZ, 1.0 - (xX + zZ), yZ - wX],
                     [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])


def _euler2quat(ai, aj, ak):
    # sxyz: q = qz(ak) * qy(aj) * qx(ai)
    ci, si = np.cos(ai / 2.0), np.sin(ai / 2.0)
    cj, sj = np.cos(aj / 2.0), np.sin(aj / 2.0)
    ck, sk = np.cos(ak / 2.0), np.sin(ak / 2.0)
    return np.array([ci * cj * ck + si * sj * sk,
                     si * cj * ck - ci * sj * sk,
                     ci * sj * ck + si * cj * sk,
                     ci * cj * sk - si * sj * ck])


def _quat2euler(q):
    return _mat2euler(_quat2mat(q))


def _compose(T, R, Z):
    A = np.eye(4)
    A[:3, :3] = np.asarray(R) @ np.diag(np.asarray(Z, dtype=np.float64))
    A[:3, 3] = T
    return A


def install():
    """Register the stubs; idempotent."""
    if "mushroom_rl" in sys.modules and getattr(sys.modules["mushroom_rl"], "_oly_stub", False):
        return
    if REF not in sys.path:
        sys.path.insert(0, REF)

    class _Anything:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __getattr__(self, n):
            return _Anything()

    mr = _pkg("mushroom_rl")
    mr._oly_stub = True

    class Environment:
        _registered_envs = {}

    _mod("mushroom_rl.core", Environment=Environment, Core=_Anything, Agent=_Anything)
    _mod("mushroom_rl.core.serialization")
    _mod("mushroom_rl.environments", MultiMuJoCo=object)
    _pkg("mushroom_rl.utils")
    _mod("mushroom_rl.utils.spaces", Box=_Anything)
    sys.modules["mushroom_rl.utils"].spaces = sys.modules["mushroom_rl.utils.spaces"]
    rs = _mod("mushroom_rl.utils.running_stats", np=np, deque=deque,
              RunningAveragedWindow=RunningAveragedWindow)
    rs.__all__ = ["np", "deque", "RunningAveragedWindow"]
    mj = _mod("mushroom_rl.utils.mujoco", np=np, ObservationType=ObservationType)
    mj.__all__ = ["np", "ObservationType"]
    _mod("mushroom_rl.utils.record", VideoRecorder=_Anything)
    _mod("mushroom_rl.utils.angles", euler_to_mat=None, mat_to_euler=None, euler_to_quat=None)
    _mod("mushroom_rl.utils.preprocessors", RunningStandardization=_Anything)
    _mod("mushroom_rl.utils.torch", to_float_tensor=None, get_gradient=None, zero_grad=None)
    _pkg("mushroom_rl.approximators").Regressor = _Anything
    _mod("mushroom_rl.approximators.parametric", TorchApproximator=_Anything)
    _mod("mushroom_rl.utils.dataset", parse_dataset=None, compute_J=None, arrays_as_dataset=None,
         compute_episodes_length=None)
    _mod("mushroom_rl.utils.value_functions", compute_gae=None)
    _mod("mushroom_rl.utils.minibatches", minibatch_generator=None)
    for name in ("mushroom_rl.algorithms", "mushroom_rl.algorithms.actor_critic",
                 "mushroom_rl.algorithms.actor_critic.deep_actor_critic"):
        _pkg(name)
    _mod("mushroom_rl.algorithms.actor_critic.deep_actor_critic.trpo", TRPO=object)

    mujoco = _pkg("mujoco")
    _mod("mujoco.viewer")
    mujoco.viewer = sys.modules["mujoco.viewer"]
    _mod("mujoco_viewer")
    dmc = _pkg("dm_control")
    dmc.mjcf = _mod("dm_control.mjcf")
    _mod("gymnasium", register=lambda *a, **k: None)

    ray = _mod("ray")
    ray.remote = lambda f: f
    ray.get = lambda x: x

    tf3 = _pkg("transforms3d")
    tf3.euler = _mod("transforms3d.euler", euler2mat=_euler2mat, mat2euler=_mat2euler,
                     euler2quat=_euler2quat, quat2euler=_quat2euler)
    tf3.quaternions = _mod("transforms3d.quaternions", quat2mat=_quat2mat)
    tf3.affines = _mod("transforms3d.affines", compose=_compose)

    # Reference packages registered as bare namespaces so that importing a leaf
    # module does not execute the package __init__ (which pulls every robot,
    # gymnasium registration, ...).
    _pkg("olympic_mujoco", f"{REF}/olympic_mujoco").__file__ = f"{REF}/olympic_mujoco/__init__.py"
    for sub in ("utils", "environments", "environments/base_robot",
                "environments/real_humanoid_robots", "interfaces", "tasks", "enums"):
        _pkg("olympic_mujoco." + sub.replace("/", "."), f"{REF}/olympic_mujoco/{sub}")
    _pkg("rl", f"{REF}/rl")
    for sub in ("envs", "algos", "policies", "distributions"):
        _pkg("rl." + sub, f"{REF}/rl/{sub}")
    _pkg("imitation_lib", f"{REF}/imitation_lib")
    for sub in ("utils", "imitation"):
        _pkg("imitation_lib." + sub, f"{REF}/imitation_lib/{sub}")


def load_reference():
    """Import the reference leaf modules used for golden generation."""
    import importlib

    install()
    ns = types.SimpleNamespace()
    ns.trajectory = importlib.import_module("olympic_mujoco.utils.trajectory")
    ns.checks = importlib.import_module("olympic_mujoco.utils.checks")
    ns.umath = importlib.import_module("olympic_mujoco.utils.math")
    ns.reward = importlib.import_module("olympic_mujoco.utils.reward")
    u = sys.modules["olympic_mujoco.utils"]
    for m in (ns.trajectory, ns.checks, ns.reward):
        for k, v in m.__dict__.items():
            if not k.startswith("_"):
                setattr(u, k, v)
    ns.enums = importlib.import_module("olympic_mujoco.enums.enums")
    ns.mri = importlib.import_module("olympic_mujoco.interfaces.mujoco_robot_interface")
    ns.loco = importlib.import_module("olympic_mujoco.environments.loco_env_base")
    e = sys.modules["olympic_mujoco.environments"]
    e.LocoEnvBase = ns.loco.LocoEnvBase
    e.ValidTaskConf = ns.loco.ValidTaskConf
    ns.base = importlib.import_module("olympic_mujoco.environments.base_robot.base_humanoid_robot")
    ns.h1 = importlib.import_module("olympic_mujoco.environments.real_humanoid_robots.UnitreeH1")
    ns.rewards = importlib.import_module("olympic_mujoco.tasks.rewards")
    t = sys.modules["olympic_mujoco.tasks"]
    t.rewards = ns.rewards
    ns.walking_task = importlib.import_module("olympic_mujoco.tasks.walking_task")
    t.walking_task = ns.walking_task
    ns.robot = importlib.import_module("olympic_mujoco.environments.robot")
    e.robot = ns.robot
    ns.a3 = importlib.import_module("olympic_mujoco.environments.real_humanoid_robots.StickFigureA3")
    ns.wrappers = importlib.import_module("rl.envs.wrappers")
    sys.modules["rl.envs"].WrapEnv = ns.wrappers.WrapEnv
    ns.normalize = importlib.import_module("rl.envs.normalize")
    ns.ppo = importlib.import_module("rl.algos.ppo")
    ns.ilmath = importlib.import_module("imitation_lib.utils.math")
    ns.networks = importlib.import_module("imitation_lib.utils.networks")
    iu = sys.modules["imitation_lib.utils"]
    iu.GailDiscriminatorLoss = ns.ilmath.GailDiscriminatorLoss
    iu.to_float_tensors = ns.ilmath.to_float_tensors
    ns.gail = importlib.import_module("imitation_lib.imitation.gail_TRPO")
    return ns
