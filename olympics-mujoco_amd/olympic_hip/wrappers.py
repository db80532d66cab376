"""Vector / symmetry wrappers with the reference's interface (rl/envs/wrappers.py).

  WrapEnv          :5-22   batch-of-one view of a single env
  SymmetricEnv     :24-72  mirror_action / mirror_observation / mirror_clock_observation
  _get_symmetry_matrix :75-82

The reference builds dense signed permutation matrices and multiplies; a signed permutation
is an index vector and a sign vector, applied here as one gather + multiply (exactly the
same numbers: every output is +-1 times one input).  The dense matrices stay available.
"""
import numpy as np
import torch


class WrapEnv:
    """Gives a vectorized interface to a single environment."""

    def __init__(self, env_fn):
        self.env = env_fn()

    def __getattr__(self, attr):
        return getattr(self.env, attr)

    def step(self, action):
        state, reward, done, info = self.env.step(action[0])
        return np.array([state]), np.array([reward]), np.array([done]), np.array([info])

    def render(self):
        self.env.render()

    def reset(self):
        return np.array([self.env.reset()])


def _get_symmetry_matrix(mirrored):
    numel = len(mirrored)
    mat = np.zeros((numel, numel))
    for i, j in zip(np.arange(numel), np.abs(np.array(mirrored).astype(int))):
        mat[i, j] = np.sign(mirrored[i])
    return mat


def _signed_perm(mirrored):
    """out[:, j] = sign[i] * x[:, i] for the single i with |mirrored[i]| == j  (x @ M)."""
    m = np.asarray(mirrored, dtype=np.float64)
    src = np.empty(len(m), dtype=np.int64)
    sgn = np.empty(len(m), dtype=np.float32)
    cols = np.abs(m).astype(int)
    if sorted(cols.tolist()) != list(range(len(m))):
        raise ValueError("mirror table is not a permutation")
    for i, j in enumerate(cols):
        src[j] = i
        sgn[j] = np.sign(m[i])
    return src, sgn


class SymmetricEnv:
    def __init__(self, env_fn, mirrored_obs=None, mirrored_act=None, clock_inds=None, obs_fn=None, act_fn=None):
        assert (bool(mirrored_act) ^ bool(act_fn)) and (bool(mirrored_obs) ^ bool(obs_fn)), \
            "You must provide either mirror indices or a mirror function, but not both, for observation and action."
        if mirrored_act:
            self.act_mirror_matrix = torch.Tensor(_get_symmetry_matrix(mirrored_act))
            self._act_src, self._act_sgn = (torch.as_tensor(a) for a in _signed_perm(mirrored_act))
        elif act_fn:
            assert callable(act_fn), "Action mirror function must be callable"
            self.mirror_action = act_fn
        if mirrored_obs:
            self.obs_mirror_matrix = torch.Tensor(_get_symmetry_matrix(mirrored_obs))
            self._obs_src, self._obs_sgn = (torch.as_tensor(a) for a in _signed_perm(mirrored_obs))
        elif obs_fn:
            assert callable(obs_fn), "Observation mirror function must be callable"
            self.mirror_observation = obs_fn
        self.clock_inds = clock_inds
        self.env = env_fn()

    def __getattr__(self, attr):
        return getattr(self.env, attr)

    def _on(self, name, device):
        """Mirror tables cached per device (no host->device copy per call; graph-capturable)."""
        cache = self.__dict__.setdefault("_dev_tables", {})
        key = (name, str(device))
        if key not in cache:
            cache[key] = getattr(self, name).to(device)
        return cache[key]

    def mirror_action(self, action):
        return action[..., self._on("_act_src", action.device)] * self._on("_act_sgn", action.device)

    def mirror_observation(self, obs):
        return obs[..., self._on("_obs_src", obs.device)] * self._on("_obs_sgn", obs.device)

    def mirror_clock_observation(self, obs):
        """Mirror, then shift the clock entries by half a period: sin(arcsin(x) + pi) = -x
        (reference :59-72, history length fixed to 1)."""
        out = torch.zeros_like(obs)
        n = self.base_obs_len
        block = self.mirror_observation(obs[:, :n])
        for i in self.clock_inds:
            block[:, i] = torch.sin(torch.arcsin(block[:, i]) + np.pi)
        out[:, :n] = block
        return out
