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


def _signed_perm(mirrored):
    """(src, sign) of a mirror table: out[:, j] = sign[j] * x[:, src[j]], i.e. x @ M with
    M[i, |mirrored[i]|] = sign(mirrored[i]).  Index 0 arrives encoded as 0.1 so that its sign is +1
    (StickFigureA3.py:118,128-129): truncation towards zero recovers the index."""
    table = np.asarray(mirrored, dtype=np.float64)
    dest = np.trunc(np.abs(table)).astype(np.int64)          # column each input feeds
    if not np.array_equal(np.sort(dest), np.arange(table.size)):
        raise ValueError("mirror table is not a permutation")
    src = np.empty(table.size, dtype=np.int64)
    src[dest] = np.arange(table.size)
    return src, np.sign(table)[src].astype(np.float32)


def _get_symmetry_matrix(mirrored):
    """Dense [n,n] form of the same signed permutation (what rl/envs/wrappers.py:75-82 returns),
    scattered from the (src, sign) pair."""
    src, sign = _signed_perm(mirrored)
    dense = np.zeros((src.size, src.size))
    dense[src, np.arange(src.size)] = sign
    return dense


class WrapEnv:
    """Batch-of-one facade over one scalar environment (rl/envs/wrappers.py:5-22): every result of
    step() / reset() gains a leading axis of length 1, attribute access falls through to the env."""

    def __init__(self, env_fn):
        self.env = env_fn()

    def __getattr__(self, name):
        return getattr(self.env, name)

    @staticmethod
    def _batched(x):
        return np.array([x])

    def reset(self):
        return self._batched(self.env.reset())

    def step(self, action):
        return tuple(self._batched(part) for part in self.env.step(action[0]))

    def render(self):
        self.env.render()


class SymmetricEnv:
    """Mirror-symmetry view of an env (rl/envs/wrappers.py:24-72).  Each of observation / action is
    mirrored EITHER by an index table (signed permutation) OR by a caller-supplied function."""

    def __init__(self, env_fn, mirrored_obs=None, mirrored_act=None, clock_inds=None, obs_fn=None, act_fn=None):
        for what, table, fn in (("action", mirrored_act, act_fn), ("observation", mirrored_obs, obs_fn)):
            if bool(table) == bool(fn):
                raise AssertionError(f"{what}: give mirror indices or a mirror function, exactly one of them")
            if fn is not None and not callable(fn):
                raise AssertionError(f"{what} mirror function must be callable")
        if mirrored_act:
            self._install_table("act", mirrored_act)
        else:
            self.mirror_action = act_fn
        if mirrored_obs:
            self._install_table("obs", mirrored_obs)
        else:
            self.mirror_observation = obs_fn
        self.clock_inds = clock_inds
        self.env = env_fn()

    def _install_table(self, kind, table):
        src, sign = _signed_perm(table)
        setattr(self, f"_{kind}_src", torch.as_tensor(src))
        setattr(self, f"_{kind}_sgn", torch.as_tensor(sign))
        setattr(self, f"{kind}_mirror_matrix", torch.Tensor(_get_symmetry_matrix(table)))   # reference attribute

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
