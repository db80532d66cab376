"""Observation / return normalisation with the reference's interface, statistics on the device.

Reference: rl/envs/normalize.py
  RunningMeanStd.update      :182-208  (parallel-variance merge, count starts at epsilon=1e-4)
  Normalize.step / _obfilt   :127-147  (online update, clip((obs-mean)/sqrt(var+eps), +-clipob))
  PreNormalizer / Normalizer :52-95
  get_normalization_params   :35-48    -> rollout.get_normalization_params

The batch moments come from oly_col_stats (fp64 column sums of the float32 batch, one pass, no
host copy); the merge is the reference's formula evaluated in fp64 torch ops on [D] vectors;
the filter is oly_obs_filter.  The reference wraps ONE env, so a "batch" there is a single
row; here a batch is the [N,D] observation of N vectorised envs.
"""
import numpy as np
import torch


class RunningMeanStd:
    def __init__(self, engine, epsilon=1e-4, shape=()):
        self.eng = engine
        d = int(np.prod(shape)) if shape != () else 1
        self.shape = tuple(shape) if shape != () else ()
        self.mean = torch.zeros(d, dtype=torch.float64, device=engine.device)
        self.var = torch.zeros(d, dtype=torch.float64, device=engine.device)
        self.count = float(epsilon)

    def update(self, x):
        """x: [B,D] (or [B] for shape=()) float32 device tensor."""
        x2 = x.reshape(x.shape[0], -1)
        if x2.dtype != torch.float32:
            x2 = x2.to(torch.float32)
        cs = self.eng.col_stats(x2.contiguous())
        n = float(x2.shape[0])
        batch_mean = cs[1] / n
        batch_var = cs[2] / n - batch_mean * batch_mean          # np.var: biased
        self.update_from_moments(batch_mean, torch.clamp(batch_var, min=0.0), n)

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot
        m2 = self.var * self.count + batch_var * batch_count + delta * delta * self.count * batch_count / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot


class Normalize:
    """Vectorised-env wrapper: filtered observations (and optionally scaled rewards)."""

    def __init__(self, venv, ob_rms=None, ob=True, ret=False, clipob=10.0, cliprew=10.0, online=True, gamma=1.0,
                 epsilon=1e-8):
        self.venv = venv
        self.eng = venv.eng
        info = getattr(venv, "info", None)                     # mushroom-style envs keep the spaces in .info
        self._observation_space = venv.observation_space if hasattr(venv, "observation_space") else info.observation_space
        self._action_space = venv.action_space if hasattr(venv, "action_space") else info.action_space
        d = int(np.prod(self._observation_space.shape))
        self.ob_rms = ob_rms if ob_rms is not None else (RunningMeanStd(self.eng, shape=(d,)) if ob else None)
        self.ret_rms = RunningMeanStd(self.eng, shape=()) if ret else None
        self.clipob, self.cliprew = clipob, cliprew
        self.ret = torch.zeros(self.num_envs, dtype=torch.float32, device=self.eng.device)
        self.gamma, self.epsilon, self.online = gamma, epsilon, online

    def __getattr__(self, attr):
        return getattr(self.venv, attr)

    def step(self, vac):
        obs, rews, news, infos = self.venv.step(vac)
        obs = self._obfilt(obs)
        if self.ret_rms:
            if self.online:
                self.ret_rms.update(self.ret)
            rews = torch.clamp(rews / torch.sqrt(self.ret_rms.var + self.epsilon).to(rews.dtype), -self.cliprew,
                               self.cliprew)
        return obs, rews, news, infos

    def _obfilt(self, obs):
        if not self.ob_rms:
            return obs
        x = obs if obs.dtype == torch.float32 else obs.to(torch.float32)
        x = x.contiguous()
        if self.online:
            self.ob_rms.update(x)
        return self.eng.obs_filter(x, self.ob_rms.mean, self.ob_rms.var, self.epsilon, self.clipob)

    def reset(self, *a, **kw):
        return self._obfilt(self.venv.reset(*a, **kw))

    @property
    def action_space(self):
        return self._action_space

    @property
    def observation_space(self):
        return self._observation_space

    @property
    def num_envs(self):
        return self.venv.num_envs


def Normalizer(*args, **kwargs):
    def _normalizer(venv):
        return Normalize(venv, *args, **kwargs)
    return _normalizer


def PreNormalizer(iters, noise_std, policy, *args, **kwargs):
    """Normalised env whose statistics are first filled by `iters` noisy policy steps
    (normalize.py:52-86; the reference's policy returns (value, action) here)."""

    @torch.no_grad()
    def pre_normalize(env, num_iter):
        online, env.online = env.online, True
        state = env.reset()
        for _ in range(num_iter):
            out = policy(state)
            action = out[1] if isinstance(out, tuple) else out
            action = action + torch.randn_like(action) * noise_std
            state, _, done, _ = env.step(action)
            if bool(done.any()):
                state = env.reset(env_mask=done.bool()) if _takes_mask(env.venv) else env.reset()
        env.online = online

    def _normalizer(venv):
        venv = Normalize(venv, *args, **kwargs)
        pre_normalize(venv, iters)
        return venv
    return _normalizer


def _takes_mask(venv):
    import inspect
    try:
        return "env_mask" in inspect.signature(venv.reset).parameters
    except (TypeError, ValueError):
        return False
