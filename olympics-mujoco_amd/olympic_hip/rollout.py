"""Vectorised rollout post-processing for PPO and GAIL/VAIL on the GPU.

Reference semantics (file:line under the reference tree):
  PPOBuffer.store / finish_path       rl/algos/ppo.py:56-84     -> RolloutBuffer + oly_return_scan(RETURN)
  PPO.sample loop                     rl/algos/ppo.py:150-198   -> collect()
  advantage normalisation             rl/algos/ppo.py:335-336   -> oly_adv_stats/normalize (ddof 1, eps)
  compute_gae + normalisation (GAIL)  gail_TRPO.py:126-128      -> oly_return_scan(GAE) (ddof 0, 1e-8)

The reference runs n_proc single-env workers, each sampling episodes back to back; here N
environments advance in lock step for T steps and the episode structure is carried by a
flags tensor (ABSORBING = terminal, LAST = segment end).  Per-env semantics are identical.
With several ranks the environments are sharded and the normalisation statistics are the
only thing exchanged (dist.global_stats).
"""
import torch

from . import _abi, dist as odist


class RolloutBuffer:
    """[T,N] device storage of one rollout block (what PPOBuffer keeps as python lists)."""

    def __init__(self, T, N, obs_dim, act_dim, device, reward_dtype=torch.float32):
        """reward_dtype float64 keeps env.step's reward un-narrowed, as PPOBuffer does (the return
        scan is then bit-exact for any reward, rl/algos/ppo.py:74-76)."""
        f32 = dict(dtype=torch.float32, device=device)
        self.T, self.N = T, N
        self.states = torch.empty((T, N, obs_dim), **f32)
        self.actions = torch.empty((T, N, act_dim), **f32)
        self.rewards = torch.empty((T, N), dtype=reward_dtype, device=device)
        self.values = torch.empty((T, N), **f32)
        self.next_values = torch.zeros((T, N), **f32)
        self.flags = torch.zeros((T, N), dtype=torch.uint8, device=device)
        self.returns = torch.empty((T, N), **f32)
        self.advantages = torch.empty((T, N), **f32)
        self.mu = None                       # [T,N,act_dim] policy means, when the sampler stores them (vecstep.DeviceRollout)
        self.mu_from_fused_forward = False
        self.ptr = 0

    def store(self, state, action, reward, value):
        t = self.ptr
        self.states[t], self.actions[t], self.rewards[t], self.values[t] = state, action, reward, value
        self.ptr += 1

    def __len__(self):
        return self.ptr * self.N

    def episode_stats(self):
        """ep_returns / ep_lens of the segments that END inside the block (logging only), in
        env-major, time-ascending order; computed on the device, one copy back."""
        last = (self.flags & _abi.FLAG_LAST).bool()
        last[-1] = True
        csum = torch.cumsum(self.rewards.double(), dim=0)
        n_i, t_i = torch.nonzero(last.t(), as_tuple=True)          # sorted by env, then time
        end = csum[t_i, n_i]
        same = torch.zeros_like(n_i, dtype=torch.bool)
        same[1:] = n_i[1:] == n_i[:-1]
        prev_t = torch.where(same, torch.roll(t_i, 1), torch.full_like(t_i, -1))
        base = torch.where(same, torch.roll(end, 1), torch.zeros_like(end))
        return (end - base).cpu().tolist(), (t_i - prev_t).cpu().tolist()


def _episode_sums(self):
    """(sum of episode returns, sum of episode lengths, episodes) of the segments that end inside the block as ONE [3] f64
    device tensor: the means PPO.train logs (ppo.py:418-421) without copying every episode to the host.  Every stored
    step belongs to exactly one such segment (the block's last row closes the open ones), so the sums are those of
    the rewards and of the steps."""
    last = (self.flags & _abi.FLAG_LAST).bool()
    last[-1] = True
    T, N = self.rewards.shape
    return torch.stack([self.rewards.double().sum(), torch.tensor(float(T * N), dtype=torch.float64, device=self.rewards.device),
                        last.sum().double()])


RolloutBuffer.episode_sums = _episode_sums


class PPORollout:
    """finish_path + advantage normalisation for a whole [T,N] block on the device."""

    def __init__(self, engine, gamma=0.99, lam=0.95, eps=1e-5):
        self.eng, self.gamma, self.lam, self.eps = engine, gamma, lam, eps
        self._stats = torch.empty(3, dtype=torch.float64, device=engine.device)

    MODE, DDOF = _abi.SCAN_RETURN, 1

    def finish(self, buf, normalize=True):
        """returns, advantages (normalised in place when `normalize`): the scan leaves the
        advantage statistics of its own shard behind (same pass), ONE all-gather of 24 B per rank,
        then the normalisation adds the rank triples on the device."""
        self.eng.return_scan(self.MODE, self.gamma, self.lam, buf.rewards, buf.values, buf.next_values,
                             buf.flags, buf.returns, buf.advantages, stats3=self._stats if normalize else None)
        if normalize:
            self._normalize_with(buf.advantages, self._stats, self.DDOF, self._eps())
        return buf.returns, buf.advantages

    def _eps(self):
        return self.eps

    def _normalize_with(self, adv, stats3, ddof, eps):
        parts = odist.gather_stats(stats3)               # [world,3]: 24 B per rank over RCCL / xGMI
        self.eng.adv_normalize(adv, parts, ddof, eps)
        return adv

    def normalize(self, adv, ddof, eps):
        """Statistics + normalisation of an advantage tensor that did not come out of finish()."""
        self.eng.adv_stats(adv, self._stats)
        return self._normalize_with(adv, self._stats, ddof, eps)


class GAERollout(PPORollout):
    """compute_gae(V, x, xn, r, absorbing, last, gamma, lam) + GAIL's normalisation."""

    MODE, DDOF = _abi.SCAN_GAE, 0

    def _eps(self):
        return 1e-8


@torch.no_grad()
def collect(env, policy, critic, buf, max_traj_len, deterministic=False):
    """PPO.sample for N environments in lock step (rl/algos/ppo.py:169-196).

    env: VecLocoEnv-like (reset(env_mask)->obs, step(a)->(obs,r,done,info)); policy/critic:
    torch modules mapping [N,obs] -> [N,act] / [N,1].  An episode segment ends when the env
    reports done, when it reaches max_traj_len (bootstrapped with V(next state), :195-196) or
    at the end of the block."""
    T, N = buf.T, buf.N
    dev = buf.rewards.device
    state = env.reset().to(torch.float32)
    traj_len = torch.zeros(N, dtype=torch.int32, device=dev)
    buf.ptr = 0
    buf.flags.zero_()
    for t in range(T):
        action = policy(state) if deterministic else policy(state)
        value = critic(state).reshape(N)
        next_state, reward, done, _ = env.step(action)
        next_state = next_state.to(torch.float32)
        buf.store(state, action, reward.to(torch.float32), value)
        traj_len += 1
        done = done.bool()
        cut = done | (traj_len >= max_traj_len) | (t == T - 1)
        buf.next_values[t] = critic(next_state).reshape(N)
        buf.flags[t] = (cut.to(torch.uint8) * _abi.FLAG_LAST) | (done.to(torch.uint8) * _abi.FLAG_ABSORBING)
        if bool(cut.any()) and t < T - 1:
            next_state = torch.where(cut.unsqueeze(1), env.reset(env_mask=cut).to(torch.float32), next_state)
            traj_len = torch.where(cut, torch.zeros_like(traj_len), traj_len)
        state = next_state
    return buf


@torch.no_grad()
def get_normalization_params(iters, policy, env, noise_std, engine=None):
    """Observation mean / std from a noisy rollout, as rl/envs/normalize.py:35-48 computes them
    (np.mean(states, 0), np.sqrt(np.var(states, 0) + 1e-8)) - with N environments in lock step
    instead of `procs` ray workers, and the column sums reduced on the device (oly_col_stats).
    Returns float64 numpy arrays like the reference."""
    eng = engine or env.eng
    N = env.num_envs
    steps = max(1, iters // N)
    state = env.reset().to(torch.float32)
    cs = None
    for _ in range(steps):
        cs = eng.col_stats(state.contiguous(), cs)
        action = policy(state)
        action = action + torch.randn_like(action) * noise_std
        state, _, done, _ = env.step(action)
        state = state.to(torch.float32)
        if bool(done.any()):
            state = torch.where(done.bool().unsqueeze(1), env.reset(env_mask=done.bool()).to(torch.float32), state)
    cnt = cs[0]
    mean = cs[1] / cnt
    var = torch.clamp(cs[2] / cnt - mean * mean, min=0.0)
    return mean.cpu().numpy(), torch.sqrt(var + 1e-8).cpu().numpy()
