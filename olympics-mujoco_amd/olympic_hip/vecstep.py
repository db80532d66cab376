"""Device-resident PPO rollout for N StickFigureA3 environments: one fused launch per vec step.

Reference loop: PPO.sample (rl/algos/ppo.py:150-198), one env.step per policy forward.  With the
physics readback resident on the device (the synthetic config-3 regime, or blocks a host batcher
staged ahead) nothing in that loop needs the host:

    per vec step:   actor + critic forward  ->  oly_a3_vec_step (K10)
                    [Gaussian sample from pre-drawn noise, memory.store, PD target, contacts,
                     task.step / reward / done / get_obs, cut bookkeeping, bootstrap side list,
                     env.reset() of the cut environments from pre-drawn records]
    per rollout:    host: refill the reset-record pool (the random draws of WalkingTask.reset),
                    one critic pass over the bootstrap side list -> next_values

The step index and the readback row are device counters, so U consecutive steps are captured ONCE
as a HIP graph and replayed T / U times; the eager loop runs the very same launches op by op.

Random draws are inputs: the action noise is a [T,N,nu] block drawn before the rollout,
the reset records (mode, phase, local step sequence: walking_task.py:137-182,353-392) are drawn on
the host into a per-environment ring of `pool_depth` records; an environment that resets more than
pool_depth times within ONE rollout re-uses its oldest record (counted in `last_info`).
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._ffi import OlyError
from .rollout import RolloutBuffer

REC = np.dtype([("mode", "<i4"), ("phase", "<i4"), ("seq_len", "<i4"), ("pad", "<i4"),
                ("seq", "<f8", (_abi.OLY_MAX_SEQ, 4))])
assert REC.itemsize == C.sizeof(_abi.A3ResetRecord) == 656


def draw_reset_records(rs, count, spec, iter_count=0, out=None):
    """`count` draws of WalkingTask.reset's random part, vectorised (walking_task.py:353-392 with
    generate_step_sequence :137-182): phase in {0, period/2}, mode STANDING 0.2 / FORWARD 0.8, step
    height +-h(iteration), first-step offset U(0.095, 0.105), the step index c ~ randint(2,4) after
    which the height accumulates.  Sequences are LOCAL (before transform_sequence); x / z accumulate by
    repeated addition exactly like the reference's loop.  `out`: a REC array of `count` entries to fill
    (e.g. a view of pinned staging memory) instead of a new one."""
    period = int(np.floor(2 * spec.total_duration * (1 / spec.control_dt)))
    rec = np.zeros(count, REC) if out is None else out
    if rec.shape != (count,) or rec.dtype != REC:
        raise OlyError("draw_reset_records: `out` must be a REC array of `count` entries")
    phase = rs.choice([0, period / 2], size=count).astype(np.int64)
    mode = rs.choice([_abi.MODE_STANDING, _abi.MODE_BACKWARD, _abi.MODE_LATERAL, _abi.MODE_FORWARD], size=count,
                     p=[0.2, 0, 0, 0.8])
    h = np.clip((iter_count - 3000) / 8000, 0, 1) * 0.1
    step_height = rs.choice([-h, h], size=count)
    first = rs.uniform(0.095, 0.105, size=count)
    c = rs.randint(2, 4, size=count)
    half = phase == 0.5 * period
    fwd = mode == _abi.MODE_FORWARD
    if np.any((mode != _abi.MODE_FORWARD) & (mode != _abi.MODE_STANDING)):
        raise OlyError("draw_reset_records: only STANDING / FORWARD have non-zero probability in the reference")
    seq = np.zeros((count, _abi.OLY_MAX_SEQ, 4))
    seq[:, 0, 1] = np.where(half, -1 * first, 1 * first)
    x = np.zeros(count)
    y = np.where(half, -0.15, 0.15)
    z = np.zeros(count)
    for i in range(1, 20):
        x = x + 0.3
        y = y * -1
        z = np.where(i > c, z + step_height, z)
        seq[:, i, 0], seq[:, i, 1], seq[:, i, 2] = x, y, z
    seq[~fwd, 1:] = 0.0
    rec["mode"], rec["phase"], rec["seq_len"], rec["pad"] = mode, phase, np.where(fwd, 20, 1), 0
    rec["seq"] = seq
    return rec


def gaussian_head(policy, state):
    """(mean [N,A], per-dimension std [A]) of a diagonal-Gaussian policy whose std does not depend on
    the state (fixed_std, the reference's PPO setting: rl/policies/actor.py:152-158)."""
    validate = torch.distributions.Distribution._validate_args
    torch.distributions.Distribution.set_default_validate_args(False)     # the check syncs with the host
    try:
        pdf = policy.distribution(state)
    finally:
        torch.distributions.Distribution.set_default_validate_args(validate)
    return pdf.loc, pdf.scale


class TorchForward:
    """Actor mean + critic value with the modules' own torch ops."""

    def __init__(self, policy, critic):
        self.policy, self.critic = policy, critic

    def std(self, state, act_dim):
        _, scale = gaussian_head(self.policy, state)
        scale = torch.as_tensor(scale, dtype=torch.float32, device=state.device)
        if scale.dim() == 2:
            if scale.stride(0) != 0 and scale.shape[0] > 1 and not bool((scale == scale[:1]).all()):
                raise OlyError("state-dependent policy std: use PPO.sample_vec's per-step path")
            scale = scale[0]
        return scale.reshape(-1).expand(act_dim).contiguous()

    def __call__(self, state):
        mu, _ = gaussian_head(self.policy, state)
        return mu.contiguous(), self.critic(state).reshape(state.shape[0]).contiguous()


class A3DeviceRollout:
    """PPO.sample for a VecA3Env whose physics readback is a set of [K,N,...] device blocks."""

    def __init__(self, env, blocks, pool_depth=4, rs=None, keep_rew6=False):
        self.env, self.eng, self.spec = env, env.eng, env.spec
        self.blocks = blocks
        self.N = int(blocks["qpos"].shape[1])
        if self.N != env.num_envs:
            raise OlyError(f"blocks hold {self.N} environments, the env {env.num_envs}")
        self.rs = rs if rs is not None else np.random
        self.depth = int(pool_depth)
        self.keep_rew6 = keep_rew6
        dev, N = self.eng.device, self.N
        self.pool = torch.zeros(N * self.depth * REC.itemsize, dtype=torch.uint8, device=dev)
        self.pool_count = torch.zeros(N, dtype=torch.int32, device=dev)
        self.ctr = torch.zeros(self.eng.a3_vec_ctr_len(N), dtype=torch.int32, device=dev)
        self.state_obs = torch.zeros((N, self.spec.n_obs), dtype=torch.float32, device=dev)
        self.pd_target = torch.zeros((N, self.spec.nu), dtype=torch.float64, device=dev)
        self.traj_len = torch.zeros(N, dtype=torch.int32, device=dev)
        self.scale = torch.zeros(self.spec.nu, dtype=torch.float32, device=dev)
        # two pinned stashes of pre-drawn records, used alternately: one is being uploaded (asynchronously) while the
        # other is refilled on the host behind the next rollout's kernels
        cap = N * self.depth
        self._stash_t = [torch.empty(cap * REC.itemsize, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self._stash = [t.numpy().view(REC) for t in self._stash_t]
        self._stash_n, self._stash_i = 0, 0
        self._last_total = N
        self._env_ids = torch.arange(N, device=dev)
        self._shape = None
        self._graphs = {}
        self.last_info = {}
        self._refill(np.full(N, self.depth))

    # ------------------------------------------------------------------ reset-record pool
    def _predraw(self, want):
        """Top the current stash up to `want` records (host RNG: the random part of WalkingTask.reset)."""
        want = min(int(want), len(self._stash[0]))
        if self._stash_n < want:
            draw_reset_records(self.rs, want - self._stash_n, self.spec, getattr(self.env, "iteration_count", 0),
                               out=self._stash[self._stash_i][self._stash_n:want])
            self._stash_n = want

    def _refill(self, consumed):
        """Replace the `consumed[n]` oldest records of every ring and rewind the cursors.  The records come from the
        pinned stash (drawn while the rollout ran: rollout() calls _predraw right after its last launch), go up in ONE
        asynchronous copy and are scattered into the rings on the device; records drawn but not needed are dropped."""
        consumed = np.minimum(np.asarray(consumed), self.depth)
        total = int(consumed.sum())
        if total:
            self._predraw(total)
            isz, dev = REC.itemsize, self.eng.device
            new = torch.empty(total * isz, dtype=torch.uint8, device=dev)
            new.copy_(self._stash_t[self._stash_i][:total * isz], non_blocking=True)
            cons = torch.as_tensor(consumed, dtype=torch.int64).to(dev, non_blocking=True)
            rows = torch.repeat_interleave(self._env_ids, cons, output_size=total)
            cols = torch.arange(total, device=dev) - (torch.cumsum(cons, 0) - cons)[rows]
            self.pool.view(torch.int64).view(self.N, self.depth, isz // 8)[rows, cols] = new.view(torch.int64).view(total, isz // 8)
            self._last_total = total
        self._stash_i ^= 1          # the other stash is free: its upload finished a whole rollout ago
        self._stash_n = 0
        self.pool_count.zero_()

    # ------------------------------------------------------------------ buffers for one (T, max_traj_len)
    def _ensure(self, T, max_traj_len, deterministic):
        key = (T, max_traj_len, bool(deterministic))
        if self._shape == key:
            return
        for g in self._graphs.values():
            g.reset()
        self._graphs = {}
        dev, N, sp = self.eng.device, self.N, self.spec
        self.buf = RolloutBuffer(T, N, sp.n_obs, sp.nu, dev, reward_dtype=torch.float64)
        nv = torch.zeros(T * N + 1, dtype=torch.float32, device=dev)      # + one dummy slot for unused side rows
        self._nv_flat, self.buf.next_values = nv, nv[:T * N].view(T, N)
        self.rew6 = torch.zeros((T, N, 6), dtype=torch.float32, device=dev) if self.keep_rew6 else None
        # the policy mean behind every stored action: the update phase's old_policy(obs) (ppo.py:236-237, 341)
        self.buf.mu = torch.empty((T, N, sp.nu), dtype=torch.float32, device=dev)
        self.eps = None if deterministic else torch.empty((T, N, sp.nu), dtype=torch.float32, device=dev)
        self.slots = T // max(1, max_traj_len) + 2            # time-limit cuts + the block end, per environment
        self.side_obs = torch.zeros((N * self.slots, sp.n_obs), dtype=torch.float32, device=dev)
        self.side_t = torch.full((N * self.slots,), -1, dtype=torch.int32, device=dev)
        self.side_count = torch.zeros(N, dtype=torch.int32, device=dev)
        self._side_env = torch.arange(N * self.slots, device=dev) // self.slots
        self.mu = torch.zeros((N, sp.nu), dtype=torch.float32, device=dev)
        self.value = torch.zeros(N, dtype=torch.float32, device=dev)
        ro = dict(T=T, max_traj_len=max_traj_len, deterministic=deterministic, side_slots=self.slots,
                  pool_depth=self.depth, mu=self.mu, value=self.value, scale=None if deterministic else self.scale,
                  eps=self.eps, state=self.state_obs, pd_target=self.pd_target, buf_states=self.buf.states,
                  buf_actions=self.buf.actions, buf_rewards=self.buf.rewards, buf_values=self.buf.values,
                  buf_flags=self.buf.flags, buf_rew6=self.rew6, traj_len=self.traj_len, side_obs=self.side_obs,
                  side_t=self.side_t, side_count=self.side_count, pool=self.pool, pool_count=self.pool_count,
                  ctr=self.ctr, buf_mu=self.buf.mu)
        self.launch = self.eng.a3_vec_prepare(self.blocks, self.env.state, ro)
        self._shape = key

    # ------------------------------------------------------------------ the rollout
    @torch.no_grad()
    def rollout(self, policy, critic, T, max_traj_len, deterministic=False, anneal=1.0, graph=True, graph_steps=8,
                forward=None, persistent=None):
        """Fills and returns the RolloutBuffer (states, actions, float64 rewards, values, next_values,
        flags), exactly what PPO.sample + finish_path's bootstrap need.  `forward(state) -> (mu, value)`
        defaults to the fused K11 forward (the modules' torch forward for other network shapes).
        persistent (default: whenever the forward is the fused one): all T steps in ONE launch (K13,
        oly_a3_rollout_persistent) instead of T x (K11 + K10); the buffers are bit-identical."""
        self._ensure(int(T), int(max_traj_len), deterministic)
        fw = forward if forward is not None else self._default_forward(policy, critic)
        if hasattr(fw, "refresh"):
            fw.refresh()                      # the optimiser moved the weights since the last rollout
        N, dev, buf = self.N, self.eng.device, self.buf
        # env.reset(): every environment takes its next record (RESET_ALL writes the first observation)
        self.traj_len.zero_()
        self.side_count.zero_()
        self.side_t.fill_(-1)
        self.ctr[0::2] = 0
        buf.ptr = T
        if not deterministic:
            self.eps.normal_()
            self.scale.copy_(fw.std(self.state_obs, self.spec.nu) * float(anneal))
        self.launch(_abi.VSTEP_RESET_ALL)
        from .mlp import FusedMLPForward
        fused = isinstance(fw, FusedMLPForward)
        buf.mu_from_fused_forward = fused                     # K11 / K13 arithmetic: bit-identical to a later oly_mlp_forward2
        if persistent is None:
            persistent = fused
        if persistent:
            if not fused:
                raise OlyError("persistent rollout needs the fused MLP forward (2 x 256 relu actor / critic)")
            mu, v = fw.outputs(N)
            self.launch.persistent(fw.packed_a, fw.norm_a, fw.packed_c, fw.norm_c, mu, v)
            self._predraw(1.15 * self._last_total + 64)       # host work behind the kernel
            self._finalize(fw, critic)
            return buf

        def one_step():
            mu, value = fw(self.state_obs)
            self.launch(0, mu, value)
        U = max(1, min(int(graph_steps), T)) if graph else 0
        t = 0
        if U:
            g = self._graph(one_step, U, fw)
            while t + U <= T:
                g.replay()
                t += U
        while t < T:
            one_step()
            t += 1
        self._predraw(1.15 * self._last_total + 64)           # host work behind the queued launches
        self._finalize(fw, critic)
        if U and not fused:
            # A graph that holds torch GEMM / elementwise nodes lives for ONE rollout.  profiles/r02/graph_drift: a
            # captured torch graph replays bit-exactly until [synchronize -> kernel write into a newly allocated
            # block >= 1 MB] happens between two replays, which is exactly what an update phase does; the cause
            # sits inside the runtime / the captured temporaries.  Graphs of this library's own two kernels (the
            # fused forward) use no torch temporaries and stay cached.
            self.close()
        return buf

    def _default_forward(self, policy, critic):
        """The fused MFMA forward (K11) when the modules have the reference's 2 x 256 relu structure,
        else their own torch forward."""
        from .mlp import FusedMLPForward
        key = (id(policy), id(critic))
        if getattr(self, "_fw_key", None) != key:
            self._fw_key = key
            self._fw = (FusedMLPForward(self.eng, policy, critic) if FusedMLPForward.supports(policy, critic)
                        else TorchForward(policy, critic))
        return self._fw

    def _graph(self, one_step, U, fw):
        # the launch arguments a capture bakes in: buffers (shape key), the forward object, its input-normalisation
        # switches (the critic normalises only in eval mode: a train()/eval() flip needs a new capture)
        key = (U, id(fw), self._shape, getattr(fw, "norm_a", None), getattr(fw, "norm_c", None))
        if key not in self._graphs:
            from .ppo import _graph_streams
            side, cap = _graph_streams(self.eng.device)
            # warm-up outside capture (BLAS workspaces, lazy initialisations), on a copy of the mutable state
            snap = self._snapshot()
            side.wait_stream(torch.cuda.current_stream(self.eng.device))
            with torch.cuda.stream(side):
                one_step()
            torch.cuda.current_stream(self.eng.device).wait_stream(side)
            self._restore(snap)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):
                for _ in range(U):
                    one_step()
            self._restore(snap)            # capture does not execute, but keep the invariant explicit
            self._graphs[key] = g
        return self._graphs[key]

    def _mutable(self):
        st = self.env.state
        return [st[k] for k in _abi.A3_STATE_FIELDS] + [self.state_obs, self.traj_len, self.side_count, self.side_t,
                                                        self.pool_count, self.ctr]

    def _snapshot(self):
        return [t.clone() for t in self._mutable()]

    def _restore(self, snap):
        for t, s in zip(self._mutable(), snap):
            t.copy_(s)

    def _finalize(self, fw, critic):
        """next_values: V(s_{t+1}) is the next row's value for an uncut step; for a cut that is not
        terminal it is the critic on the side-list row (the observation BEFORE the reset)."""
        buf, T, N = self.buf, self.buf.T, self.N
        if T > 1:
            buf.next_values[:-1].copy_(buf.values[1:])
        v_side = (fw.value(self.side_obs) if hasattr(fw, "value")
                  else critic(self.side_obs).reshape(-1).to(torch.float32))
        valid = self.side_t >= 0
        lin = torch.where(valid, self.side_t.long() * N + self._side_env, torch.full_like(self._side_env, T * N))
        self._nv_flat.index_put_((lin,), v_side)
        # one host round trip per rollout: pool cursors (to refill the consumed records), the readback
        # cursor, and the side-list high-water mark
        host = torch.cat([self.pool_count, self.side_count.max().reshape(1), self.ctr[1:2],
                          valid.sum().to(torch.int32).reshape(1), self.ctr[-2:-1]]).cpu().numpy()
        consumed, hw, k, side_rows = host[:N], int(host[N]), int(host[N + 1]), int(host[N + 2])
        if int(host[N + 3]):
            self.ctr[-2:] = 0
            raise OlyError("a vec step was launched with its device step counter outside [0, T): nothing was written "
                           "by that launch; rewind ctr[0::2] before a rollout")
        if hw > self.slots:
            raise OlyError(f"bootstrap side list overflow: {hw} cuts in one environment, {self.slots} slots")
        self.last_info = dict(resets=int(consumed.sum()), reused_records=int(np.maximum(consumed - self.depth, 0).sum()),
                              side_rows=side_rows)
        phys = getattr(self.env, "physics", None)
        if phys is not None and hasattr(phys, "k"):
            phys.k = k % int(self.blocks["qpos"].shape[0])
        self._refill(consumed)

    def close(self):
        for g in self._graphs.values():
            g.reset()
        self._graphs = {}
