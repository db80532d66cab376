"""PPO (clip objective) with the reference's agent interface on top of the vectorised engine.

Reference: rl/algos/ppo.py
  PPO.__init__ args / log files        :94-132
  PPO.sample / sample_parallel         :150-230  -> sample_vec (N envs in lock step, no ray)
  PPO.update_policy                    :232-282  (clip loss, value loss, entropy, mirror loss)
  PPO.train                            :284-477  (anneal / minibatch epochs / CSV logs / eval)
  PPOBuffer.finish_path + adv-norm     :68-84, :335-336 -> HIP kernels via rollout.PPORollout

Where the reference fans n_proc single-env workers out over ray, here `env_fn()` returns ONE
vectorised env and `num_procs` maps to its num_envs; the optimiser side is PyTorch (MFMA GEMMs)
with the loss terms and their gradients evaluated by one HIP kernel (update_policy_fused;
update_policy is the plain-torch evaluation of the same terms).  Actor / critic modules are the reference's own (anything with
`forward(state, deterministic, anneal)`, `distribution(obs)` and a critic `forward(obs)`).
"""
import os
import time
from copy import deepcopy

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

from . import _abi
from .rollout import PPORollout, RolloutBuffer


class MLPGaussianActor(torch.nn.Module):
    """Shape-compatible stand-in for rl.policies.actor.Gaussian_FF_Actor (relu MLP, fixed
    std): used by tests and examples when the reference's policy classes are not importable."""

    def __init__(self, state_dim, action_dim, layers=(256, 256), fixed_std=None):
        super().__init__()
        dims = [state_dim] + list(layers)
        self.actor_layers = torch.nn.ModuleList([torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(layers))])
        self.means = torch.nn.Linear(layers[-1], action_dim)
        std = fixed_std if fixed_std is not None else torch.exp(torch.tensor(-1.5))
        self.register_buffer("fixed_std", torch.as_tensor(std, dtype=torch.float32), persistent=False)
        self.obs_mean, self.obs_std = 0.0, 1.0

    def _mean(self, state):
        x = (state - self.obs_mean) / self.obs_std
        for lin in self.actor_layers:
            x = torch.relu(lin(x))
        return self.means(x)

    def forward(self, state, deterministic=True, anneal=1.0):
        mu = self._mean(state)
        if deterministic:
            return mu
        return torch.distributions.Normal(mu, self.fixed_std * anneal, validate_args=False).sample()

    def distribution(self, inputs):
        return torch.distributions.Normal(self._mean(inputs), self.fixed_std)


class MLPCritic(torch.nn.Module):
    """Stand-in for rl.policies.critic.FF_V."""

    def __init__(self, state_dim, layers=(256, 256)):
        super().__init__()
        dims = [state_dim] + list(layers)
        self.critic_layers = torch.nn.ModuleList([torch.nn.Linear(dims[i], dims[i + 1]) for i in range(len(layers))])
        self.network_out = torch.nn.Linear(layers[-1], 1)

    def forward(self, inputs):
        x = inputs
        for lin in self.critic_layers:
            x = torch.relu(lin(x))
        return self.network_out(x)


def _plain_std(std, B, A):
    """Normal() broadcasts a scalar / per-dim scale to a stride-0 [B,A] view: undo that so the
    kernel reads one value (or one row) and autograd reduces the gradient for us."""
    if std.dim() == 2 and tuple(std.shape) == (B, A):
        st = std.stride()
        if st == (0, 0):
            return std[0, :1]
        if st[0] == 0:
            return std[0]
        return std.contiguous()
    return std.reshape(-1)


class FusedPPOLoss(torch.autograd.Function):
    """oly_ppo_loss: (actor_loss, entropy_penalty, critic_loss, approx_kl, clip_fraction) and
    the gradients w.r.t. mu / std / value from the same pass (ppo.py:236-259,270-273)."""

    @staticmethod
    def forward(ctx, eng, mu, std, old_mu, old_std, action, adv, ret, value, clip, vf_coeff):
        B, A = mu.shape
        need_std = std.requires_grad
        r = eng.ppo_loss(mu.contiguous(), std.contiguous(), old_mu.contiguous(), old_std.contiguous(),
                         action.contiguous(), adv.reshape(B).contiguous(), ret.reshape(B).contiguous(),
                         value.reshape(B).contiguous(), clip, vf_coeff, want_grad=True, want_grad_std=need_std)
        ctx.save_for_backward(r["grad_mu"], r["grad_value"], r["grad_std"] if need_std else None,
                              std if need_std else None)
        ctx.value_shape, ctx.std_shape = value.shape, std.shape
        out = r["scal"].to(torch.float32)
        outs = tuple(out[i].clone() for i in range(5))
        ctx.mark_non_differentiable(outs[3], outs[4])
        return outs

    @staticmethod
    def backward(ctx, g_actor, g_ent, g_critic, g_kl, g_cf):
        gmu, gv, gsd, std = ctx.saved_tensors
        grad_mu = g_actor * gmu
        grad_value = (g_critic * gv).reshape(ctx.value_shape)
        grad_std = None
        if gsd is not None:
            B, A = gmu.shape
            full = g_actor * gsd
            if std.numel() == 1:
                grad_std = full.sum().reshape(ctx.std_shape) - g_ent / std
            elif std.numel() == A:
                grad_std = full.sum(0).reshape(ctx.std_shape) - g_ent / (A * std)
            else:
                grad_std = full - g_ent / (B * A * std)
        return None, grad_mu, grad_std, None, None, None, None, None, grad_value, None, None


class FusedMirrorLoss(torch.autograd.Function):
    """oly_mirror_loss: mean((det - mirror_action(mir))^2) with the signed permutation applied
    in-kernel (ppo.py:261-268, wrappers.py:51-57)."""

    @staticmethod
    def forward(ctx, eng, det, mir, src, sign):
        loss, gd, gm = eng.mirror_loss(det.contiguous(), mir.contiguous(), src, sign)
        ctx.save_for_backward(gd, gm)
        return loss.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        gd, gm = ctx.saved_tensors
        return None, g * gd, g * gm, None, None


class KernelUpdate:
    """One PPO minibatch update (rl/algos/ppo.py:232-282,396-410) as three launches of this repository's own kernels:
      oly_ppo_update_grads (K14)  actor and critic forward, the loss terms of update_policy (mirror loss included) and the
                                  backward pass on the f32 matrix cores, + the finishing launch that adds the parts;
      oly_ppo_adam_step           torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step for both networks on flat
                                  buffers, the stepped weights written into the packed streams in the same launch.
    The modules keep their parameters: `.data` of every weight / bias becomes a view of one flat buffer per network
    (same values), `.grad` a view of the flat gradient buffer the kernel writes; Adam's moments and the step count live
    here.  Per iteration (`begin`): the old policy's means of every rollout row (the means the rollout kernels stored, or
    one K11 forward: the old policy is constant during an update phase) and the mirrored observations (the env's own
    mirror_clock_observation, any function).  Per minibatch (`step`) or per epoch (`epoch`, one C call): the rows are
    gathered by index inside the kernel."""

    def __init__(self, eng, policy, critic, old_policy, clip, vf_coeff, mirror_coeff, obs_mirr=None, act_src=None,
                 act_sign=None, lr=1e-4, eps=1e-5, max_grad_norm=0.05, betas=(0.9, 0.999)):
        from ._ffi import lib
        from .mlp import FusedMLPForward
        self.eng, self.policy, self.critic, self.old_policy = eng, policy, critic, old_policy
        self.fw = FusedMLPForward(eng, policy, critic)
        self.fw_old = None
        self.clip, self.vf_coeff, self.mirror_coeff = float(clip), float(vf_coeff), float(mirror_coeff)
        self.lr, self.eps, self.max_grad_norm, self.betas = float(lr), float(eps), float(max_grad_norm), betas
        self.obs_mirr = obs_mirr if act_src is not None else None
        dev = eng.device
        self.act_src = act_src.to(dev, torch.int32).contiguous() if act_src is not None else None
        self.act_sign = act_sign.to(dev, torch.float32).contiguous() if act_sign is not None else None
        in_dim, act_dim = self.fw.in_dim, self.fw.act_dim
        self.nets = []
        for parts, out_dim, packed in ((self.fw.pa, act_dim, self.fw.packed_a), (self.fw.pc, 1, self.fw.packed_c)):
            gf = int(lib().oly_ppo_update_grad_floats(in_dim, 256, out_dim))
            z = lambda: torch.zeros(gf, dtype=torch.float32, device=dev)
            nt = dict(param=z(), grad=z(), exp_avg=z(), exp_avg_sq=z(), out_dim=out_dim, packed=packed)
            off = 0
            with torch.no_grad():
                for lin in parts:                           # W1 | b1 | W2 | b2 | W3 | b3: the kernel's flat order
                    for prm in (lin.weight, lin.bias):
                        n = prm.numel()
                        nt["param"][off:off + n].copy_(prm.detach().reshape(-1))
                        prm.data = nt["param"][off:off + n].view_as(prm)
                        prm.grad = nt["grad"][off:off + n].view_as(prm)
                        off += n
            assert off == gf
            self.nets.append(nt)
        self.grad_actor, self.grad_critic = self.nets[0]["grad"], self.nets[1]["grad"]
        self.scal = torch.zeros(6, dtype=torch.float64, device=dev)
        self.adam_ws = torch.zeros(1024, dtype=torch.float64, device=dev)
        self.steps = 0
        self._ws, self._launch, self._apply, self._norm_ready = {}, {}, None, False
        self.old_mu = self.mir_obs = None
        self._old_mu_buf = None

    @staticmethod
    def supports(policy, critic, act_mirr_is_table):
        from .mlp import FusedMLPForward
        return (FusedMLPForward.supports(policy, critic) and act_mirr_is_table
                and next(policy.parameters()).is_cuda and policy.means.out_features <= 16)

    def _std(self, module):
        """(std [A], log std [A]) of a fixed-std Gaussian policy (actor.py:199-201)."""
        t = torch.as_tensor(module.fixed_std, dtype=torch.float32, device=self.eng.device).reshape(-1)
        sd = t.expand(self.fw.act_dim).contiguous() if t.numel() == 1 else t.contiguous()
        return sd, torch.log(sd)

    @torch.no_grad()
    def begin(self, observations, sampled_mu=None):
        """Once per iteration, after old_policy.load_state_dict(policy.state_dict()).  `sampled_mu` [n, act_dim]: the
        means the sampling policy produced for these rows with the fused forward's arithmetic (the rollout kernels store
        them).  The old policy of this update phase IS the sampling policy (ppo.py:341), so they are taken as
        old_policy(obs) when the two modules' input normalisation tables agree as well; otherwise (or without them) the
        old policy runs over the buffer once (K11)."""
        from .mlp import FusedMLPForward
        self.fw.refresh()
        self.nets[0]["packed"], self.nets[1]["packed"] = self.fw.packed_a, self.fw.packed_c
        if self.fw_old is None:
            self.fw_old = FusedMLPForward(self.eng, self.old_policy, self.critic)
        else:
            self.fw_old.refresh()
        n = int(observations.shape[0])
        reuse = (sampled_mu is not None and tuple(sampled_mu.shape) == (n, self.fw.act_dim) and sampled_mu.is_contiguous()
                 and self.fw_old.norm_a == self.fw.norm_a
                 and all(a is b or (a is not None and b is not None and torch.equal(a, b))
                         for a, b in zip(self.fw_old.norm_tables()[0], self.fw.norm_tables()[0])))
        if reuse:
            self.old_mu = sampled_mu
        else:
            if self._old_mu_buf is None or self._old_mu_buf.shape[0] != n:
                self._old_mu_buf = torch.empty((n, self.fw.act_dim), dtype=torch.float32, device=self.eng.device)
            self.old_mu = self._old_mu_buf
            self.eng.mlp_forward2(observations, self.fw_old.packed_a, self.fw.act_dim, self.old_mu, normalize_a=True)
        self.mir_obs = self.obs_mirr(observations).to(torch.float32).contiguous() if self.obs_mirr is not None else None
        self.sd, self.log_sd = self._std(self.policy)
        self.old_sd, self.old_log_sd = self._std(self.old_policy)

    def grads(self, observations, actions, returns, advantages, idx, scal=None, repack=False):
        """Gradients of the minibatch `idx` (int32 row indices) into the parameters' .grad; -> scal [6] f64 (device):
        actor_loss, entropy_penalty, critic_loss, approx_kl, mirror_loss, clip_fraction.  The call is validated once per
        (minibatch size, buffers); later minibatches re-issue it with the new index vector only."""
        if repack:
            self.fw.refresh()                               # someone else moved the weights (not `apply`)
        B = int(idx.shape[0])
        adv, ret = advantages.reshape(-1), returns.reshape(-1)
        key = (B, observations.data_ptr(), actions.data_ptr(), adv.data_ptr(), ret.data_ptr(), self.old_mu.data_ptr(),
               None if self.mir_obs is None else self.mir_obs.data_ptr(), self.fw.packed_a.data_ptr(), self.fw.norm_a, self.fw.norm_c)
        scal = self.scal if scal is None else scal
        launch = self._launch.get(key)
        if launch is None:
            if B not in self._ws:
                n_ws, pa, pc = self.eng.ppo_update_plan(B, self.fw.in_dim, self.fw.act_dim, self.mir_obs is not None)
                self._ws[B] = (torch.empty(n_ws, dtype=torch.float32, device=self.eng.device), (pa, pc))
            ws, parts = self._ws[B]
            if len(self._launch) > 8:
                self._launch.clear()
            launch = self._launch[key] = self.eng.ppo_update_grads(
                observations, actions, adv, ret, self.old_mu, self.fw.packed_a, self.fw.packed_c, self.sd, self.log_sd,
                self.old_sd, self.old_log_sd, self.grad_actor, self.grad_critic, scal, ws, idx=idx, mir_obs=self.mir_obs,
                act_src=self.act_src, act_sign=self.act_sign, normalize_actor=self.fw.norm_a, normalize_critic=self.fw.norm_c,
                clip=self.clip, vf_coeff=self.vf_coeff, mirror_coeff=self.mirror_coeff, parts=parts, gnorm_ws=self.adam_ws,
                prepare=True)
        launch(idx, scal)
        self._last_launch = launch
        self._norm_ready = True                             # adam_ws holds the squared-norm partials of these gradients
        return scal

    def apply(self, norm_ready=None):
        """clip_grad_norm_ + Adam.step on both networks and the re-pack of the stepped weights (ppo.py:399-410).  After
        the gradients were changed outside (an all-reduce over ranks) pass norm_ready=False: the norm is formed anew."""
        self.steps += 1
        ready = self._norm_ready if norm_ready is None else bool(norm_ready)
        self._norm_ready = False
        key = (self.fw.packed_a.data_ptr(), self.fw.norm_a, self.fw.norm_c)
        if self._apply is None or self._apply[0] != key:
            for nt, norm, (mean, std) in zip(self.nets, (self.fw.norm_a, self.fw.norm_c), self.fw.norm_tables()):
                nt["in_mean"], nt["in_std"] = (mean, std) if norm else (None, None)
            self._apply = (key, self.eng.ppo_adam_step(self.fw.in_dim, self.steps, self.lr, self.eps, self.max_grad_norm,
                                                       self.nets, self.adam_ws, beta1=self.betas[0], beta2=self.betas[1],
                                                       prepare=True))
        self._apply[1](self.steps, ready)

    def step(self, observations, actions, returns, advantages, idx, scal=None):
        scal = self.grads(observations, actions, returns, advantages, idx, scal)
        self.apply()
        return scal

    def epoch(self, observations, actions, returns, advantages, perm, minibatch, n_batches, scal):
        """The epoch's whole minibatch loop in one C call (oly_ppo_update_epoch): minibatch b = rows perm[b * minibatch :
        (b + 1) * minibatch], scalars into scal[b] ([n_batches, 6] f64), one optimiser step each.  Same launches in the
        same order as n_batches calls of step()."""
        if n_batches <= 0:
            return scal
        # the first minibatch through the ordinary path: it prepares (validates) the two launches
        self.step(observations, actions, returns, advantages, perm[:minibatch], scal[0])
        if n_batches > 1:
            self.eng.ppo_update_epoch(self._last_launch, self._apply[1], self.steps + 1, perm[minibatch:], n_batches - 1, scal[1:])
            self.steps += n_batches - 1
        return scal


_GRAPH_STREAMS = {}


def _graph_streams(device):
    """One warm-up stream and one capture stream per device, reused by every capture: PyTorch keeps a
    BLAS workspace per (handle, stream) for the life of the process, so a fresh stream per capture
    (the default of torch.cuda.graph) leaks ~76 MB each time."""
    key = torch.device(device).index or 0
    if key not in _GRAPH_STREAMS:
        _GRAPH_STREAMS[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
    return _GRAPH_STREAMS[key]


class GraphedUpdate:
    """One PPO minibatch update (fused losses -> backward -> grad clip -> Adam on actor and
    critic) captured once as a HIP graph and replayed per minibatch: at the reference's
    minibatch sizes the update is ~150 tiny launches and purely launch-bound.

    The minibatch is copied into static buffers; parameters, optimiser state and the old
    policy are updated in place, so replay sees their current values."""

    def __init__(self, ppo, eng, batch, obs_dim, act_dim, obs_mirr=None, act_src=None, act_sign=None, warmup=3):
        dev = eng.device
        self.obs = torch.zeros((batch, obs_dim), dtype=torch.float32, device=dev)
        self.act = torch.zeros((batch, act_dim), dtype=torch.float32, device=dev)
        self.ret = torch.zeros((batch, 1), dtype=torch.float32, device=dev)
        self.adv = torch.zeros((batch, 1), dtype=torch.float32, device=dev)
        opts = (ppo.actor_optimizer, ppo.critic_optimizer)
        for o in opts:
            for gparam in o.param_groups:
                if not gparam.get("capturable", False):
                    raise ValueError("GraphedUpdate needs Adam(capturable=True)")
        nets = (ppo.policy, ppo.critic)
        keep = [deepcopy(m.state_dict()) for m in nets]
        keep_opt = [{p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in o.state.items()}
                    for o in opts]

        def body():
            # Normal(validate_args=True) reads its check back to the host: not capturable
            validate = torch.distributions.Distribution._validate_args
            torch.distributions.Distribution.set_default_validate_args(False)
            try:
                return body_()
            finally:
                torch.distributions.Distribution.set_default_validate_args(validate)

        def body_():
            a_l, ent, c_l, kl, m_l, clipf = ppo.update_policy_fused(eng, self.obs, self.act, self.ret, self.adv,
                                                                    obs_mirr, act_src, act_sign)
            for o in opts:
                o.zero_grad(set_to_none=True)
            (a_l + ppo.mirror_coeff * m_l + ppo.ent_coeff * ent + c_l).sum().backward()
            torch.nn.utils.clip_grad_norm_(ppo.policy.parameters(), ppo.grad_clip)
            opts[0].step()
            torch.nn.utils.clip_grad_norm_(ppo.critic.parameters(), ppo.grad_clip)
            opts[1].step()
            return torch.stack([a_l.detach(), ent.detach(), c_l.detach(), kl.detach(), m_l.detach().reshape(()),
                                clipf.detach()])
        side, cap = _graph_streams(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                body()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=cap):
            self.stats = body()
        for m, sd in zip(nets, keep):                         # warm-up steps must not count as training
            m.load_state_dict(sd)
        for o, saved in zip(opts, keep_opt):                  # in place: the graph holds these buffers
            for p, st in o.state.items():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if p in saved:
                            v.copy_(saved[p][k])
                        else:
                            v.zero_()

    def close(self):
        """Release the captured graph and its private memory pool now (not at the next GC pass)."""
        if self.graph is not None:
            self.graph.reset()
            self.graph = None
        self.stats = None

    def __call__(self, observations, actions, returns, advantages, idx):
        torch.index_select(observations, 0, idx, out=self.obs)
        torch.index_select(actions, 0, idx, out=self.act)
        torch.index_select(returns, 0, idx, out=self.ret)
        torch.index_select(advantages, 0, idx, out=self.adv)
        self.graph.replay()
        return self.stats


class GraphedActorCritic:
    """mean / std of the Gaussian policy and the critic value for the N rollout environments as
    ONE replayed HIP graph (the op-by-op forward of the two MLPs is ~40 tiny launches = ~280 us of
    host time per vec step at N = 4096; the replay is one launch).  Forward only.  The sample
    a = mu + (std * anneal) * eps is drawn outside the graph (torch.normal with a tensor std reads
    a validity check back to the host and cannot be captured), which is the same arithmetic
    Normal(mu, std * anneal).sample() performs.  Outputs live in static buffers that the next
    replay overwrites."""

    def __init__(self, policy, critic, num_envs, obs_dim, device, deterministic, anneal, warmup=2):
        self.state = torch.zeros((num_envs, obs_dim), dtype=torch.float32, device=device)
        self.deterministic, self.anneal = deterministic, float(anneal)

        def body():
            validate = torch.distributions.Distribution._validate_args
            torch.distributions.Distribution.set_default_validate_args(False)
            try:
                pdf = policy.distribution(self.state)
                return pdf.loc, pdf.scale, critic(self.state).reshape(num_envs)
            finally:
                torch.distributions.Distribution.set_default_validate_args(validate)
        with torch.no_grad():
            side, cap = _graph_streams(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    body()
            torch.cuda.current_stream(device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=cap):
                self.mu, self.scale, self.value = body()

    def close(self):
        if self.graph is not None:
            self.graph.reset()
            self.graph = None
        self.mu = self.scale = self.value = None

    def __call__(self, state):
        self.state.copy_(state)
        self.graph.replay()
        if self.deterministic:
            return self.mu, self.value
        return self.mu + (self.scale * self.anneal) * torch.randn_like(self.mu), self.value


def _global_episode_means(ep_ret, ep_len, multi_rank, sums=None):
    """(mean return, mean length) over the episodes of every rank (this rank's alone without a process group).
    sums: (sum of returns, sum of lengths, episodes) already reduced on the device (RolloutBuffer.episode_sums)."""
    tot = list(sums) if sums is not None else [float(np.sum(ep_ret)) if len(ep_ret) else 0.0,
                                                float(np.sum(ep_len)) if len(ep_len) else 0.0, float(len(ep_ret))]
    if multi_rank:
        from . import dist as odist
        tot = odist.allreduce_sum(tot)
    return (tot[0] / tot[2], tot[1] / tot[2]) if tot[2] else (0.0, 0.0)


class PPO:
    def __init__(self, args, save_path):
        self.gamma, self.lam = args["gamma"], args["lam"]
        self.lr, self.eps = args["lr"], args["eps"]
        self.ent_coeff, self.clip = args["entropy_coeff"], args["clip"]
        self.minibatch_size, self.epochs = args["minibatch_size"], args["epochs"]
        self.max_traj_len, self.use_gae = args["max_traj_len"], args["use_gae"]
        self.n_proc = args["num_procs"]
        self.grad_clip, self.mirror_coeff = args["max_grad_norm"], args["mirror_coeff"]
        self.eval_freq = args["eval_freq"]
        self.recurrent = False
        self.batch_size = self.n_proc * self.max_traj_len
        self.vf_coeff = 0.5
        self.target_kl = None
        self.total_steps, self.highest_reward, self.iteration_count = 0, -1, 0
        self.save_path = save_path
        os.makedirs(save_path, exist_ok=True)
        self.eval_fn = os.path.join(save_path, "eval.txt")
        with open(self.eval_fn, "w") as out:
            out.write("test_ep_returns,test_ep_lens\n")
        self.train_fn = os.path.join(save_path, "train.txt")
        with open(self.train_fn, "w") as out:
            out.write("ep_returns,ep_lens\n")

    def save(self, policy, critic, suffix=""):
        os.makedirs(self.save_path, exist_ok=True)
        torch.save(policy, os.path.join(self.save_path, "actor" + suffix + ".pt"))
        torch.save(critic, os.path.join(self.save_path, "critic" + suffix + ".pt"))

    # ------------------------------------------------------------------ rollout
    @torch.no_grad()
    def sample_vec(self, env, policy, critic, T, max_traj_len, deterministic=False, anneal=1.0):
        """N envs x T steps; episode cuts as in PPO.sample (:169-196)."""
        if getattr(self, "use_device_rollout", True) and getattr(env, "has_device_physics", False):
            # physics readback resident on the device: one fused launch per vec step, replayed from a graph
            return env.device_rollout(policy, critic, T, max_traj_len, deterministic, anneal,
                                      graph=bool(getattr(self, "use_graph_rollout", False)))
        N, dev = env.num_envs, env.device
        obs_dim = int(np.prod(env.observation_space.shape)) if hasattr(env, "observation_space") else env.spec.n_obs
        act_dim = int(np.prod(env.action_space.shape)) if hasattr(env, "action_space") else env.spec.n_act
        buf = RolloutBuffer(T, N, obs_dim, act_dim, dev)
        state = env.reset().to(torch.float32)
        traj_len = torch.zeros(N, dtype=torch.int32, device=dev)
        from .mlp import FusedMLPForward
        eng_ = getattr(env, "eng", None)
        if eng_ is not None and getattr(self, "fused_forward", True) and FusedMLPForward.supports(policy, critic) \
                and next(policy.parameters()).is_cuda:
            # host-stepped physics: the actor + critic forward of every vec step is still ONE launch (K11)
            # instead of the modules' ten (rl/algos/ppo.py:181-182); the sample is mu + (std * anneal) * eps
            fw = FusedMLPForward(eng_, policy, critic)
            scale = None if deterministic else fw.std(state, act_dim) * float(anneal)

            def ac(s):
                mu, v = fw(s.contiguous())
                if deterministic:
                    return mu.clone(), v.clone()
                return mu + scale * torch.randn_like(mu), v.clone()
        elif getattr(self, "use_graph_rollout", False):
            # one graph per rollout (same lifetime rule as the update graph: no eager allocations of
            # other phases between its replays)
            ac = GraphedActorCritic(policy, critic, N, obs_dim, dev, deterministic, anneal)
        else:
            def ac(s):
                return policy(s, deterministic=deterministic, anneal=anneal), critic(s).reshape(N)
        action, value = ac(state)
        eng = getattr(env, "eng", None)
        n_cut = torch.zeros(1, dtype=torch.int32, device=dev)
        for t in range(T):
            next_state, reward, done, _ = env.step(action)
            next_state = next_state.to(torch.float32)
            buf.store(state, action, reward.to(torch.float32), value)
            if eng is not None:                                   # one launch: traj_len, cut, flags, count
                eng.rollout_cuts(done.to(torch.uint8), traj_len, buf.flags[t], n_cut, max_traj_len, t == T - 1)
            else:
                traj_len += 1
                done = done.bool()
                cut = done | (traj_len >= max_traj_len) | (t == T - 1)
                buf.flags[t] = (cut.to(torch.uint8) * _abi.FLAG_LAST) | (done.to(torch.uint8) * _abi.FLAG_ABSORBING)
                traj_len = torch.where(cut, torch.zeros_like(traj_len), traj_len)
                n_cut = cut.sum().reshape(1)
            action, value = ac(next_state)                        # V(s_{t+1}): bootstrap now, and step t+1's pair
            buf.next_values[t] = value
            if t < T - 1 and int(n_cut.item()):
                cut = (buf.flags[t] & _abi.FLAG_LAST).bool()
                fresh = env.reset(env_mask=cut).to(torch.float32)
                next_state = torch.where(cut.unsqueeze(1), fresh, next_state)
                action, value = ac(next_state)                    # reset envs start from a new state
            state = next_state
        if hasattr(ac, "close"):
            ac.close()
        return buf

    # ------------------------------------------------------------------ losses
    def update_policy_fused(self, eng, obs_batch, action_batch, return_batch, advantage_batch,
                            mirror_observation=None, act_src=None, act_sign=None):
        """update_policy with the loss terms evaluated by oly_ppo_loss / oly_mirror_loss (mask = 1,
        diagonal Gaussian policy).  Same return tuple; clip_fraction is a python float."""
        policy, critic, old_policy = self.policy, self.critic, self.old_policy
        values = critic(obs_batch)
        pdf = policy.distribution(obs_batch)
        B, A = pdf.loc.shape
        with torch.no_grad():
            old = old_policy.distribution(obs_batch)
            old_mu, old_std = old.loc, _plain_std(old.scale, B, A)
        actor_loss, entropy_penalty, critic_loss, approx_kl_div, clipf = FusedPPOLoss.apply(
            eng, pdf.loc, _plain_std(pdf.scale, B, A), old_mu, old_std, action_batch, advantage_batch,
            return_batch, values, self.clip, self.vf_coeff)
        if mirror_observation is not None and act_src is not None:
            det = policy(obs_batch)
            mir = policy(mirror_observation(obs_batch))
            mirror_loss = FusedMirrorLoss.apply(eng, det, mir, act_src, act_sign)
        else:
            mirror_loss = torch.zeros(1, device=obs_batch.device)
        return actor_loss, entropy_penalty, critic_loss, approx_kl_div, mirror_loss, clipf

    def update_policy(self, obs_batch, action_batch, return_batch, advantage_batch, mask,
                      mirror_observation=None, mirror_action=None):
        """The same six quantities as update_policy_fused (rl/algos/ppo.py:232-282) evaluated with
        plain torch ops and autograd: the path for mirrors given as functions and for policies that
        are not diagonal Gaussians, and the checker the fused kernel is tested against."""
        new_pdf = self.policy.distribution(obs_batch)
        with torch.no_grad():
            old_pdf = self.old_policy.distribution(obs_batch)
            logp_old = old_pdf.log_prob(action_batch).sum(-1, keepdim=True)
        log_ratio = new_pdf.log_prob(action_batch).sum(-1, keepdim=True) - logp_old
        ratio = torch.exp(log_ratio)
        lo, hi = 1.0 - self.clip, 1.0 + self.clip
        weighted = advantage_batch * mask
        surrogate = torch.minimum(ratio * weighted, torch.clamp(ratio, lo, hi) * weighted)
        terms = dict(
            actor=-surrogate.mean(),
            entropy=-(new_pdf.entropy() * mask).mean(),
            critic=self.vf_coeff * F.mse_loss(return_batch, self.critic(obs_batch)),
            kl=((ratio - 1.0) - log_ratio).detach().mean(),                    # the k3 estimator
            clipped=((ratio.detach() - 1.0).abs() > self.clip).float().mean().item())
        if mirror_observation is None or mirror_action is None:
            mirror = torch.zeros(1, device=obs_batch.device)
        else:
            mirrored = mirror_action(self.policy(mirror_observation(obs_batch)))
            mirror = torch.square(self.policy(obs_batch) - mirrored).mean()
        return terms["actor"], terms["entropy"], terms["critic"], terms["kl"], mirror, terms["clipped"]

    # ------------------------------------------------------------------ training loop
    def train(self, env_fn, policy, critic, n_itr, anneal_rate=1.0, verbose=True):
        self.old_policy = deepcopy(policy)
        self.policy, self.critic = policy, critic
        use_graph = bool(getattr(self, "use_graph", False))
        # ONE optimiser arithmetic for the eager and the graph-replayed update: Adam(capturable=True) forms
        # its bias corrections with device tensors, capturable=False with python floats - last-bit
        # differences that Adam amplifies to 1e-3 within a few updates (profiles/r02/graph_drift)
        capturable = use_graph or next(policy.parameters()).is_cuda
        self.actor_optimizer = optim.Adam(policy.parameters(), lr=self.lr, eps=self.eps, capturable=capturable)
        self.critic_optimizer = optim.Adam(critic.parameters(), lr=self.lr, eps=self.eps, capturable=capturable)
        if getattr(self, "tuned_gemms", False) and next(policy.parameters()).is_cuda:
            # The update phase is PyTorch's.  hipBLASLt's default pick for the weight-gradient
            # GEMMs ([256, B] x [B, 256]) is a 32 x 64 macro tile; torch's TunableOp times the candidates once
            # per GEMM shape and keeps the fastest: 2.70 -> 1.92 ms per 65536-row update on MI355X.  Opt-in:
            # tuning costs a few seconds per process and changes the GEMMs' summation order.
            torch.cuda.tunable.enable(True)
            torch.cuda.tunable.tuning_enable(True)
            os.makedirs(self.save_path, exist_ok=True)
            torch.cuda.tunable.set_filename(os.path.join(self.save_path, "tunableop_results.csv"))
        env = env_fn()
        from . import dist as odist
        multi_rank = odist.is_dist() and torch.distributed.get_world_size() > 1
        if multi_rank:
            # one learner, replicated: same initial weights everywhere, gradients averaged before every step;
            # the advantage statistics are already global (PPORollout).  The graph-captured update holds no
            # collective, so multi-rank training takes the eager update path.
            odist.broadcast_parameters([policy, critic, self.old_policy])
            use_graph = False
        is_writer = (not multi_rank) or torch.distributed.get_rank() == 0     # ONE set of logs / checkpoints
        post = PPORollout(env.eng, gamma=self.gamma, lam=self.lam, eps=self.eps)
        obs_mirr = getattr(env, "mirror_clock_observation", None) if hasattr(env, "mirror_observation") else None
        act_mirr = getattr(env, "mirror_action", None)
        fused = bool(getattr(self, "fused_loss", True)) and hasattr(env, "eng")
        act_src = act_sign = None
        if fused and act_mirr is not None and hasattr(env, "_act_src"):
            act_src = env._act_src.to(env.eng.device, torch.int32)
            act_sign = env._act_sgn.to(env.eng.device, torch.float32)
        elif act_mirr is not None:
            fused = False                                   # mirror given as a function: unfused torch path
        # the update path: "kernel" (default where it applies: K14), else the fused-loss torch update (graph-replayed
        # with use_graph), else plain torch
        use_kernel = (fused and getattr(self, "update_kernel", True) and (act_mirr is None or act_src is not None)
                      and KernelUpdate.supports(policy, critic, True))
        kupd = None
        # The minibatch permutation: the reference's BatchSampler(SubsetRandomSampler) draws torch.randperm(n) from the
        # default CPU generator.  For config 3's 1.6 M rows that draw costs the host 25-35 ms per epoch, longer than the GPU
        # needs for the epoch's 25 kernel updates, so the kernel path draws the permutation on the device
        # (`device_permutation = False` keeps the host draw, e.g. to train seed-for-seed like the torch paths).
        device_perm = getattr(self, "device_permutation", None)
        device_perm = use_kernel if device_perm is None else bool(device_perm)
        T = max(1, self.batch_size // env.num_envs)
        curr_anneal, start = 1.0, time.time()
        history = []
        for itr in range(n_itr):
            self.iteration_count = itr
            if hasattr(env, "iteration_count"):
                env.iteration_count = itr
            t0 = time.time()
            if self.highest_reward > (2 / 3) * self.max_traj_len and curr_anneal > 0.5:
                curr_anneal *= anneal_rate
            buf = self.sample_vec(env, self.policy, self.critic, T, self.max_traj_len, anneal=curr_anneal)
            returns, advantages = post.finish(buf, normalize=True)       # finish_path + (adv-mean)/(std+eps)
            sample_s = time.time() - t0
            n = buf.T * buf.N
            observations = buf.states.reshape(n, -1)
            actions = buf.actions.reshape(n, -1)
            returns, advantages = returns.reshape(n, 1), advantages.reshape(n, 1)
            self.total_steps += n
            self.old_policy.load_state_dict(policy.state_dict())
            minibatch = self.minibatch_size or n
            n_batches = n // minibatch
            rows_all = float(minibatch)
            if multi_rank:
                # Convention: minibatch_size is PER RANK (every rank contributes `minibatch` rows to each of the
                # min-over-ranks n_batches updates, so the global minibatch is world x minibatch rows and a rank with a
                # larger shard drops its surplus tail rows of the epoch's permutation; to keep the reference's single-learner
                # batch, pass minibatch_size / world).  Shards may differ in size (shard_range hands out near-equal
                # ranges): every rank must enter the same number of gradient all-reduces, and a rank's gradient counts
                # with its minibatch's rows
                n_batches = odist.allreduce_min(n_batches)
                rows_all = odist.allreduce_sum([minibatch])[0]
            if n_batches == 0:
                # np.mean over an empty loss list would put NaN into the history and the epoch loop would silently do nothing
                raise ValueError(f"PPO.train: minibatch_size {minibatch} exceeds the {n} samples of "
                                 f"{'the smallest rank shard' if multi_rank else 'one iteration'}: no update would run")
            t1 = time.time()
            stats = []
            graphed = None
            kernel = None
            if use_kernel:
                # K14: forward + losses + backward of a minibatch in one launch; gradient clipping and Adam stay torch's
                if kupd is None:
                    kupd = KernelUpdate(env.eng, policy, critic, self.old_policy, self.clip, self.vf_coeff,
                                        self.mirror_coeff, obs_mirr, act_src, act_sign, lr=self.lr, eps=self.eps,
                                        max_grad_norm=self.grad_clip)
                kernel = self.kupd = kupd
                kernel.begin(observations, buf.mu.reshape(n, -1) if getattr(buf, "mu", None) is not None
                             and getattr(buf, "mu_from_fused_forward", False) else None)
                kstats = torch.zeros((self.epochs * max(n_batches, 1), 6), dtype=torch.float64, device=observations.device)
                adv_flat, ret_flat = advantages.reshape(-1).contiguous(), returns.reshape(-1).contiguous()
            elif use_graph and fused and self.target_kl is None:   # (a replayed graph contains the optimiser steps: no KL check)
                # Captured anew for every iteration's update phase (~15 ms).  profiles/r02/graph_drift/README.md:
                # a captured torch update replays bit-exactly until a [synchronize -> kernel write into a newly
                # allocated block of >= 1 MB] happens between two replays; after that the multi-block
                # reductions of the first-layer bias gradients come back wrong (no tensor visible from python
                # is overwritten: the victim is internal to the captured graph / the runtime).  A rollout does
                # exactly that between two update phases, the minibatch loop below does not (index_select into
                # static buffers, replay, tiny clones, a host-drawn permutation copied H2D), so the graph's
                # lifetime is one update phase.
                graphed = GraphedUpdate(self, env.eng, minibatch, observations.shape[1], actions.shape[1],
                                        obs_mirr, act_src, act_sign)
            continue_training = True                           # False once 1.5 * target_kl is breached (ppo.py:343, 391-393)

            def kl_breached(kl):
                """ppo.py:391: stop the update phase BEFORE this minibatch's optimiser steps when its approximate KL exceeds
                1.5 target_kl (over ranks: their mean, so that the replicas of the one learner stop together)."""
                if self.target_kl is None:
                    return False
                kl = float(kl)
                if multi_rank:
                    kl = odist.allreduce_sum([kl])[0] / torch.distributed.get_world_size()
                if kl > 1.5 * self.target_kl:
                    if verbose:
                        print(f"Early stopping at step {epoch} due to reaching max kl: {kl:.2f}")
                    return True
                return False
            for epoch in range(self.epochs):
                # BatchSampler(SubsetRandomSampler(range(n)), minibatch, drop_last=True) draws ONE
                # torch.randperm(n) from the default CPU generator and cuts it into consecutive
                # batches; the same permutation is cut on the device here (no per-index Python loop)
                # Drawing 1.6 M indices takes the host ~20 ms, as long as the GPU needs for a whole epoch of kernel updates: the
                # permutation of the NEXT epoch (or of the next iteration's first one) is drawn right after this epoch's
                # minibatches have been queued, while the GPU works through them.  Same draws in the same order.
                if device_perm:
                    perm = torch.randperm(n, device=observations.device).to(torch.int32)
                else:
                    perm_host = self._next_perm if getattr(self, "_next_perm", None) is not None and len(self._next_perm) == n \
                        else torch.randperm(n)
                    self._next_perm = None
                    perm = (perm_host.to(torch.int32) if kernel is not None else perm_host).to(observations.device)
                one_call = kernel is not None and not multi_rank and self.target_kl is None
                if one_call:
                    # the whole minibatch loop of the epoch in one C call (the per-minibatch KL check below needs the host)
                    rows = kstats[len(stats):len(stats) + n_batches]
                    kernel.epoch(observations, actions, ret_flat, adv_flat, perm, minibatch, n_batches, rows)
                    stats.extend(rows.unbind(0))
                for b in (() if one_call else range(n_batches)):
                    idx = perm[b * minibatch:(b + 1) * minibatch]
                    if kernel is not None:
                        slot = kstats[len(stats)]
                        kernel.grads(observations, actions, ret_flat, adv_flat, idx, slot)
                        stats.append(slot)
                        if kl_breached(slot[3]):
                            continue_training = False
                            break
                        if multi_rank:
                            odist.allreduce_flat([kernel.grad_actor, kernel.grad_critic], weight=minibatch, total_weight=rows_all)
                            kernel.apply(norm_ready=False)      # the norm of the REDUCED gradients
                        else:
                            kernel.apply()
                        continue
                    if graphed is not None:
                        every = getattr(self, "graph_recapture_every", None)     # test hook: fresh graph every k replays
                        if every and len(stats) and len(stats) % every == 0:
                            old, graphed = graphed, GraphedUpdate(self, env.eng, minibatch, observations.shape[1],
                                                                  actions.shape[1], obs_mirr, act_src, act_sign)
                            old.close()
                        stats.append(graphed(observations, actions, returns, advantages, idx).clone())
                        continue
                    if fused:
                        a_l, ent, c_l, kl, m_l, clipf = self.update_policy_fused(
                            env.eng, observations[idx], actions[idx], returns[idx], advantages[idx], obs_mirr,
                            act_src, act_sign)
                    else:
                        a_l, ent, c_l, kl, m_l, clipf = self.update_policy(
                            observations[idx], actions[idx], returns[idx], advantages[idx], 1, obs_mirr, act_mirr)
                    if kl_breached(kl):
                        stats.append((a_l.item(), ent.item(), c_l.item(), kl.item(), float(m_l), float(clipf)))
                        continue_training = False
                        break
                    self.actor_optimizer.zero_grad()
                    self.critic_optimizer.zero_grad()
                    if fused:
                        # one graph node carries all terms; actor and critic share no parameters, so a
                        # single backward gives each network exactly its own loss' gradient
                        (a_l + self.mirror_coeff * m_l + self.ent_coeff * ent + c_l).sum().backward()
                    else:
                        (a_l + self.mirror_coeff * m_l + self.ent_coeff * ent).sum().backward()
                        c_l.backward()
                    if multi_rank:
                        odist.allreduce_gradients(list(policy.parameters()) + list(critic.parameters()),
                                                  weight=minibatch, total_weight=rows_all)
                    torch.nn.utils.clip_grad_norm_(policy.parameters(), self.grad_clip)
                    self.actor_optimizer.step()
                    torch.nn.utils.clip_grad_norm_(critic.parameters(), self.grad_clip)
                    self.critic_optimizer.step()
                    stats.append((a_l.item(), ent.item(), c_l.item(), kl.item(), float(m_l), float(clipf)))
                if not continue_training:
                    break
                if kernel is not None and not device_perm and (epoch + 1 < self.epochs or itr + 1 < n_itr):
                    self._next_perm = torch.randperm(n)            # behind the queued minibatches
            if stats and torch.is_tensor(stats[0]):
                stats = torch.stack(stats).cpu().tolist()      # one device->host copy per iteration
            if graphed is not None:
                for p_ in list(policy.parameters()) + list(critic.parameters()):
                    p_.grad = None                             # gradients live in the graph's pool
                graphed.close()
            del graphed
            if hasattr(buf, "episode_sums"):
                mean_ret, mean_len = _global_episode_means(None, None, multi_rank, buf.episode_sums().cpu().tolist())
            else:
                ep_ret, ep_len = buf.episode_stats()
                mean_ret, mean_len = _global_episode_means(ep_ret, ep_len, multi_rank)
            if is_writer:
                with open(self.train_fn, "a") as out:
                    out.write("{},{}\n".format(mean_ret, mean_len))
            eval_rec = {}
            if (itr + 1) % self.eval_freq == 0:               # deterministic evaluation + checkpoints (:443-477)
                t2 = time.time()
                test = self.sample_vec(env, self.policy, self.critic, T, self.max_traj_len, deterministic=True)
                t_ret, t_len = test.episode_stats()
                # the evaluation return over ALL ranks' episodes: highest_reward drives the exploration anneal above,
                # which must not diverge between the replicas of the one learner
                avg_eval_reward, avg_eval_len = _global_episode_means(t_ret, t_len, multi_rank)
                if is_writer:
                    with open(self.eval_fn, "a") as out:
                        out.write("{},{}\n".format(avg_eval_reward, avg_eval_len))
                    self.save(policy, critic, "_" + repr(itr))
                if self.highest_reward < avg_eval_reward:
                    self.highest_reward = avg_eval_reward
                    if is_writer:
                        self.save(policy, critic)
                eval_rec = dict(eval_return=avg_eval_reward, eval_s=time.time() - t2)
                if verbose:
                    print("====EVALUATE EPISODE====  (Return = {})".format(avg_eval_reward))
            rec = dict(itr=itr, ep_return=mean_ret, ep_len=mean_len,
                       sample_s=sample_s, optim_s=time.time() - t1,
                       fps=self.total_steps / (time.time() - start), losses=np.mean(stats, axis=0).tolist(), **eval_rec)
            history.append(rec)
            if verbose:
                print("itr {itr}: return {ep_return:.3f} len {ep_len:.1f} sampling {sample_s:.2f}s "
                      "optimizer {optim_s:.2f}s fps {fps:.0f}".format(**rec))
        return history
