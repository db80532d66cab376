"""Discriminator reward of GAIL / VAIL with the reference's semantics.

  Standardizer                 imitation_lib/utils/networks.py:48-81   -> DeviceStandardizer
  VariationalNet.forward       networks.py:258-284                     -> VariationalDiscriminator
  GAIL.make_discrim_reward     imitation_lib/imitation/gail_TRPO.py:320-327
  prepare_discrim_inputs       gail_TRPO.py:297-313 (state mask)

The reward path (mask + standardise, 32->256->128->(128,128)->1 for UnitreeH1, examples/
imitation_learning/utils.py:151-161, reparameterisation, reward formula) is ONE launch on the f32
matrix cores (K12, oly_disc_forward) after the statistics update (oly_col_stats); the statistics
live on the device (no CPU bounce as in networks.py:70).  The discriminator's TRAINING forward /
backward stays in PyTorch-ROCm.  The reparameterisation noise is an INPUT so results are
reproducible.
"""
import numpy as np
import torch
import torch.nn as nn


class DeviceStandardizer:
    """Running (count, sum, sumsq) per column on the device; mean/std as the reference
    derives them (_sum=0, _sumsq=1e-2, _count=1e-2, variance floor 1e-2)."""

    def __init__(self, engine, dim):
        self.eng, self.dim = engine, dim
        self.colstats = torch.zeros((3, dim), dtype=torch.float64, device=engine.device)
        self._fresh = True

    def update_mean_std(self, x):
        self.colstats = self.eng.col_stats(x, None if self._fresh else self.colstats)
        self._fresh = False

    @property
    def mean(self):
        return self.colstats[1] / (self.colstats[0] + 1e-2)

    @property
    def std(self):
        cnt = self.colstats[0] + 1e-2
        mean = self.colstats[1] / cnt
        return torch.sqrt(torch.clamp((self.colstats[2] + 1e-2) / cnt - mean * mean, min=1e-2))

    def forward(self, x, mask=None):
        """Updates the statistics with x (as Standardizer.forward does on EVERY call), then
        returns the masked, standardised float32 batch."""
        xm = x if mask is None else x[:, mask.long()].contiguous()
        self.update_mean_std(xm)
        return self.eng.disc_standardize(x, mask, self.mean.contiguous(), self.std.contiguous())


class VariationalDiscriminator(nn.Module):
    """encoder -> (mu, logvar) -> z = mu + exp(logvar/2) eps -> decoder."""

    def __init__(self, in_dim=32, enc_features=(256,), enc_out=128, z_size=128):
        super().__init__()
        dims = [in_dim] + list(enc_features) + [enc_out]
        self.encoder = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.mu_out = nn.Linear(enc_out, z_size)
        self.logvar_out = nn.Linear(enc_out, z_size)
        self.decoder = nn.Linear(z_size, 1)

    def load_reference_arrays(self, g):
        """Weights in the layout gen_golden.py saves (state_dict of the reference network)."""
        with torch.no_grad():
            for lin, w, b in ((self.encoder[0], "enc_w0", "enc_b0"), (self.encoder[1], "enc_w1", "enc_b1"),
                              (self.mu_out, "mu_w", "mu_b"), (self.logvar_out, "lv_w", "lv_b"),
                              (self.decoder, "dec_w", "dec_b")):
                lin.weight.copy_(torch.as_tensor(np.asarray(g[w])))
                lin.bias.copy_(torch.as_tensor(np.asarray(g[b])))
        return self

    def encode(self, xs):
        h = xs
        for lin in self.encoder:
            h = torch.relu(lin(h))
        return self.mu_out(h), self.logvar_out(h)


class DiscriminatorReward:
    """make_discrim_reward for a batch of observations on the device."""

    def __init__(self, engine, net, state_mask=None, standardizer=None):
        self.eng, self.net = engine, net
        self.mask = None if state_mask is None else torch.as_tensor(np.asarray(state_mask, dtype=np.int32),
                                                                   device=engine.device)
        dim = net.encoder[0].in_features
        self.stand = standardizer or DeviceStandardizer(engine, dim)
        self._packed, self._packed_ver = None, None
        self._identity_mask = state_mask is not None and np.array_equal(np.asarray(state_mask), np.arange(dim))

    # ---- fused path (K12)
    def _params(self):
        n = self.net
        return [n.encoder[0].weight, n.encoder[0].bias, n.encoder[1].weight, n.encoder[1].bias, n.mu_out.weight,
                n.mu_out.bias, n.logvar_out.weight, n.logvar_out.bias, n.decoder.weight, n.decoder.bias]

    @property
    def fused(self):
        n = self.net
        return (len(n.encoder) == 2 and n.encoder[0].in_features <= 64 and n.encoder[0].out_features == 256
                and n.encoder[1].out_features == 128 and n.mu_out.out_features == 128)

    def packed(self):
        """The MFMA operand stream of the CURRENT weights: re-packed on every call (one ~6 us launch into the same
        buffer), as FusedMLPForward.refresh() does per rollout.  A cache keyed on (data_ptr, _version) would miss writes
        through `.data` (dist.broadcast_parameters, load paths that copy into p.data): `_version` does not move for
        those, and the reward would silently keep the old weights.  `cache_packed = True` opts into that cache for
        callers that own every write to the parameters (benchmarks of a frozen network)."""
        ps = self._params()
        if getattr(self, "cache_packed", False):
            ver = tuple((p.data_ptr(), p._version) for p in ps)
            if ver == self._packed_ver and self._packed is not None:
                return self._packed
            self._packed_ver = ver
        self._packed = self.eng.disc_pack(*[p.detach().to(torch.float32).contiguous() for p in ps], packed=self._packed)
        return self._packed

    def invalidate(self):
        """Forget the cached stream (only meaningful with cache_packed)."""
        self._packed_ver = None

    def _update_statistics(self, x):
        """Standardizer.forward's update_mean_std on the masked batch (networks.py:70,76-81)."""
        whole = self.mask is None or (self._identity_mask and x.shape[1] == self.mask.numel())
        self.stand.update_mean_std(x if whole else x[:, self.mask.long()].contiguous())
        return None if whole else self.mask

    @torch.no_grad()
    def forward(self, x, eps, want=("reward",), out=None):
        """Statistics update + oly_disc_forward: any of reward / logits / mu / logvar for x [B,Dx]."""
        whole = self.mask is None or (self._identity_mask and x.shape[1] == self.mask.numel())
        if whole and isinstance(self.stand, DeviceStandardizer):
            # statistics update + forward issued by ONE C call (three launches): from Python the separate calls are
            # host-bound at the reference's batch (B = 4096: 32 us against 21)
            st = self.stand
            ps = [p.detach() for p in self._params()]
            direct = all(p.dtype == torch.float32 and p.is_contiguous() for p in ps)
            if self._packed is None or not direct or getattr(self, "cache_packed", False):
                packed, ps = self.packed(), None                      # separate pack call (or the opt-in cache)
            else:
                packed = self._packed                                # re-packed from the live parameters inside the call
            o = self.eng.disc_reward_step(x, packed, st.colstats, not st._fresh, eps=eps, want=want, out=out, weights=ps)
            st._fresh = False
            return o
        mask = self._update_statistics(x)
        return self.eng.disc_forward(x, self.packed(), mask=mask, colstats=self.stand.colstats, eps=eps, want=want,
                                     out=out)

    def prepared(self, x, eps, want=("reward",), out=None):
        """`step()` = forward(x, eps, want, out) on these same tensors with every argument validated once: the loop a
        reward evaluation over a fixed rollout block runs (whole rows, fused network).  The weights are re-packed from
        the live parameters inside every call."""
        whole = self.mask is None or (self._identity_mask and x.shape[1] == self.mask.numel())
        ps = [p.detach() for p in self._params()]
        if not (whole and self.fused and isinstance(self.stand, DeviceStandardizer)
                and all(p.dtype == torch.float32 and p.is_contiguous() for p in ps)):
            return lambda: self.forward(x, eps, want=want, out=out)
        st = self.stand
        if self._packed is None:
            self.packed()
        launch = self.eng.disc_reward_step(x, self._packed, st.colstats, None, eps=eps, want=want, out=out,
                                           weights=None if getattr(self, "cache_packed", False) else ps)

        def step():
            o = launch(not st._fresh)
            st._fresh = False
            return o
        return step

    # ---- layer-by-layer path (PyTorch GEMMs + K8 kernels): other network shapes, and the cross-check
    @torch.no_grad()
    def logits_unfused(self, x, eps):
        xs = self.stand.forward(x, self.mask)
        mu, logvar = self.net.encode(xs)
        z = self.eng.disc_reparam(mu.contiguous(), logvar.contiguous(), eps)
        return self.net.decoder(z).reshape(-1).contiguous(), mu, logvar

    @torch.no_grad()
    def logits(self, x, eps):
        if not self.fused:
            return self.logits_unfused(x, eps)
        o = self.forward(x, eps, want=("logits", "mu", "logvar"))
        return o["logits"], o["mu"], o["logvar"]

    @torch.no_grad()
    def __call__(self, x, eps=None, generator=None):
        if eps is None:
            eps = torch.randn((x.shape[0], self.net.mu_out.out_features), dtype=torch.float32,
                              device=x.device, generator=generator)
        if self.fused:
            return self.forward(x, eps)["reward"]
        d, _, _ = self.logits_unfused(x, eps)
        return self.eng.disc_reward(d)


class GAILAdvantage:
    """The reward / advantage half of GAIL.fit (gail_TRPO.py:105-129) on the device:

        r_disc = make_discrim_reward(x, u, xn)
        r      = r * env_reward_frac + r_disc * (1 - env_reward_frac)
        v_target, adv = compute_gae(V, x, xn, r, absorbing, last, gamma, lam)
        adv    = (adv - mean(adv)) / (std(adv) + 1e-8)            (numpy, biased std)

    x / xn are [T,N,obs] rollout blocks (the reference's flat dataset is the N = 1 case)."""

    def __init__(self, engine, disc_reward, critic, gamma=0.99, lam=0.97, env_reward_frac=0.0):
        from .rollout import GAERollout
        assert 0.0 <= env_reward_frac <= 1.0, "Environment reward must be between [0,1]"
        self.eng, self.disc, self.critic = engine, disc_reward, critic
        self.frac = env_reward_frac
        self.post = GAERollout(engine, gamma=gamma, lam=lam)

    @torch.no_grad()
    def __call__(self, x, xn, r_env, absorbing, last, eps=None):
        from . import _abi
        from .rollout import RolloutBuffer
        T, N, D = x.shape
        flat = x.reshape(T * N, D).contiguous()
        if self.frac < 1.0:
            r_disc = self.disc(flat, eps).reshape(T, N)
            r = r_env * self.frac + r_disc * (1 - self.frac)
        else:
            r = r_env
        buf = RolloutBuffer(T, N, D, 1, x.device)
        buf.rewards.copy_(r)
        buf.values.copy_(self.critic(flat).reshape(T, N))
        buf.next_values.copy_(self.critic(xn.reshape(T * N, D)).reshape(T, N))
        buf.flags.copy_((last.to(torch.uint8) * _abi.FLAG_LAST) | (absorbing.to(torch.uint8) * _abi.FLAG_ABSORBING))
        buf.ptr = T
        v_target, adv = self.post.finish(buf, normalize=True)
        return r, v_target, adv


# ------------------------------------------------------------------------------ discriminator fitting
def logit_bernoulli_entropy(logits):
    """(1 - sigmoid(x)) * x - logsigmoid(x)   imitation_lib/utils/math.py:34-39."""
    return (1.0 - torch.sigmoid(logits)) * logits - torch.nn.functional.logsigmoid(logits)


def gail_discriminator_loss(logits, target, entcoeff=1e-3):
    """GailDiscriminatorLoss.forward (imitation_lib/utils/math.py:24-32):
    mean(max(x,0) - x*z + log(1 + exp(-|x|))) - entcoeff * mean(bernoulli entropy)."""
    bce = torch.maximum(logits, torch.zeros_like(logits)) - logits * target + torch.log(1 + torch.exp(-torch.abs(logits)))
    return torch.mean(bce) - entcoeff * torch.mean(logit_bernoulli_entropy(logits))


class VDBLoss:
    """Variational-discriminator-bottleneck loss with the dual variable beta updated on every
    call (imitation_lib/utils/math.py:42-90): bce + beta * (mean KL - I_c) [+ entropy]."""

    def __init__(self, info_constraint, lr_beta, use_bernoulli_ent=False, entcoeff=1e-3):
        self._info_constr, self._lr_beta = info_constraint, lr_beta
        self._use_bernoulli_ent, self.entcoeff = use_bernoulli_ent, entcoeff
        self._beta = 0.1

    @staticmethod
    def kl_divergence(mu, logvar):
        return 0.5 * torch.sum(torch.pow(mu, 2) + torch.exp(logvar) - logvar - 1, dim=1)

    def __call__(self, inputs, target):
        logits, mu, logvar = inputs
        bottleneck = self.kl_divergence(mu, logvar).mean() - self._info_constr
        bce = torch.nn.functional.binary_cross_entropy_with_logits(torch.squeeze(logits), torch.squeeze(target))
        ent = logit_bernoulli_entropy(logits) if self._use_bernoulli_ent else torch.zeros_like(bce)
        loss = bce + self._beta * bottleneck + ent
        with torch.no_grad():                                   # dual ascent, clipped at 0 (:80-82)
            self._beta = max(0, self._beta + self._lr_beta * bottleneck)
        return loss


class ExpertDataset:
    """The demonstrations of GAIL / VAIL resident on the device.

    Trajectory.create_dataset (utils/trajectory.py:129-193) returns states / next_states as slices of
    the flattened trajectory table; here the table is uploaded ONCE (oly_traj_upload, the same copy
    K4's reset / next-sample kernels read) and the dataset is never materialised: a minibatch of the
    discriminator (gail_TRPO.py:176-202, demo_obs = states[indices][:, state_mask].astype(np.float32))
    is one row gather on the device by caller-drawn indices."""

    def __init__(self, engine, trajectory, ignore_keys=("q_pelvis_tx", "q_pelvis_tz"), state_mask=None):
        self.eng = engine
        if engine.traj_shape != tuple(trajectory.table.shape):
            engine.traj_upload(trajectory.table)
        keys = list(trajectory.keys)
        kept = [i for i, k in enumerate(keys) if k not in set(ignore_keys or ())]
        mask = np.arange(len(kept)) if state_mask is None else np.asarray(state_mask, dtype=np.int64)
        self.dataset_cols = torch.as_tensor(np.asarray(kept, dtype=np.int32), device=engine.device)
        self.cols = torch.as_tensor(np.asarray(kept, dtype=np.int32)[mask], device=engine.device)   # mask folded in
        self.rows = engine.expert_rows()

    def arrays(self):
        """states / next_states / absorbing / last exactly as create_dataset returns them (float64)."""
        return self.eng.expert_dataset(self.dataset_cols)

    def shuffled_indices(self, batch, rs=None):
        """The first batch of mushroom's minibatch_generator (np.random.shuffle of arange(rows), first
        `batch` entries): the reference's draw, made on the host, handed over as an input."""
        idx = np.arange(self.rows)
        (rs if rs is not None else np.random).shuffle(idx)
        return torch.as_tensor(idx[:batch].astype(np.int64), device=self.eng.device)

    def minibatch(self, idx, want_next=False):
        return self.eng.expert_gather(idx, self.cols, want_next=want_next)


class DiscriminatorTrainer:
    """The discriminator half of GAIL._fit_discriminator (gail_TRPO.py:167-218), states-only
    input as in the UnitreeH1 configuration: per epoch draw as many demonstration states as
    policy states, update the standardiser with the concatenated batch, targets 0 (policy) / 1
    (demonstrations) or the noisy variants, one optimiser step.  Random numbers (minibatch
    indices, noisy targets, VAIL eps) come from the caller's torch.Generator.

    mushroom's TorchApproximator.fit (absent) drives the optimiser in the reference; one Adam
    step per epoch on the whole concatenated batch is this class's stated reading of it."""

    def __init__(self, reward: "DiscriminatorReward", demo_states, loss, lr=5e-5, weight_decay=1e-3,
                 n_epochs=1, use_noisy_targets=False, variational=True):
        self.r, self.loss, self.n_epochs = reward, loss, n_epochs
        dev = reward.eng.device
        # an ExpertDataset serves minibatches straight from the device-resident trajectory table (its
        # columns already carry the state mask); an array is the dataset's `states` uploaded as is
        self.expert = demo_states if isinstance(demo_states, ExpertDataset) else None
        self.demo = None if self.expert is not None else torch.as_tensor(np.asarray(demo_states), dtype=torch.float32,
                                                                         device=dev)
        self.noisy, self.variational = use_noisy_targets, variational
        self.opt = torch.optim.Adam(reward.net.parameters(), lr=lr, weight_decay=weight_decay)

    def fit(self, plcy_obs, generator=None):
        r, dev = self.r, self.r.eng.device
        n = plcy_obs.shape[0]
        losses = []
        for _ in range(self.n_epochs):
            if self.expert is not None:
                idx = torch.randint(0, self.expert.rows, (n,), device=dev, generator=generator)
                plcy = plcy_obs.to(torch.float32)
                plcy = plcy if r.mask is None else plcy[:, r.mask.long()]
                x = torch.cat([plcy, self.expert.minibatch(idx)]).contiguous()
                xs = r.stand.forward(x, None)                    # updates the running statistics
            else:
                idx = torch.randint(0, self.demo.shape[0], (n,), device=dev, generator=generator)
                x = torch.cat([plcy_obs.to(torch.float32), self.demo[idx]]).contiguous()
                xs = r.stand.forward(x, r.mask)                  # updates the running statistics
            if self.noisy:
                demo_t = torch.empty((n, 1), device=dev).uniform_(0.80, 0.99, generator=generator)
                plcy_t = torch.empty((n, 1), device=dev).uniform_(0.01, 0.10, generator=generator)
            else:
                plcy_t, demo_t = torch.zeros((n, 1), device=dev), torch.ones((n, 1), device=dev)
            target = torch.cat([plcy_t, demo_t])
            mu, logvar = r.net.encode(xs)
            if self.variational:
                eps = torch.randn(mu.shape, device=dev, generator=generator)
                z = mu + torch.exp(logvar / 2) * eps
                loss = self.loss((r.net.decoder(z), mu, logvar), target)
            else:
                loss = self.loss(r.net.decoder(mu), target)
            self.opt.zero_grad()
            loss.backward()
            self.opt.step()
            losses.append(float(loss.detach()))
        return losses
