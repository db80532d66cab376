"""Discriminator reward of GAIL / VAIL with the reference's semantics.

  Standardizer                 imitation_lib/utils/networks.py:48-81   -> DeviceStandardizer
  VariationalNet.forward       networks.py:258-284                     -> VariationalDiscriminator
  GAIL.make_discrim_reward     imitation_lib/imitation/gail_TRPO.py:320-327
  prepare_discrim_inputs       gail_TRPO.py:297-313 (state mask)

The MLP GEMMs (32->256->128->(128,128)->1 for UnitreeH1, examples/imitation_learning/
utils.py:151-161) run in PyTorch-ROCm (MFMA); mask + standardise, the reparameterisation and
the reward epilogue are HIP kernels; the statistics live on the device (no CPU bounce as in
networks.py:70).  The reparameterisation noise is an INPUT so results are reproducible.
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

    @torch.no_grad()
    def logits(self, x, eps):
        xs = self.stand.forward(x, self.mask)
        mu, logvar = self.net.encode(xs)
        z = self.eng.disc_reparam(mu.contiguous(), logvar.contiguous(), eps)
        return self.net.decoder(z).reshape(-1).contiguous(), mu, logvar

    @torch.no_grad()
    def __call__(self, x, eps=None, generator=None):
        if eps is None:
            eps = torch.randn((x.shape[0], self.net.mu_out.out_features), dtype=torch.float32,
                              device=x.device, generator=generator)
        d, _, _ = self.logits(x, eps)
        return self.eng.disc_reward(d)
