"""Fused forward of the rollout's actor and critic MLPs (K11): one launch per vec step instead of the
modules' six GEMM + four ReLU launches.

Reference: Gaussian_FF_Actor._get_dist_params (rl/policies/actor.py:180-195) and FF_V.forward
(rl/policies/critic.py:62-74), called once per env.step in PPO.sample (rl/algos/ppo.py:181-182).
Supported structure (the reference's PPO setting): two hidden Linear layers of 256 with relu, a
linear head, a fixed (state-independent) standard deviation, unbounded mean.  Anything else keeps
the modules' own torch forward (vecstep.TorchForward)."""
import torch
import torch.nn.functional as F

from ._ffi import OlyError


def _relu_mlp_parts(layers, head, nonlinearity):
    layers = list(layers)
    if len(layers) != 2 or not all(isinstance(m, torch.nn.Linear) for m in layers + [head]):
        return None
    if nonlinearity not in (None, F.relu, torch.relu):
        return None
    if layers[0].out_features != 256 or layers[1].in_features != 256 or layers[1].out_features != 256:
        return None
    if head.in_features != 256 or head.out_features > 32 or layers[0].in_features > 64:
        return None
    return layers[0], layers[1], head


def _vec(v, n, device):
    """obs_mean / obs_std as the modules keep them (python float, numpy array or tensor) -> f32 [n]."""
    t = torch.as_tensor(v, dtype=torch.float32, device=device).reshape(-1)
    return t.expand(n).contiguous() if t.numel() == 1 else t.contiguous()


class FusedMLPForward:
    """forward(state) -> (mu [N,A], value [N]) with oly_mlp_forward2; `refresh()` re-packs the weights
    (call it after every optimiser phase: the packed stream is a copy)."""

    def __init__(self, eng, policy, critic):
        self.eng, self.policy, self.critic = eng, policy, critic
        pa = _relu_mlp_parts(getattr(policy, "actor_layers", ()), getattr(policy, "means", None),
                             getattr(policy, "nonlinearity", None))
        pc = _relu_mlp_parts(getattr(critic, "critic_layers", ()), getattr(critic, "network_out", None),
                             getattr(critic, "nonlinearity", None))
        if pa is None or pc is None or getattr(policy, "bounded", False) or getattr(policy, "learn_std", False):
            raise OlyError("FusedMLPForward: the policy / critic are not 2 x 256 relu MLPs with a fixed std")
        if pa[0].in_features != pc[0].in_features or pc[2].out_features != 1:
            raise OlyError("FusedMLPForward: actor and critic must read the same observation; critic head must be 1")
        self.pa, self.pc = pa, pc
        self.in_dim, self.act_dim = pa[0].in_features, pa[2].out_features
        self.packed_a = self.packed_c = None
        self._out = {}
        self.refresh()

    @staticmethod
    def supports(policy, critic):
        try:
            pa = _relu_mlp_parts(getattr(policy, "actor_layers", ()), getattr(policy, "means", None),
                                 getattr(policy, "nonlinearity", None))
            pc = _relu_mlp_parts(getattr(critic, "critic_layers", ()), getattr(critic, "network_out", None),
                                 getattr(critic, "nonlinearity", None))
        except (AttributeError, TypeError):
            return False
        return (pa is not None and pc is not None and not getattr(policy, "bounded", False)
                and not getattr(policy, "learn_std", False) and pc[2].out_features == 1
                and pa[0].in_features == pc[0].in_features)

    @torch.no_grad()
    def refresh(self):
        dev = self.eng.device
        # actor: always normalises its input (actor.py:189); critic: only in eval mode (critic.py:63-64)
        a_mean = _vec(getattr(self.policy, "obs_mean", 0.0), self.in_dim, dev)
        a_std = _vec(getattr(self.policy, "obs_std", 1.0), self.in_dim, dev)
        self.norm_a = True
        self.norm_c = (not self.critic.training) and getattr(self.critic, "obs_mean", None) is not None
        c_mean = _vec(self.critic.obs_mean, self.in_dim, dev) if self.norm_c else None
        c_std = _vec(self.critic.obs_std, self.in_dim, dev) if self.norm_c else None

        def params(parts):
            out = []
            for lin in parts:
                out += [lin.weight.detach().to(torch.float32).contiguous(), lin.bias.detach().to(torch.float32).contiguous()]
            return out
        self._norm = ((a_mean, a_std), (c_mean, c_std))
        self.packed_a = self.eng.mlp_pack(*params(self.pa), a_mean, a_std, packed=self.packed_a)
        self.packed_c = self.eng.mlp_pack(*params(self.pc), c_mean, c_std, packed=self.packed_c)

    def norm_tables(self):
        """((actor mean, std), (critic mean, std)) as the last refresh() packed them (None where not normalised)."""
        return self._norm

    def std(self, state, act_dim):
        sd = getattr(self.policy, "fixed_std", None)
        if sd is None:
            raise OlyError("FusedMLPForward needs a policy with fixed_std")
        t = torch.as_tensor(sd, dtype=torch.float32, device=state.device).reshape(-1)
        return t.expand(act_dim).contiguous() if t.numel() == 1 else t.contiguous()

    def outputs(self, N):
        """(mu [N,A], value [N]) static output tensors: the same addresses every step (graph replay)."""
        if N not in self._out:
            dev = self.eng.device
            self._out[N] = (torch.empty((N, self.act_dim), dtype=torch.float32, device=dev),
                            torch.empty((N, 1), dtype=torch.float32, device=dev))
        mu, v = self._out[N]
        return mu, v.view(N)

    def __call__(self, state):
        N = int(state.shape[0])
        mu, v = self.outputs(N)
        self.eng.mlp_forward2(state, self.packed_a, self.act_dim, mu, self.packed_c, 1, v.view(N, 1), self.norm_a,
                              self.norm_c)
        return mu, v

    def value(self, x):
        """critic alone on [M, in] rows (the bootstrap side list)."""
        out = torch.empty((int(x.shape[0]), 1), dtype=torch.float32, device=x.device)
        self.eng.mlp_forward2(x.contiguous(), self.packed_c, 1, out, normalize_a=self.norm_c)
        return out.view(-1)
