"""K14's oracle twin (oracle/oly_oracle.c: oly_ppo_update_cpu) against the reference's update step.

Pins: the six scalars of PPO.update_policy against `ppo_update.npz` (the reference's own update_policy executed on
its Gaussian_FF_Actor / FF_V weights: tests/golden/gen_golden.py), and the parameter gradients against torch
autograd of the same losses (what the reference's two backward() calls compute, rl/algos/ppo.py:396-410).  The
fixture holds no gradients, so autograd run here IS the reference for them."""
import numpy as np
import pytest

from helpers import ppo_update_arrays, ppo_update_case, torch_ppo_update_grads

# float32 GEMM summation order (torch's blocked GEMMs vs one k-ordered fma chain): elementwise relative to the
# largest entry of the same parameter tensor
GRAD_RTOL = 2e-5


def split_grads(flat, in_dim, out_dim):
    sizes = [256 * in_dim, 256, 65536, 256, out_dim * 256, out_dim]
    off = np.concatenate([[0], np.cumsum(sizes)])
    return [flat[off[i]:off[i + 1]] for i in range(6)]


def assert_grads_close(got, want, in_dim, out_dim, rtol=GRAD_RTOL):
    for i, (a, b) in enumerate(zip(split_grads(got, in_dim, out_dim), split_grads(want, in_dim, out_dim))):
        scale = np.abs(b).max()
        assert np.abs(a - b).max() <= rtol * scale + 1e-12, (i, np.abs(a - b).max(), scale)


def test_ppo_update_oracle_on_the_reference_fixture(golden, oracle):
    g = golden("ppo_update.npz")
    a = ppo_update_arrays(g)
    wb = lambda tag, layers, head: [g[f"{tag}.{layers}.0.weight"], g[f"{tag}.{layers}.0.bias"], g[f"{tag}.{layers}.1.weight"],
                                    g[f"{tag}.{layers}.1.bias"], g[f"{tag}.{head}.weight"], g[f"{tag}.{head}.bias"]]
    actor, old, critic = wb("pi", "actor_layers", "means"), wb("old", "actor_layers", "means"), wb("vf", "critic_layers", "network_out")
    obs = g["obs"].astype(np.float32)
    mobs = obs[:, a["obs_src"]] * a["obs_sign"]
    for i in (31, 32):                                      # mirror_clock_observation (wrappers.py:59-72)
        mobs[:, i] = np.sin(np.arcsin(mobs[:, i]) + np.float32(np.pi))
    old_mu = oracle.mlp_forward(obs, *old)
    sd = np.full(12, a["std"], np.float32)
    ga, gc, scal = oracle.ppo_update(obs, a["action"], a["adv"], a["ret"], old_mu, actor, critic, sd, mir_obs=mobs.astype(np.float32),
                                     act_src=a["act_src"], act_sign=a["act_sign"], clip=a["clip"], vf_coeff=0.5,
                                     mirror_coeff=0.4, parts_actor=2, parts_critic=3)
    names = ("actor_loss", "entropy_penalty", "critic_loss", "approx_kl_div", "mirror_loss", "clip_fraction")
    for i, n in enumerate(names):
        np.testing.assert_allclose(scal[i], float(g[n]), rtol=3e-5, atol=3e-7, err_msg=n)
    # gradients: torch autograd of the same losses on the fixture's weights
    c = dict(obs=obs, action=a["action"], adv=a["adv"], ret=a["ret"], actor=actor, critic=critic, old_mu=old_mu, sd=sd,
             a_mean=None, a_std=None, mir_obs=mobs.astype(np.float32), act_src=a["act_src"], act_sign=a["act_sign"])
    ta, tc, ts = torch_ppo_update_grads(c, clip=a["clip"], mirror_coeff=0.4)
    np.testing.assert_allclose(scal, ts, rtol=3e-5, atol=3e-7)
    assert_grads_close(ga, ta, 41, 12)
    assert_grads_close(gc, tc, 41, 1)


@pytest.mark.parametrize("n,B,mirror,parts", [(96, None, True, (1, 1)), (96, 50, True, (3, 2)), (70, 70, False, (5, 4)),
                                                (33, 16, False, (1, 1))])
def test_ppo_update_oracle_gradients_match_torch_autograd(oracle, n, B, mirror, parts):
    c = ppo_update_case(7 + n, n=n, mirror=mirror)
    idx = None if B is None else np.random.default_rng(n).permutation(n)[:B].astype(np.int32)
    kw = dict(mir_obs=c["mir_obs"], act_src=c["act_src"], act_sign=c["act_sign"]) if mirror else {}
    ga, gc, scal = oracle.ppo_update(c["obs"], c["action"], c["adv"], c["ret"], c["old_mu"], c["actor"], c["critic"], c["sd"],
                                     idx=idx, a_mean=c["a_mean"], a_std=c["a_std"], clip=0.2, vf_coeff=0.5, mirror_coeff=0.4,
                                     parts_actor=parts[0], parts_critic=parts[1], **kw)
    ta, tc, ts = torch_ppo_update_grads(c, idx=idx, mirror_coeff=0.4 if mirror else None)
    np.testing.assert_allclose(scal, ts, rtol=3e-5, atol=3e-7)
    assert_grads_close(ga, ta, 41, 12)
    assert_grads_close(gc, tc, 41, 1)
    assert np.abs(ga).max() > 0 and np.abs(gc).max() > 0


@pytest.mark.parametrize("in_dim,act_dim,mirror", [(10, 3, True), (16, 16, False), (30, 16, True), (49, 7, False), (64, 12, True)])
def test_ppo_update_oracle_other_network_widths(oracle, in_dim, act_dim, mirror):
    """1 - 4 groups of 16 inputs, action widths up to 16: the oracle twin against torch autograd."""
    c = ppo_update_case(in_dim + act_dim, n=70, in_dim=in_dim, act_dim=act_dim, mirror=mirror)
    idx = np.random.default_rng(3).permutation(70)[:53].astype(np.int32)
    kw = dict(mir_obs=c["mir_obs"], act_src=c["act_src"], act_sign=c["act_sign"]) if mirror else {}
    ga, gc, scal = oracle.ppo_update(c["obs"], c["action"], c["adv"], c["ret"], c["old_mu"], c["actor"], c["critic"], c["sd"],
                                     idx=idx, a_mean=c["a_mean"], a_std=c["a_std"], clip=0.2, vf_coeff=0.5, mirror_coeff=0.4,
                                     parts_actor=2, parts_critic=3, **kw)
    ta, tc, ts = torch_ppo_update_grads(c, idx=idx, mirror_coeff=0.4 if mirror else None)
    np.testing.assert_allclose(scal, ts, rtol=3e-5, atol=3e-7)
    assert_grads_close(ga, ta, in_dim, act_dim)
    assert_grads_close(gc, tc, in_dim, 1)


def test_ppo_update_oracle_part_count_changes_only_the_last_bits(oracle):
    """The split of the tiles over workgroups is a summation order: results agree to float32 rounding, and a
    single tile (B <= 16) does not depend on it at all."""
    c = ppo_update_case(3, n=80, mirror=False)
    args = (c["obs"], c["action"], c["adv"], c["ret"], c["old_mu"], c["actor"], c["critic"], c["sd"])
    kw = dict(a_mean=c["a_mean"], a_std=c["a_std"])
    g1 = oracle.ppo_update(*args, parts_actor=1, parts_critic=1, **kw)
    g5 = oracle.ppo_update(*args, parts_actor=5, parts_critic=2, **kw)
    assert_grads_close(g5[0], g1[0], 41, 12, rtol=2e-6)
    assert_grads_close(g5[1], g1[1], 41, 1, rtol=2e-6)
    np.testing.assert_allclose(g5[2], g1[2], rtol=1e-13)
    idx = np.arange(16, dtype=np.int32)
    s1 = oracle.ppo_update(*args, idx=idx, parts_actor=1, parts_critic=1, **kw)
    assert np.isfinite(s1[0]).all()


def torch_clip_adam(params, grads_per_step, lr, eps, max_norm):
    """torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step (the reference's optimiser calls, ppo.py:399-410) on a
    list of parameter arrays, one gradient list per step -> the stepped parameters (float32, CPU)."""
    import torch
    ps = [torch.nn.Parameter(torch.tensor(p)) for p in params]
    opt = torch.optim.Adam(ps, lr=lr, eps=eps)
    for grads in grads_per_step:
        for p, g in zip(ps, grads):
            p.grad = torch.tensor(g)
        torch.nn.utils.clip_grad_norm_(ps, max_norm)
        opt.step()
    return [p.detach().numpy() for p in ps]


def adam_case(seed, n=79628, steps=4, scale=1.0):
    rng = np.random.default_rng(seed)
    cuts = np.cumsum([256 * 41, 256, 65536, 256])          # several parameter tensors: the norm is over all of them
    p0 = rng.normal(0, 0.1, n).astype(np.float32)
    grads = [(scale * rng.normal(0, 10.0 ** rng.uniform(-4, -1), n)).astype(np.float32) for _ in range(steps)]
    return p0, grads, cuts


@pytest.mark.parametrize("scale", [1.0, 1e-4])           # gradients far above / below the clip threshold
def test_ppo_adam_oracle_matches_torch_clip_and_adam(oracle, scale):
    p0, grads, cuts = adam_case(3, scale=scale)
    want = np.concatenate([a.reshape(-1) for a in torch_clip_adam(np.split(p0, cuts), [np.split(g, cuts) for g in grads],
                                                                  1e-4, 1e-5, 0.05)])
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for t, g in enumerate(grads):
        p, m, v = oracle.ppo_adam_step(p, g, m, v, t + 1, 1e-4, eps=1e-5, max_norm=0.05)
    # same formulas, float32 rounding of a handful of operations per step on updates of size ~lr
    assert np.abs(p - want).max() <= 2e-7 * 4, np.abs(p - want).max()
    assert np.abs(p - p0).max() > 1e-6
