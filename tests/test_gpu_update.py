"""K14 (oly_ppo_update_grads): the gradients of one PPO minibatch update on the f32 matrix cores, through the C ABI,
against the oracle twin (bit-exact), torch autograd (summation-order tolerance) and the reference-run fixture."""
import numpy as np
import pytest
import torch

from helpers import ppo_update_arrays, ppo_update_case, torch_ppo_update_grads
from test_update_cpu import GRAD_RTOL, assert_grads_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from olympic_hip.engine import Engine
    return Engine(0)


def d(a):
    return None if a is None else torch.as_tensor(np.ascontiguousarray(a)).cuda()


def run_kernel(eng, c, idx=None, mirror=False, parts=(0, 0), mirror_coeff=0.4, clip=0.2, old_mu=None):
    in_dim, act_dim = c["obs"].shape[1], c["action"].shape[1]
    pa = eng.mlp_pack(*[d(a) for a in c["actor"]], d(c["a_mean"]), d(c["a_std"]))
    pc = eng.mlp_pack(*[d(a) for a in c["critic"]])
    B = len(idx) if idx is not None else len(c["obs"])
    ws_n, p_a, p_c = eng.ppo_update_plan(B, in_dim, act_dim, mirror)
    if parts != (0, 0):
        p_a, p_c = parts
    from olympic_hip._ffi import lib
    ga = torch.full((int(lib().oly_ppo_update_grad_floats(in_dim, 256, act_dim)),), float("nan"), device="cuda")
    gc = torch.full((int(lib().oly_ppo_update_grad_floats(in_dim, 256, 1)),), float("nan"), device="cuda")
    scal = torch.zeros(6, dtype=torch.float64, device="cuda")
    ws = torch.empty(ws_n + 4096, dtype=torch.float32, device="cuda")
    sd, lsd = d(c["sd"]), d(c["log_sd"])
    kw = dict(mir_obs=d(c["mir_obs"]), act_src=d(c["act_src"]), act_sign=d(c["act_sign"])) if mirror else {}
    eng.ppo_update_grads(d(c["obs"]), d(c["action"]), d(c["adv"]), d(c["ret"]), d(c["old_mu"] if old_mu is None else old_mu),
                         pa, pc, sd, lsd, sd, lsd, ga, gc, scal, ws, idx=d(idx), normalize_actor=c["a_mean"] is not None,
                         clip=clip, vf_coeff=0.5, mirror_coeff=mirror_coeff, parts=parts, **kw)
    torch.cuda.synchronize()
    return ga.cpu().numpy(), gc.cpu().numpy(), scal.cpu().numpy(), (p_a, p_c)


def oracle_update(oracle, c, idx, mirror, parts, mirror_coeff=0.4, clip=0.2, old_mu=None):
    kw = dict(mir_obs=c["mir_obs"], act_src=c["act_src"], act_sign=c["act_sign"]) if mirror else {}
    return oracle.ppo_update(c["obs"], c["action"], c["adv"], c["ret"], c["old_mu"] if old_mu is None else old_mu, c["actor"],
                             c["critic"], c["sd"], log_sd=c["log_sd"], old_log_sd=c["log_sd"], idx=idx, a_mean=c["a_mean"],
                             a_std=c["a_std"], clip=clip, vf_coeff=0.5, mirror_coeff=mirror_coeff, parts_actor=parts[0],
                             parts_critic=parts[1], **kw)


CASES = [  # n rows in the buffer, B rows of the minibatch (None: all, in order), mirror, normalise, parts (0, 0: the plan's)
    (16, None, False, False, (0, 0)),          # one full tile
    (5, None, False, True, (0, 0)),            # one ragged tile
    (96, None, True, True, (0, 0)),            # one tile per workgroup
    (96, 50, True, True, (2, 1)),              # several tiles per workgroup, ragged last tile, gathered rows
    (200, 131, False, True, (3, 4)),
    (200, 200, True, False, (4, 2)),
]


@pytest.mark.parametrize("n,B,mirror,normalize,parts", CASES)
def test_ppo_update_kernel_equals_the_oracle(eng, oracle, n, B, mirror, normalize, parts):
    """Gradients BIT-EXACT (same fma chains, same order, same exp32); the six scalars to 1e-12 (fp64 sums of the same
    float32 terms in a different, fixed order)."""
    c = ppo_update_case(100 + n, n=n, mirror=mirror, normalize=normalize)
    idx = None if B is None else np.random.default_rng(n).permutation(n)[:B].astype(np.int32)
    ga, gc, scal, used = run_kernel(eng, c, idx, mirror, parts)
    oa, oc, os_ = oracle_update(oracle, c, idx, mirror, used)
    assert np.isfinite(ga).all() and np.isfinite(gc).all()
    assert np.array_equal(ga, oa), np.abs(ga - oa).max()
    assert np.array_equal(gc, oc), np.abs(gc - oc).max()
    np.testing.assert_allclose(scal, os_, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("in_dim,act_dim,mirror", [(10, 3, True), (16, 16, False), (30, 16, True), (49, 7, False), (64, 12, True)])
def test_ppo_update_kernel_other_network_widths(eng, oracle, in_dim, act_dim, mirror):
    """Every instantiation of the kernel (1 - 4 groups of 16 inputs) and action widths up to the loss wave's 16 columns:
    bit-exact against the oracle, and the oracle against torch autograd."""
    c = ppo_update_case(in_dim + act_dim, n=70, in_dim=in_dim, act_dim=act_dim, mirror=mirror)
    idx = np.random.default_rng(3).permutation(70)[:53].astype(np.int32)
    ga, gc, scal, used = run_kernel(eng, c, idx, mirror, (2, 3))
    oa, oc, os_ = oracle_update(oracle, c, idx, mirror, used)
    assert np.array_equal(ga, oa) and np.array_equal(gc, oc)
    np.testing.assert_allclose(scal, os_, rtol=1e-12, atol=1e-15)
    ta, tc, ts = torch_ppo_update_grads(c, idx=idx, mirror_coeff=0.4 if mirror else None)
    np.testing.assert_allclose(scal, ts, rtol=3e-5, atol=3e-7)
    assert_grads_close(ga, ta, in_dim, act_dim, GRAD_RTOL)
    assert_grads_close(gc, tc, in_dim, 1, GRAD_RTOL)


def test_ppo_update_kernel_at_the_full_minibatch(eng):
    """BASELINE config 3's own minibatch: 65 536 gathered rows of a 70 000-row buffer, the mirror loss on, the plan's own
    split over all 256 workgroups; the oracle twin would need minutes here (the 16 384-row case below is bit-exact against
    it).  Against torch autograd in float32: the six scalars to 1e-5; the critic's gradients to 2e-5 of each tensor's
    largest element.  The actor's only to 2e-3: with 65 536 rows x 512 hidden units a few dozen pre-activations lie within
    rounding of zero, every evaluation order (the kernel's chains, torch's GEMMs, float64: 7e-4 from BOTH) puts some of
    them on the other side of the ReLU, and each such flip moves a row / column of a weight gradient by a whole row's
    contribution.  What this case guards is the bookkeeping at full size: no tile lost or counted twice (>= 4e-3)."""
    c = ppo_update_case(11, n=70000, mirror=True)
    idx = np.random.default_rng(1).permutation(70000)[:65536].astype(np.int32)
    ga, gc, scal, used = run_kernel(eng, c, idx, True)
    assert used[0] + used[1] >= 200                       # the whole chip takes part
    ta, tc, ts = torch_ppo_update_grads(c, idx=idx, mirror_coeff=0.4)
    np.testing.assert_allclose(scal, ts, rtol=1e-5, atol=1e-8)
    assert_grads_close(ga, ta, 41, 12, 2e-3)
    assert_grads_close(gc, tc, 41, 1, GRAD_RTOL)


def test_ppo_update_kernel_sixteen_thousand_rows_equal_the_oracle(eng, oracle):
    """16 384 gathered rows, mirror loss on, the plan's own split: every one of the 256 workgroups runs 6 - 13 tiles.
    Bit-exact against the oracle twin (about 20 s of scalar fma chains on the host)."""
    c = ppo_update_case(12, n=20000, mirror=True)
    idx = np.random.default_rng(2).permutation(20000)[:16384].astype(np.int32)
    ga, gc, scal, used = run_kernel(eng, c, idx, True)
    assert used[0] + used[1] >= 200
    oa, oc, os_ = oracle_update(oracle, c, idx, True, used)
    assert np.array_equal(ga, oa) and np.array_equal(gc, oc)
    np.testing.assert_allclose(scal, os_, rtol=1e-12, atol=1e-15)


def test_ppo_update_kernel_larger_minibatch_equals_the_oracle(eng, oracle):
    """2085 gathered rows of a 6000-row buffer with the plan's own split (131 tiles over 128 + 128 workgroups)."""
    c = ppo_update_case(5, n=6000, mirror=True)
    idx = np.random.default_rng(0).permutation(6000)[:2085].astype(np.int32)
    ga, gc, scal, used = run_kernel(eng, c, idx, True)
    oa, oc, os_ = oracle_update(oracle, c, idx, True, used)
    assert np.array_equal(ga, oa) and np.array_equal(gc, oc)
    np.testing.assert_allclose(scal, os_, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("mirror", [False, True])
def test_ppo_update_kernel_vs_torch_autograd(eng, mirror):
    """Against what the reference's backward() calls compute (torch autograd, float32, CPU): summation order only."""
    c = ppo_update_case(11, n=300, mirror=mirror)
    idx = np.random.default_rng(1).permutation(300)[:256].astype(np.int32)
    ga, gc, scal, _ = run_kernel(eng, c, idx, mirror)
    ta, tc, ts = torch_ppo_update_grads(c, idx=idx, mirror_coeff=0.4 if mirror else None)
    np.testing.assert_allclose(scal, ts, rtol=3e-5, atol=3e-7)
    assert_grads_close(ga, ta, 41, 12, GRAD_RTOL)
    assert_grads_close(gc, tc, 41, 1, GRAD_RTOL)


def test_ppo_update_kernel_on_the_reference_fixture(eng, golden):
    """The reference's own update_policy outputs (ppo_update.npz: its Gaussian_FF_Actor / FF_V weights, mirror tables)."""
    g = golden("ppo_update.npz")
    a = ppo_update_arrays(g)
    wb = lambda tag, layers, head: [g[f"{tag}.{layers}.0.weight"], g[f"{tag}.{layers}.0.bias"], g[f"{tag}.{layers}.1.weight"],
                                    g[f"{tag}.{layers}.1.bias"], g[f"{tag}.{head}.weight"], g[f"{tag}.{head}.bias"]]
    obs = g["obs"].astype(np.float32)
    mobs = obs[:, a["obs_src"]] * a["obs_sign"]
    for i in (31, 32):
        mobs[:, i] = np.sin(np.arcsin(mobs[:, i]) + np.float32(np.pi))
    sd = np.full(12, a["std"], np.float32)
    c = dict(obs=obs, action=a["action"], adv=a["adv"], ret=a["ret"], actor=wb("pi", "actor_layers", "means"),
             critic=wb("vf", "critic_layers", "network_out"), sd=sd, log_sd=np.log(sd), a_mean=None, a_std=None,
             mir_obs=mobs.astype(np.float32), act_src=a["act_src"], act_sign=a["act_sign"])
    old = wb("old", "actor_layers", "means")
    po = eng.mlp_pack(*[d(x) for x in old])
    old_mu = torch.empty((64, 12), dtype=torch.float32, device="cuda")
    eng.mlp_forward2(d(obs), po, 12, old_mu)                 # the old policy's means as the product computes them (K11)
    c["old_mu"] = old_mu.cpu().numpy()
    ga, gc, scal, _ = run_kernel(eng, c, None, True, clip=a["clip"])
    for i, n in enumerate(("actor_loss", "entropy_penalty", "critic_loss", "approx_kl_div", "mirror_loss", "clip_fraction")):
        np.testing.assert_allclose(scal[i], float(g[n]), rtol=3e-5, atol=3e-7, err_msg=n)
    ta, tc, _ = torch_ppo_update_grads(c, clip=a["clip"], mirror_coeff=0.4)
    assert_grads_close(ga, ta, 41, 12, GRAD_RTOL)
    assert_grads_close(gc, tc, 41, 1, GRAD_RTOL)


def test_ppo_update_first_minibatch_has_ratio_one(eng):
    """old_policy == policy at the first minibatch of an iteration: with old_mu from K11 on the same weights the
    kernel's own forward must reproduce it bit for bit, i.e. ratio == 1 exactly: approx_kl == 0, clip_fraction == 0,
    actor_loss == -mean(adv)."""
    c = ppo_update_case(21, n=500, mirror=False)
    pa = eng.mlp_pack(*[d(a) for a in c["actor"]], d(c["a_mean"]), d(c["a_std"]))
    old_mu = torch.empty((500, 12), dtype=torch.float32, device="cuda")
    eng.mlp_forward2(d(c["obs"]), pa, 12, old_mu, normalize_a=True)
    ga, gc, scal, _ = run_kernel(eng, c, None, False, old_mu=old_mu.cpu().numpy())
    assert scal[3] == 0.0 and scal[5] == 0.0
    np.testing.assert_allclose(scal[0], -np.mean(c["adv"].astype(np.float64)), rtol=1e-12)


def test_ppo_update_rejects_bad_arguments(eng):
    from olympic_hip._ffi import OlyError
    c = ppo_update_case(1, n=40, mirror=False)
    with pytest.raises(OlyError, match="parts"):
        run_kernel(eng, c, None, False, parts=(7, 1))            # 3 tiles cannot feed 7 workgroups
    with pytest.raises(OlyError, match="idx"):
        run_kernel(eng, c, np.arange(8, dtype=np.int64), False)  # int64 indices are refused on the host
    big = dict(c, action=np.zeros((40, 17), np.float32), old_mu=np.zeros((40, 17), np.float32), sd=np.ones(17, np.float32),
               log_sd=np.zeros(17, np.float32))
    with pytest.raises(OlyError):
        run_kernel(eng, big, None, False)                        # 17 actions: the loss wave holds 16 columns


@pytest.mark.parametrize("scale", [1.0, 1e-4])
def test_ppo_adam_step_equals_the_oracle_and_torch(eng, oracle, scale):
    """oly_ppo_adam_step: clip_grad_norm_ + Adam.step for both networks + the re-pack, four steps in a row.  Bit-exact
    against the oracle twin; against torch's own clip + Adam on the CPU to float32 rounding; the re-packed stream equals
    a fresh oly_mlp_pack of the stepped parameters."""
    from test_update_cpu import adam_case, torch_clip_adam
    from olympic_hip._ffi import lib
    sizes = (int(lib().oly_ppo_update_grad_floats(41, 256, 12)), int(lib().oly_ppo_update_grad_floats(41, 256, 1)))
    cases = [adam_case(5 + i, n=n, scale=scale) for i, n in enumerate(sizes)]
    mean, std = d(np.linspace(-1, 1, 41, dtype=np.float32)), d(np.linspace(0.5, 2, 41, dtype=np.float32))
    nets = []
    for (p0, grads, _), out_dim in zip(cases, (12, 1)):
        z = torch.zeros(len(p0), device="cuda")
        nets.append(dict(param=d(p0), grad=z.clone(), exp_avg=z.clone(), exp_avg_sq=z.clone(), out_dim=out_dim,
                         packed=torch.zeros(eng._mlp_floats(41, out_dim), device="cuda"),
                         in_mean=mean if out_dim == 12 else None, in_std=std if out_dim == 12 else None))
    ws = torch.zeros(1024, dtype=torch.float64, device="cuda")
    ref = [(c[0].copy(), np.zeros_like(c[0]), np.zeros_like(c[0])) for c in cases]
    for t in range(4):
        for nt, c in zip(nets, cases):
            nt["grad"].copy_(d(c[1][t]))
        eng.ppo_adam_step(41, t + 1, 1e-4, 1e-5, 0.05, nets, ws)
        ref = [oracle.ppo_adam_step(r[0], c[1][t], r[1], r[2], t + 1, 1e-4, eps=1e-5, max_norm=0.05) for r, c in zip(ref, cases)]
    torch.cuda.synchronize()
    for nt, r, (p0, grads, cuts) in zip(nets, ref, cases):
        for k, want in zip(("param", "exp_avg", "exp_avg_sq"), r):
            assert np.array_equal(nt[k].cpu().numpy(), want), k
        cuts = np.cumsum([256 * 41, 256, 65536, 256, nt["out_dim"] * 256])
        tw = np.concatenate([a.reshape(-1) for a in torch_clip_adam(np.split(p0, cuts), [np.split(g, cuts) for g in grads],
                                                                      1e-4, 1e-5, 0.05)])
        assert np.abs(nt["param"].cpu().numpy() - tw).max() <= 8e-7
        parts = [a.contiguous() for a in torch.split(nt["param"], [256 * 41, 256, 65536, 256, nt["out_dim"] * 256, nt["out_dim"]])]
        fresh = eng.mlp_pack(parts[0].view(256, 41), parts[1], parts[2].view(256, 256), parts[3], parts[4].view(-1, 256), parts[5],
                             nt["in_mean"], nt["in_std"])
        assert torch.equal(fresh, nt["packed"])


def test_kernel_update_object_steps_like_torch(eng):
    """ppo.KernelUpdate on the reference-shaped modules: after begin() + three step()s the modules' own parameters (now
    views of the flat buffers) equal a plain torch run of update_policy + backward + clip_grad_norm_ + Adam.step on
    copies of the modules, to float32 rounding."""
    import copy
    from olympic_hip.ppo import PPO, KernelUpdate, MLPCritic, MLPGaussianActor
    torch.manual_seed(3)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    pi.obs_mean, pi.obs_std = torch.randn(41, device="cuda") * 0.1, torch.rand(41, device="cuda") + 0.5
    old = copy.deepcopy(pi)
    pi_t, vf_t, old_t = copy.deepcopy(pi), copy.deepcopy(vf), copy.deepcopy(pi)
    n, B = 512, 128
    obs, act = torch.randn(n, 41, device="cuda"), torch.randn(n, 12, device="cuda") * 0.3
    ret, adv = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    ku = KernelUpdate(eng, pi, vf, old, 0.2, 0.5, 0.0, lr=1e-3, eps=1e-5, max_grad_norm=0.05)
    ku.begin(obs)
    ppo = PPO.__new__(PPO)
    ppo.clip, ppo.vf_coeff, ppo.policy, ppo.critic, ppo.old_policy = 0.2, 0.5, pi_t, vf_t, old_t
    oa = torch.optim.Adam(pi_t.parameters(), lr=1e-3, eps=1e-5)
    oc = torch.optim.Adam(vf_t.parameters(), lr=1e-3, eps=1e-5)
    for s in range(3):
        idx = torch.randperm(n, device="cuda")[:B]
        scal = ku.step(obs, act, ret, adv, idx.to(torch.int32)).clone()
        out = ppo.update_policy(obs[idx], act[idx], ret[idx].reshape(-1, 1), adv[idx].reshape(-1, 1), 1)
        oa.zero_grad(); oc.zero_grad()
        out[0].backward(); out[2].backward()
        torch.nn.utils.clip_grad_norm_(pi_t.parameters(), 0.05); oa.step()
        torch.nn.utils.clip_grad_norm_(vf_t.parameters(), 0.05); oc.step()
        np.testing.assert_allclose(scal[[0, 2, 3]].cpu().numpy(), [float(out[0]), float(out[2]), float(out[3])], rtol=5e-5, atol=1e-6)
    for a, b in zip(list(pi.parameters()) + list(vf.parameters()), list(pi_t.parameters()) + list(vf_t.parameters())):
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-3)
    assert ku.steps == 3


@pytest.mark.parametrize("B,n_batches", [(64, 7), (128, 1), (100, 4)])
def test_kernel_update_epoch_call_equals_the_step_loop(eng, B, n_batches):
    """oly_ppo_update_epoch (one C call per epoch) against n_batches calls of KernelUpdate.step on the consecutive cuts of
    the same permutation: every parameter, both Adam moments, the packed streams and the scalars of every minibatch, to
    the bit (the same launches in the same order); the step count advances by n_batches."""
    import copy
    from olympic_hip.ppo import KernelUpdate, MLPCritic, MLPGaussianActor
    torch.manual_seed(4)
    n = 900
    obs, act = torch.randn(n, 41, device="cuda"), torch.randn(n, 12, device="cuda") * 0.3
    ret, adv = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    perm = torch.randperm(n, device="cuda").to(torch.int32)
    runs = []
    for one_call in (False, True):
        torch.manual_seed(5)
        pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
        pi.obs_mean, pi.obs_std = torch.linspace(-0.2, 0.2, 41).cuda(), torch.linspace(0.7, 1.4, 41).cuda()
        ku = KernelUpdate(eng, pi, vf, copy.deepcopy(pi), 0.2, 0.5, 0.0, lr=1e-3, eps=1e-5, max_grad_norm=0.05)
        ku.begin(obs)
        scal = torch.zeros((2 * n_batches, 6), dtype=torch.float64, device="cuda")
        for ep in range(2):                              # a second epoch: the call continues the step count
            rows = scal[ep * n_batches:(ep + 1) * n_batches]
            if one_call:
                ku.epoch(obs, act, ret, adv, perm, B, n_batches, rows)
            else:
                for b in range(n_batches):
                    ku.step(obs, act, ret, adv, perm[b * B:(b + 1) * B], rows[b])
        torch.cuda.synchronize()
        assert ku.steps == 2 * n_batches
        runs.append(dict(scal=scal, **{f"{k}{i}": nt[k].clone() for i, nt in enumerate(ku.nets)
                                       for k in ("param", "exp_avg", "exp_avg_sq", "grad", "packed")}))
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k
    assert float(runs[0]["scal"][:, 3].abs().max()) > 0                    # the policy moved between minibatches


@pytest.mark.parametrize("in_dim,out_a", [(10, 3), (16, 16), (33, 7), (64, 12)])
def test_ppo_adam_step_keeps_the_packed_streams_current_for_other_widths(eng, in_dim, out_a):
    """The optimiser launch writes every stepped weight to its places in the packed stream (the inverse of oly_mlp_pack's
    map): after two steps on random gradients the stream equals a fresh oly_mlp_pack of the stepped parameters, element
    for element, for 1 - 4 groups of 16 inputs and several output widths (tables and zero padding included)."""
    from olympic_hip._ffi import lib
    g = torch.Generator(device="cuda").manual_seed(in_dim + out_a)
    mean = torch.empty(in_dim, device="cuda").normal_(0, 0.3, generator=g)
    std = torch.empty(in_dim, device="cuda").uniform_(0.5, 2.0, generator=g)
    nets = []
    for out_dim in (out_a, 1):
        n = int(lib().oly_ppo_update_grad_floats(in_dim, 256, out_dim))
        p0 = torch.empty(n, device="cuda").normal_(0, 0.2, generator=g)
        z = torch.zeros(n, device="cuda")
        nets.append(dict(param=p0, grad=z.clone(), exp_avg=z.clone(), exp_avg_sq=z.clone(), out_dim=out_dim,
                         packed=torch.zeros(eng._mlp_floats(in_dim, out_dim), device="cuda"),
                         in_mean=mean if out_dim == out_a else None, in_std=std if out_dim == out_a else None))
    ws = torch.zeros(1024, dtype=torch.float64, device="cuda")
    for t in range(2):
        for nt in nets:
            nt["grad"].normal_(0, 1e-3, generator=g)
        before = [nt["param"].clone() for nt in nets]
        eng.ppo_adam_step(in_dim, t + 1, 1e-3, 1e-5, 0.05, nets, ws)
        assert all(not torch.equal(b, nt["param"]) for b, nt in zip(before, nets))
    for nt in nets:
        od = nt["out_dim"]
        parts = [a.contiguous() for a in torch.split(nt["param"], [256 * in_dim, 256, 65536, 256, od * 256, od])]
        fresh = eng.mlp_pack(parts[0].view(256, in_dim), parts[1], parts[2].view(256, 256), parts[3], parts[4].view(-1, 256),
                             parts[5], nt["in_mean"], nt["in_std"])
        assert torch.equal(fresh, nt["packed"]), (in_dim, od, int((fresh != nt["packed"]).sum()))


def test_update_epoch_call_rejects_inconsistent_arguments(eng):
    """oly_ppo_update_epoch refuses what would silently train on the wrong buffers: an optimiser block whose gradient /
    packed-stream pointers are not the update's, a norm workspace that is not shared, a permutation shorter than the
    minibatches asked for (host side), a negative count."""
    import copy
    import ctypes as C
    from olympic_hip._ffi import OlyError, lib
    from olympic_hip.ppo import KernelUpdate, MLPCritic, MLPGaussianActor
    torch.manual_seed(9)
    pi, vf = MLPGaussianActor(41, 12).cuda(), MLPCritic(41).cuda()
    n, B = 300, 64
    obs, act = torch.randn(n, 41, device="cuda"), torch.randn(n, 12, device="cuda") * 0.3
    ret, adv = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    ku = KernelUpdate(eng, pi, vf, copy.deepcopy(pi), 0.2, 0.5, 0.0)
    ku.begin(obs)
    perm = torch.randperm(n, device="cuda").to(torch.int32)
    scal = torch.zeros((4, 6), dtype=torch.float64, device="cuda")
    ku.step(obs, act, ret, adv, perm[:B], scal[0])                     # prepares the two launches
    g, a = ku._last_launch, ku._apply[1]
    with pytest.raises(OlyError, match="perm"):
        eng.ppo_update_epoch(g, a, ku.steps + 1, perm[:100], 3, scal[1:])     # 3 x 64 rows asked of 100 indices
    keep = a.struct.ws
    a.struct.ws = scal.data_ptr()                                              # not the update's gnorm_ws
    with pytest.raises(OlyError, match="gnorm_ws"):
        eng.ppo_update_epoch(g, a, ku.steps + 1, perm[B:], 3, scal[1:])
    a.struct.ws = keep
    keep = a.struct.net[1].grad
    a.struct.net[1].grad = a.struct.net[0].grad                                # the critic's optimiser on the actor's gradient
    with pytest.raises(OlyError, match="network 1"):
        eng.ppo_update_epoch(g, a, ku.steps + 1, perm[B:], 3, scal[1:])
    a.struct.net[1].grad = keep
    rc = lib().oly_ppo_update_epoch(eng.ctx.handle, C.byref(g.struct), C.byref(a.struct), perm.data_ptr(), -1, scal.data_ptr(), None)
    assert rc != 0
    before = ku.nets[0]["param"].clone()
    eng.ppo_update_epoch(g, a, ku.steps + 1, perm[B:], 3, scal[1:])            # and the well-formed call runs
    torch.cuda.synchronize()
    assert not torch.equal(before, ku.nets[0]["param"]) and bool(torch.isfinite(scal).all())
