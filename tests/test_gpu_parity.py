"""GPU parity: the HIP kernels (through the C ABI) against the CPU oracle on the same seeded
inputs, against the golden vectors, and - at full BASELINE sizes - through size-independent
properties.  Bars: integers / flags / indices bit-exact; floating point within the tolerance
written next to each check."""
import os
import numpy as np
import pytest
import torch

from olympic_hip import _abi, specs
from helpers import (a3_analytic_cases, a3_fixture_arrays, check_a3_analytic, h1_rows_from_full, h1_synthetic_block,
                     ulp_diff)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from olympic_hip.engine import Engine
    return Engine(0)


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def host(t):
    return t.cpu().numpy()


# ------------------------------------------------------------------------------ K1/K5
def _run_il(eng, spec, qpos, qvel, action, prev, **kw):
    eng.il_configure(spec)
    o = eng.il_step(dev(qpos), dev(qvel), None if action is None else dev(action, torch.float32),
                    dev(prev, torch.float64), **kw)
    torch.cuda.synchronize()
    return {k: (None if v is None else host(v)) for k, v in o.items()}


def _cmp_il(o, ref, f64):
    assert np.array_equal(o["obs"], ref["obs"])                    # pure gather + rounding: bit-exact
    assert np.array_equal(o["absorbing"], ref["absorbing"])        # bit-exact
    assert np.array_equal(o["fall_code"], ref["fall_code"])        # bit-exact
    assert ulp_diff(o["reward"], ref["reward"]).max() <= 1          # device exp vs libm exp, 1 ulp f32
    assert np.array_equal(o["prev"], ref["prev"])
    if ref["ctrl"] is not None:
        assert np.array_equal(o["ctrl"], ref["ctrl"])               # mul/add/clamp in fp64: bit-exact


@pytest.mark.parametrize("f64", [False, True])
def test_h1_golden(eng, golden, oracle, f64):
    g = golden("h1_step.npz")
    spec = specs.unitree_h1("walk")
    qpos, qvel = h1_rows_from_full(spec, g["full_obs"])
    a = g["action"].astype(np.float32)
    prev = g["obs"][:, spec.reward_idx].copy()
    o = _run_il(eng, spec, qpos[None], qvel[None], a[None], prev, obs_f64=f64, ctrl_f64=f64)
    ref = oracle.il_step(spec, qpos[None], qvel[None], a[None], prev, obs_f64=f64, ctrl_f64=f64)
    _cmp_il(o, ref, f64)
    # directly against the reference's outputs
    exp_obs = g["obs"] if f64 else g["obs"].astype(np.float32)
    assert np.array_equal(o["obs"][0], exp_obs)
    assert np.array_equal(o["absorbing"][0].astype(bool), g["fallen"])
    assert np.array_equal(o["fall_code"][0], g["msg_code"])
    assert ulp_diff(o["reward"][0], g["reward_walk"].astype(np.float32)).max() <= 1


@pytest.mark.parametrize("T,N", [(1, 1), (1, 127), (1, 128), (3, 129), (7, 333), (2, 4096), (5, 1000)])
def test_h1_vs_oracle_shapes(eng, oracle, T, N):
    """Ragged tails (R % 128 != 0), single row, T > 1 carry chain."""
    spec = specs.unitree_h1("walk")
    qpos, qvel, act = h1_synthetic_block(spec, T, N, seed=T * 1000 + N, fall_frac="wide")
    prev = np.random.default_rng(5).normal(1.25, 0.5, N)
    o = _run_il(eng, spec, qpos, qvel, act, prev)
    ref = oracle.il_step(spec, qpos, qvel, act, prev)
    _cmp_il(o, ref, False)


def test_h1_variants(eng, oracle):
    qp, qv, act = h1_synthetic_block(specs.unitree_h1("walk"), 2, 300, seed=3, fall_frac="wide")
    prev = np.linspace(0, 2.5, 300)
    for spec in (specs.unitree_h1("run"), specs.unitree_h1("walk", use_absorbing_states=False),
                 specs.unitree_h1("walk", reward_type="x_pos"), specs.unitree_h1("walk", reward_type=None)):
        o = _run_il(eng, spec, qp, qv, act, prev)
        ref = oracle.il_step(spec, qp, qv, act, prev)
        if spec.reward_type == _abi.REWARD_NONE:
            ref["prev"] = o["prev"]            # untouched by both
        _cmp_il(o, ref, False)
    # no action -> no ctrl
    o = _run_il(eng, specs.unitree_h1("walk"), qp, qv, None, prev)
    assert o["ctrl"] is None


def test_h1_generic_robot_path(eng, oracle):
    """A spec that is not H1-shaped (arms kept: nq=25, n_obs=48, 19 actions) runs the
    runtime-dimension kernel."""
    spec = specs.unitree_h1("walk", disable_arms=False)
    assert (spec.nq, spec.n_obs, spec.n_act) == (25, 48, 19)
    rng = np.random.default_rng(9)
    T, N = 3, 211
    qpos = rng.uniform(-0.5, 0.5, (T, N, spec.nq))
    qvel = rng.normal(0, 1, (T, N, spec.nv))
    act = rng.uniform(-1.3, 1.3, (T, N, spec.n_act)).astype(np.float32)
    prev = rng.normal(1.25, 0.5, N)
    o = _run_il(eng, spec, qpos, qvel, act, prev)
    ref = oracle.il_step(spec, qpos, qvel, act, prev)
    _cmp_il(o, ref, False)


def test_h1_full_size_properties(eng):
    """BASELINE config 2 size (T=400, N=4096): properties that need no CPU pass."""
    spec = specs.unitree_h1("walk")
    eng.il_configure(spec)
    T, N = 400, 4096
    g = torch.Generator(device="cuda").manual_seed(1234)
    qpos = torch.empty((T, N, 17), dtype=torch.float64, device="cuda").uniform_(-0.4, 0.4, generator=g)
    qvel = torch.empty((T, N, 17), dtype=torch.float64, device="cuda").normal_(0, 1.5, generator=g)
    act = torch.empty((T, N, 11), dtype=torch.float32, device="cuda").uniform_(-1.2, 1.2, generator=g)
    prev = torch.full((N,), 1.25, dtype=torch.float64, device="cuda")
    o = eng.il_step(qpos, qvel, act, prev, obs_f64=True)
    qa = torch.as_tensor(spec.qpos_adr.astype(np.int64)).cuda()
    # (1) obs is exactly the permuted input
    assert torch.equal(o["obs"][..., :15], qpos[..., qa[2:]])
    assert torch.equal(o["obs"][..., 15:], qvel[..., qa])
    # (2) flags agree with the thresholds evaluated by torch in fp64
    ob = o["obs"]
    pi = np.pi
    fallen = ((ob[..., 0] < -0.3) | (ob[..., 0] > 0.1) | (ob[..., 1] < -pi / 4.5) | (ob[..., 1] > pi / 12) |
              (ob[..., 2] < -pi / 12) | (ob[..., 2] > pi / 8) | (ob[..., 3] < -pi / 8) | (ob[..., 3] > pi / 8))
    assert torch.equal(o["absorbing"].bool(), fallen)
    assert torch.equal(o["fall_code"] > 0, fallen)
    # (3) reward chain: reward[t] = exp(-(xvel[t-1]-1.25)^2), reward[0] from prev (=1 here)
    xv = ob[..., 15]
    exp_r = torch.exp(-(xv[:-1] - 1.25) ** 2).float()
    assert (o["reward"][1:] - exp_r).abs().max().item() <= 1.2e-7      # 1 ulp f32 at <= 1.0
    assert torch.equal(o["reward"][0], torch.ones(N, device="cuda"))
    assert torch.equal(o["prev"], xv[-1])
    # (4) ctrl: clamp(a*0.95) scattered; idempotent under the inverse permutation
    a2c = torch.as_tensor(spec.act_to_ctrl.astype(np.int64)).cuda()
    exp_c = torch.clamp(act.double() * 0.95, -0.95, 0.95).float()
    assert torch.equal(o["ctrl"][..., a2c], exp_c)
    # (5) f32 observation path = rounding of the f64 one
    o32 = eng.il_step(qpos, qvel, act, prev)
    assert torch.equal(o32["obs"], ob.float())
    assert torch.equal(o32["absorbing"], o["absorbing"])


def test_il_argument_errors(eng):
    from olympic_hip._ffi import OlyError
    spec = specs.unitree_h1("walk")
    eng.il_configure(spec)
    q = torch.zeros((2, 8, 17), dtype=torch.float64, device="cuda")
    p = torch.zeros(8, dtype=torch.float64, device="cuda")
    with pytest.raises(OlyError):
        eng.il_step(q, q[..., :16].contiguous(), None, p)            # wrong nv
    with pytest.raises(OlyError):
        eng.il_step(q, q, None, p, prev_out=p)                        # aliasing with T > 1
    with pytest.raises(OlyError):
        eng.il_step(q.float(), q, None, p)                            # wrong dtype
    with pytest.raises(OlyError):
        eng.il_step(q.cpu(), q, None, p)                              # host tensor


# --------------------------------------------------------------------------------- K6
@pytest.mark.parametrize("mode", [_abi.SCAN_RETURN, _abi.SCAN_GAE])
@pytest.mark.parametrize("T,N", [(1, 1), (5, 3), (400, 257), (33, 4096),       # fallback kernel (N % 4), one tile
                                 (7, 20), (400, 4096), (97, 1028),             # 16-env workgroups, ragged last group
                                 (70, 16388), (45, 32776)])                    # 32- and 64-env workgroups
def test_scan_vs_oracle(eng, oracle, mode, T, N):
    rng = np.random.default_rng(T * 7 + N)
    r = rng.uniform(-0.3, 1, (T, N)).astype(np.float32)
    v = rng.normal(0, 1, (T, N)).astype(np.float32)
    vn = rng.normal(0, 1, (T, N)).astype(np.float32)
    last = rng.uniform(size=(T, N)) < 1 / 30
    ab = last & (rng.uniform(size=(T, N)) < 0.5)
    flags = (last * _abi.FLAG_LAST + ab * _abi.FLAG_ABSORBING).astype(np.uint8)
    ret, adv = eng.return_scan(mode, 0.99, 0.97, dev(r), dev(v), dev(vn), dev(flags))
    e_ret, e_adv = oracle.return_scan(mode, 0.99, 0.97, r, v, vn, flags)
    assert np.array_equal(host(ret), e_ret)        # same fp op sequence, no FMA: bit-exact
    assert np.array_equal(host(adv), e_adv)


def test_scan_random_shapes(eng, oracle):
    """Random [T,N] with N % 4 == 0 (pipelined kernel: ragged last tile, ragged last workgroup,
    1..3 tiles in flight) and random cut densities, both modes, bit-exact."""
    rng = np.random.default_rng(2024)
    for case in range(2 * int(__import__("os").environ.get("OLY_FUZZ", "12"))):
        T = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 95, 96, 97, 129, 200]))
        N = 4 * int(rng.integers(1, 700))
        dens = float(rng.choice([0.0, 1 / 300, 0.05, 0.5, 1.0]))
        r, v, vn = (rng.normal(0, 1, (T, N)).astype(np.float32) for _ in range(3))
        last = rng.uniform(size=(T, N)) < dens
        ab = last & (rng.uniform(size=(T, N)) < 0.5)
        ab |= (~last) & (rng.uniform(size=(T, N)) < 0.01)              # absorbing without last: ignored by RETURN
        flags = (last * _abi.FLAG_LAST + ab * _abi.FLAG_ABSORBING).astype(np.uint8)
        mode = _abi.SCAN_RETURN if case % 2 == 0 else _abi.SCAN_GAE
        gam, lam = float(rng.choice([0.99, 1.0, 0.0, 0.9])), float(rng.choice([0.97, 1.0, 0.0]))
        ret, adv = eng.return_scan(mode, gam, lam, dev(r), dev(v), dev(vn), dev(flags))
        e_ret, e_adv = oracle.return_scan(mode, gam, lam, r, v, vn, flags)
        assert np.array_equal(host(ret), e_ret) and np.array_equal(host(adv), e_adv), (case, T, N, dens, mode)


def test_scan_unaligned_views_take_the_fallback(eng, oracle):
    """Buffers that are not 16-byte aligned (views at an odd element offset) are legal input."""
    rng = np.random.default_rng(2)
    T, N = 50, 64
    r, v, vn = (rng.normal(0, 1, T * N + 1).astype(np.float32) for _ in range(3))
    flags = (rng.uniform(size=T * N + 1) < 0.05).astype(np.uint8) * 3
    view = lambda a: dev(a)[1:].view(T, N)
    for mode in (_abi.SCAN_RETURN, _abi.SCAN_GAE):
        ret, adv = eng.return_scan(mode, 0.99, 0.97, view(r), view(v), view(vn), view(flags))
        e_ret, e_adv = oracle.return_scan(mode, 0.99, 0.97, r[1:].reshape(T, N), v[1:].reshape(T, N),
                                          vn[1:].reshape(T, N), flags[1:].reshape(T, N))
        assert np.array_equal(host(ret), e_ret) and np.array_equal(host(adv), e_adv)


@pytest.mark.parametrize("mode", [_abi.SCAN_RETURN, _abi.SCAN_GAE])
def test_scan_config5_full_size_shard_independence(eng, mode):
    """Config 5 at full size ([400, 32768] = 8 x 4096): environments are independent, so the
    whole block (64-env workgroups) must equal its eight 4096-env shards scanned separately
    (16-env workgroups) bit for bit; the (count, sum, sumsq) triples of the shards add up to
    the block's (the quantity the RCCL all-gather carries)."""
    T, N, G = 400, 32768, 8
    g = torch.Generator(device="cuda").manual_seed(11)
    r, v, vn = (torch.empty((T, N), device="cuda").normal_(0, 1, generator=g) for _ in range(3))
    fl = (torch.rand((T, N), device="cuda", generator=g) < 1 / 300).to(torch.uint8) * _abi.FLAG_LAST
    fl |= ((torch.rand((T, N), device="cuda", generator=g) < 0.5).to(torch.uint8) * _abi.FLAG_ABSORBING) & (fl >> 1)
    ret, adv = eng.return_scan(mode, 0.99, 0.97, r, v, vn, fl)
    tot = torch.zeros(3, dtype=torch.float64, device="cuda")
    for k in range(G):
        sl = slice(k * N // G, (k + 1) * N // G)
        rs, as_ = eng.return_scan(mode, 0.99, 0.97, *(t[:, sl].contiguous() for t in (r, v, vn, fl)))
        assert torch.equal(rs, ret[:, sl]) and torch.equal(as_, adv[:, sl])
        tot += eng.adv_stats(as_)
    whole = eng.adv_stats(adv)
    assert float(whole[0]) == float(tot[0]) == T * N
    np.testing.assert_allclose(host(whole), host(tot), rtol=1e-12)
    # episode cuts really cut: a LAST step's return does not depend on anything after it
    t0 = 123
    cut = fl[t0].bool() & (fl[t0] & _abi.FLAG_LAST).bool()
    if mode == _abi.SCAN_RETURN and bool(cut.any()):
        boot = torch.where((fl[t0] & _abi.FLAG_ABSORBING).bool(), torch.zeros_like(vn[t0]), vn[t0])
        want = ((0.99 * torch.ones((), dtype=torch.float32, device="cuda")) * boot).double() + r[t0].double()
        assert torch.equal(ret[t0][cut], want.float()[cut])


def test_scan_golden_ppo(eng, golden):
    g = golden("ppo_returns.npz")
    L = g["ep_len"]
    n = int(L.sum())
    flags = np.zeros(n, np.uint8)
    nv = np.zeros(n, np.float32)
    end = np.cumsum(L) - 1
    for e, dn, lv in zip(end, g["done_tail"], g["last_val"]):
        flags[e] = _abi.FLAG_LAST | (_abi.FLAG_ABSORBING if dn else 0)
        nv[e] = lv
    ret, adv = eng.return_scan(_abi.SCAN_RETURN, float(g["gamma"]), 0.95, dev(g["rewards"][:, None]),
                               dev(g["values"][:, None]), dev(nv[:, None]), dev(flags[:, None]))
    assert np.array_equal(host(ret)[:, 0], g["returns"])       # bit-exact vs PPOBuffer.finish_path
    assert np.array_equal(host(adv)[:, 0], g["adv"])
    st = eng.adv_stats(adv)
    eng.adv_normalize(adv, st, 1, float(g["eps"]))
    np.testing.assert_allclose(host(adv)[:, 0], g["adv_norm"], rtol=2e-6, atol=2e-7)   # torch f32 mean/std


def test_scan_golden_ppo_float64_rewards(eng, golden, oracle):
    """env.step's reward is float64 and PPOBuffer never narrows it (rl/envs/wrappers.py:14,
    rl/algos/ppo.py:74-76): with the reward handed over as float64 the returns are bit-exact for
    rewards float32 cannot represent (the f32 path is exact only for f32 rewards)."""
    g = golden("ppo_returns_f64.npz")
    L = g["ep_len"]
    n = int(L.sum())
    flags = np.zeros(n, np.uint8)
    nv = np.zeros(n, np.float32)
    for e, dn, lv in zip(np.cumsum(L) - 1, g["done_tail"], g["last_val"]):
        flags[e] = _abi.FLAG_LAST | (_abi.FLAG_ABSORBING if dn else 0)
        nv[e] = lv
    # [n,1] takes the scalar-load kernel, [n,4] (four copies side by side) the pipelined one
    for reps in (1, 4):
        tile = lambda a: dev(np.repeat(a[:, None], reps, axis=1))
        ret, adv = eng.return_scan(_abi.SCAN_RETURN, float(g["gamma"]), 0.95, tile(g["rewards"]), tile(g["values"]),
                                   tile(nv), tile(flags))
        for c in range(reps):
            assert np.array_equal(host(ret)[:, c], g["returns"])   # bit-exact vs PPOBuffer.finish_path
            assert np.array_equal(host(adv)[:, c], g["adv"])


@pytest.mark.parametrize("T,N", [(5, 3), (400, 4096), (97, 1028), (70, 16388), (45, 32776)])
def test_scan_float64_rewards_vs_oracle(eng, oracle, T, N):
    rng = np.random.default_rng(T + N)
    r = rng.uniform(-0.3, 1, (T, N))
    v, vn = (rng.normal(0, 1, (T, N)).astype(np.float32) for _ in range(2))
    last = rng.uniform(size=(T, N)) < 1 / 30
    flags = (last * _abi.FLAG_LAST + (last & (rng.uniform(size=(T, N)) < 0.5)) * _abi.FLAG_ABSORBING).astype(np.uint8)
    ret, adv = eng.return_scan(_abi.SCAN_RETURN, 0.99, 0.0, dev(r), dev(v), dev(vn), dev(flags))
    e_ret, e_adv = oracle.return_scan_r64(0.99, r, v, vn, flags)
    assert np.array_equal(host(ret), e_ret) and np.array_equal(host(adv), e_adv)


@pytest.mark.parametrize("mode", [_abi.SCAN_RETURN, _abi.SCAN_GAE])
@pytest.mark.parametrize("T,N", [(1, 4), (400, 4096), (97, 1028), (33, 257), (70, 16388), (45, 32776)])
def test_scan_fused_statistics(eng, oracle, mode, T, N):
    """oly_return_scan_stats: ret / adv unchanged (bit-exact vs the oracle) and, from the same
    pass, (count, sum adv, sum adv^2): fp64 sums in a fixed order, 1e-12 relative to the oracle's
    sequential sums and bit-identical run to run."""
    rng = np.random.default_rng(T * 3 + N)
    r, v, vn = (rng.normal(0, 1, (T, N)).astype(np.float32) for _ in range(3))
    last = rng.uniform(size=(T, N)) < 1 / 100
    flags = (last * _abi.FLAG_LAST).astype(np.uint8)
    st = torch.zeros(3, dtype=torch.float64, device="cuda")
    ret, adv = eng.return_scan(mode, 0.99, 0.97, dev(r), dev(v), dev(vn), dev(flags), stats3=st)
    e_ret, e_adv = oracle.return_scan(mode, 0.99, 0.97, r, v, vn, flags)
    assert np.array_equal(host(ret), e_ret) and np.array_equal(host(adv), e_adv)
    e_st = oracle.adv_stats(e_adv)
    got = host(st)
    assert got[0] == T * N
    np.testing.assert_allclose(got[1:], e_st[1:], rtol=1e-12, atol=1e-9)
    st2 = torch.zeros(3, dtype=torch.float64, device="cuda")
    eng.return_scan(mode, 0.99, 0.97, dev(r), dev(v), dev(vn), dev(flags), stats3=st2)
    assert torch.equal(st, st2)                                  # deterministic


def test_scan_lane_kernel_on_every_scan_case():
    """The lane-per-environment scan (taken by default from 256 environments per CU, i.e. shapes no test here
    reaches) forced onto every scan test of this file: OLY_K6_PIPE is read once per process, hence the child."""
    import subprocess
    import sys
    env = dict(os.environ, OLY_K6_PIPE="7")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                          "scan and not lane_kernel", "-p", "no:cacheprovider"], capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout


def test_scan_lane_kernel_wide_shape(eng, oracle):
    """A shape the default dispatch sends to the lane kernel ([24, 70000]: N >= 256 per CU), float64 rewards and
    fused statistics, against the oracle (bit-exact returns / advantages)."""
    rng = np.random.default_rng(77)
    T, N = 24, 70000
    r = rng.normal(0.3, 1, (T, N))
    v, vn = (rng.normal(0, 1, (T, N)).astype(np.float32) for _ in range(2))
    last = rng.uniform(size=(T, N)) < 0.05
    flags = (last * _abi.FLAG_LAST + (last & (rng.uniform(size=(T, N)) < 0.5)) * _abi.FLAG_ABSORBING).astype(np.uint8)
    st = torch.zeros(3, dtype=torch.float64, device="cuda")
    ret, adv = eng.return_scan(_abi.SCAN_RETURN, 0.99, 1.0, dev(r), dev(v), dev(vn), dev(flags), stats3=st)
    e_ret, e_adv = oracle.return_scan_r64(0.99, r, v, vn, flags)
    assert np.array_equal(host(ret), e_ret) and np.array_equal(host(adv), e_adv)
    got = host(st)
    assert got[0] == T * N
    np.testing.assert_allclose(got[1:], oracle.adv_stats(e_adv)[1:], rtol=1e-12, atol=1e-9)


def test_adv_normalize_parts(eng, oracle):
    """The multi-rank normalisation: [parts,3] triples combined on the device by the balanced
    rank-order tree = the oracle's; 1 part = oly_adv_normalize."""
    rng = np.random.default_rng(8)
    x = rng.normal(0.2, 1.7, 400 * 1024).astype(np.float32)
    for parts in (1, 2, 3, 8, 37, 64):                       # 64 = OLY_MAX_STAT_PARTS
        shards = np.array_split(x, parts)
        p3 = np.stack([host(eng.adv_stats(dev(s))) for s in shards])
        for ddof, eps in ((1, 1e-5), (0, 1e-8)):
            got = host(eng.adv_normalize(dev(x), dev(p3) if parts > 1 else dev(p3[0]), ddof, eps))
            assert np.array_equal(got, oracle.adv_normalize_parts(x, p3, ddof, eps))
            np.testing.assert_allclose(got, (x - x.mean()) / (x.std(ddof=ddof) + eps), rtol=2e-5, atol=2e-6)
    from olympic_hip._ffi import OlyError
    with pytest.raises(OlyError, match="parts"):
        eng.adv_normalize(dev(x), dev(np.zeros((65, 3))), 1, 1e-5)


# --------------------------------------------------------------------------------- K7
@pytest.mark.parametrize("n", [1, 3, 1000, 4096 * 400 + 3])
def test_adv_stats_normalize(eng, oracle, n):
    rng = np.random.default_rng(n)
    x = (rng.normal(0.3, 2.0, n)).astype(np.float32)
    xd = dev(x)
    st = eng.adv_stats(xd)
    e_st = oracle.adv_stats(x)
    s = host(st)
    assert s[0] == n
    np.testing.assert_allclose(s[1:], e_st[1:], rtol=1e-12)      # fp64 sums, different tree order
    st2 = eng.adv_stats(xd)
    assert torch.equal(st, st2)                                   # deterministic run to run
    if n > 1:
        for ddof, eps in ((1, 1e-5), (0, 1e-8)):
            y = eng.adv_normalize(xd.clone(), st, ddof, eps)
            e_y = oracle.adv_normalize(x, s, ddof, eps)
            assert np.array_equal(host(y), e_y)                   # same stats -> bit-exact
            xt = torch.as_tensor(x)
            ref = (xt - xt.mean()) / (xt.std(unbiased=bool(ddof)) + eps)
            np.testing.assert_allclose(host(y), ref.numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("B,D", [(1, 32), (1000, 32), (4096, 41), (513, 12), (50000, 32)])
def test_col_stats(eng, oracle, B, D):
    rng = np.random.default_rng(B + D)
    x = rng.normal(0.5, 1.5, (B, D)).astype(np.float32)
    cs = eng.col_stats(dev(x))
    e = oracle.col_stats(x)
    np.testing.assert_allclose(host(cs), e, rtol=1e-12)
    cs2 = eng.col_stats(dev(x), cs.clone())
    np.testing.assert_allclose(host(cs2), 2 * e, rtol=1e-12)


# --------------------------------------------------------------------------------- K8
def test_disc_golden(eng, golden, oracle):
    g = golden("vail_disc.npz")
    d = g["d"].reshape(-1)
    r = host(eng.disc_reward(dev(d)))
    p = 1.0 / (1.0 + np.exp(-d.astype(np.float64)))
    tol = 4 * 2.0 ** -24 / (1 - p + 1e-8) + 4e-7 * np.abs(g["reward"]) + 1e-7   # see test_oracle_golden
    assert (np.abs(r - g["reward"]) <= tol).all()
    re = host(eng.disc_reward(dev(g["d_ext"].reshape(-1))))
    pe = 1.0 / (1.0 + np.exp(-g["d_ext"].reshape(-1).astype(np.float64)))
    tole = 4 * 2.0 ** -24 / (1 - pe + 1e-8) + 4e-7 * np.abs(g["reward_ext"]) + 1e-7
    assert (np.abs(re - g["reward_ext"]) <= tole).all() and np.isfinite(re).all()
    xs = host(eng.disc_standardize(dev(g["x"]), dev(np.arange(32, dtype=np.int32)), dev(g["st_mean"]),
                                   dev(g["st_std"])))
    assert np.array_equal(xs, oracle.disc_standardize(g["x"], np.arange(32), g["st_mean"], g["st_std"]))
    z = host(eng.disc_reparam(dev(g["mu"]), dev(g["logvar"]), dev(g["eps"])))
    np.testing.assert_allclose(z, oracle.disc_reparam(g["mu"], g["logvar"], g["eps"]), rtol=3e-7, atol=1e-7)


@pytest.mark.parametrize("n,offset", [(1, 0), (3, 0), (4, 0), (4099, 0), (4099, 1), (65536, 0), (65537, 3)])
def test_disc_reward_vector_and_scalar_paths(eng, oracle, n, offset):
    """float4 body + scalar tail, and the scalar kernel for buffers that are not 16-byte aligned: same
    float32 formula as the oracle (device expf / logf vs glibc's: the stated float32 tolerance)."""
    rng = np.random.default_rng(n + offset)
    d = np.concatenate([rng.normal(0, 4, n - min(n, 3)), [-30.0, 0.0, 25.0][:min(n, 3)]]).astype(np.float32)
    buf, out = dev(np.zeros(n + 8, np.float32)), dev(np.full(n + 8, 7.0, np.float32))
    buf[offset:offset + n] = dev(d)
    r = host(eng.disc_reward(buf[offset:offset + n], out[offset:offset + n]))
    want = oracle.disc_reward(d)
    p = 1.0 / (1.0 + np.exp(-d.astype(np.float64)))
    tol = 4 * 2.0 ** -24 / (1 - p + 1e-8) + 4e-7 * np.abs(want) + 1e-7
    assert (np.abs(r - want) <= tol).all() and np.isfinite(r).all()
    o = host(out)
    assert (o[:offset] == 7.0).all() and (o[offset + n:] == 7.0).all()          # nothing outside the slice


def test_vail_reward_end_to_end(eng, golden):
    """mask+standardise (HIP) -> encoder/decoder GEMMs (torch-ROCm) -> reparam + reward (HIP)
    reproduces the reference VariationalNet + make_discrim_reward."""
    g = golden("vail_disc.npz")
    t = lambda k: dev(g[k])
    x = t("x")
    cs = eng.col_stats(x)
    cnt = cs[0] + 1e-2
    mean = cs[1] / cnt
    std = torch.sqrt(torch.clamp((cs[2] + 1e-2) / cnt - mean ** 2, min=1e-2))
    np.testing.assert_allclose(host(mean), g["st_mean"], rtol=2e-5, atol=2e-6)
    xs = eng.disc_standardize(x, None, mean, std)
    with torch.no_grad():
        h = torch.relu(xs @ t("enc_w0").T + t("enc_b0"))
        h = torch.relu(h @ t("enc_w1").T + t("enc_b1"))
        mu, lv = h @ t("mu_w").T + t("mu_b"), h @ t("lv_w").T + t("lv_b")
        z = eng.disc_reparam(mu.contiguous(), lv.contiguous(), t("eps"))
        d = (z @ t("dec_w").T + t("dec_b")).reshape(-1).contiguous()
    np.testing.assert_allclose(host(d), g["d"].reshape(-1), rtol=2e-3, atol=2e-3)    # GEMM order
    r = host(eng.disc_reward(d))
    ok = np.abs(g["d"].reshape(-1)) < 8
    np.testing.assert_allclose(r[ok], g["reward"][ok], rtol=5e-3, atol=5e-3)


# --------------------------------------------------------------------------------- K12
def _disc_weights(rng, D, scale=1.0):
    """Random discriminator of the reference's shape; logvar head large enough that exp32 matters."""
    f = lambda *s: (rng.normal(0, 1, s) * scale).astype(np.float32)
    return dict(enc_w0=f(256, D) / np.float32(np.sqrt(D)), enc_b0=f(256) * np.float32(0.1),
                enc_w1=f(128, 256) / np.float32(16), enc_b1=f(128) * np.float32(0.1),
                mu_w=f(128, 128) / np.float32(11), mu_b=f(128) * np.float32(0.1),
                lv_w=f(128, 128) / np.float32(11), lv_b=f(128) * np.float32(0.1),
                dec_w=f(1, 128) / np.float32(11), dec_b=f(1))


_DISC_KEYS = ("enc_w0", "enc_b0", "enc_w1", "enc_b1", "mu_w", "mu_b", "lv_w", "lv_b", "dec_w", "dec_b")


def _disc_pack(eng, w):
    return eng.disc_pack(*[dev(w[k]) for k in _DISC_KEYS])


@pytest.mark.parametrize("B,Dx,D,stats", [(1, 32, 32, "meanstd"), (31, 32, 32, "colstats"), (32, 32, 32, "none"),
                                          (33, 32, 32, "meanstd"), (512, 36, 30, "meanstd"), (4099, 32, 32, "colstats"),
                                          (777, 40, 34, "colstats"), (300, 64, 64, "meanstd"), (65, 7, 5, "none")])
def test_disc_forward_vs_oracle(eng, oracle, B, Dx, D, stats):
    """K12 against its oracle twin: every output BIT-EXACT (f32 fma chains on the matrix cores, the fixed
    exp32, the fp64-rounded reward steps), over ragged tiles, both layer-1 widths, a state mask, the three
    standardisation modes and the no-noise forward."""
    rng = np.random.default_rng(B * 131 + D)
    w = _disc_weights(rng, D)
    x = rng.normal(0.3, 2.0, (B, Dx)).astype(np.float32)
    mask = None if D == Dx else np.sort(rng.choice(Dx, D, replace=False)).astype(np.int32)
    eps = rng.normal(0, 1, (B, 128)).astype(np.float32)
    kw, okw = {}, {}
    if stats == "meanstd":
        mean, std = rng.normal(0.3, 0.2, D), rng.uniform(0.5, 2.5, D)
        kw, okw = dict(mean=dev(mean), std=dev(std)), dict(mean=mean, std=std)
    elif stats == "colstats":
        xm = x if mask is None else x[:, mask]
        cs = np.stack([np.full(D, float(B)), xm.astype(np.float64).sum(0), (xm.astype(np.float64) ** 2).sum(0)])
        kw, okw = dict(colstats=dev(cs)), dict(colstats=cs)
    packed = _disc_pack(eng, w)
    want = ("reward", "logits", "mu", "logvar")
    for e in (eps, None):
        o = eng.disc_forward(dev(x), packed, mask=None if mask is None else dev(mask), eps=None if e is None else dev(e),
                             want=want, **kw)
        ref = oracle.disc_forward(x, w, mask=mask, eps=e, **okw)
        for k in want:
            assert np.array_equal(host(o[k]), ref[k]), (k, e is None)
    # outputs are optional and independent: the reward alone equals the reward of the full call
    r = eng.disc_forward(dev(x), packed, mask=None if mask is None else dev(mask), eps=dev(eps), **kw)["reward"]
    assert np.array_equal(host(r), oracle.disc_forward(x, w, mask=mask, eps=eps, **okw)["reward"])


def test_disc_forward_golden(eng, golden, oracle):
    """K12 on the reference run of vail_disc.npz (VariationalNet + Standardizer + make_discrim_reward executed
    by gen_golden.py): logits within the summation-order tolerance of the Linear layers, reward within that
    plus the stated float32 tolerance of the formula; and the statistics-from-running-sums mode."""
    g = golden("vail_disc.npz")
    packed = _disc_pack(eng, g)
    o = eng.disc_forward(dev(g["x"]), packed, mask=dev(np.arange(32, dtype=np.int32)), mean=dev(g["st_mean"]),
                         std=dev(g["st_std"]), eps=dev(g["eps"]), want=("reward", "logits", "mu", "logvar"))
    ref = oracle.disc_forward(g["x"], g, mask=np.arange(32), mean=g["st_mean"], std=g["st_std"], eps=g["eps"])
    for k in ref:
        assert np.array_equal(host(o[k]), ref[k]), k
    d = g["d"].reshape(-1)
    np.testing.assert_allclose(host(o["mu"]), g["mu"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(host(o["logvar"]), g["logvar"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(host(o["logits"]), d, rtol=2e-5, atol=2e-6)
    p = 1.0 / (1.0 + np.exp(-d.astype(np.float64)))
    tol = (2e-6 + 2e-5 * np.abs(d)) + 4 * 2.0 ** -24 / (1 - p + 1e-8) + 4e-7 * np.abs(g["reward"]) + 1e-7
    assert (np.abs(host(o["reward"]) - g["reward"]) <= tol).all()
    cs = eng.col_stats(dev(g["x"]))                      # Standardizer.forward: update, then standardise
    o2 = eng.disc_forward(dev(g["x"]), packed, colstats=cs, eps=dev(g["eps"]), want=("logits",))
    np.testing.assert_allclose(host(o2["logits"]), d, rtol=1e-4, atol=1e-5)
    assert np.array_equal(host(o2["logits"]), oracle.disc_forward(g["x"], g, colstats=host(cs), eps=g["eps"])["logits"])


def test_disc_forward_edge_values(eng, oracle):
    """NaN / infinity rows stay confined to their own sample, saturated logits give the clamped rewards
    (-log(1e-8) and -log(1 + 1e-8) = 0 in float32), a huge logvar overflows like exp does."""
    rng = np.random.default_rng(5)
    w = _disc_weights(rng, 32)
    x = rng.normal(0, 1, (70, 32)).astype(np.float32)
    x[3, 5], x[40, 0], x[41, 31] = np.nan, np.inf, -1e30
    eps = rng.normal(0, 1, (70, 128)).astype(np.float32)
    for scale in (1.0, 60.0):                             # 60: |d| in the hundreds, logvar / 2 beyond +-88
        ws = dict(w, dec_w=w["dec_w"] * np.float32(scale), lv_w=w["lv_w"] * np.float32(scale))
        o = eng.disc_forward(dev(x), _disc_pack(eng, ws), eps=dev(eps), want=("reward", "logits", "mu", "logvar"))
        ref = oracle.disc_forward(x, ws, eps=eps)
        for k in ref:
            assert np.array_equal(host(o[k]), ref[k], equal_nan=True), (k, scale)
        good = np.ones(70, bool)
        good[[3, 40]] = False
        assert np.isfinite(host(o["mu"])[good]).all() and np.isnan(host(o["logits"])[3])


def test_disc_forward_full_size_properties(eng, oracle):
    """BASELINE config 4 at full size ([400, 4096] samples through one launch): samples are independent, so
    (a) any row subset run alone reproduces its rows bit for bit, (b) a row permutation permutes the outputs,
    (c) 2048 random rows equal the oracle."""
    rng = np.random.default_rng(11)
    B = 400 * 4096
    w = _disc_weights(rng, 32)
    packed = _disc_pack(eng, w)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((B, 32), device="cuda", generator=g) * 1.5 + 0.2
    eps = torch.randn((B, 128), device="cuda", generator=g)
    cs = eng.col_stats(x)
    want = ("reward", "logits")
    full = eng.disc_forward(x, packed, colstats=cs, eps=eps, want=want)
    r, d = host(full["reward"]), host(full["logits"])
    assert np.isfinite(r).all() and (r >= 0).all()
    lo, hi = 123 * 32 + 7, 123 * 32 + 7 + 100001                      # an unaligned row range
    part = eng.disc_forward(x[lo:hi].contiguous(), packed, colstats=cs, eps=eps[lo:hi].contiguous(), want=want)
    assert np.array_equal(host(part["reward"]), r[lo:hi]) and np.array_equal(host(part["logits"]), d[lo:hi])
    perm = torch.randperm(B, device="cuda", generator=g)
    pp = eng.disc_forward(x[perm].contiguous(), packed, colstats=cs, eps=eps[perm].contiguous(), want=want)
    assert np.array_equal(host(pp["reward"]), r[host(perm)])
    idx = np.sort(rng.choice(B, 2048, replace=False))
    ref = oracle.disc_forward(host(x[idx]), w, colstats=host(cs), eps=host(eps[idx]))
    assert np.array_equal(ref["logits"], d[idx]) and np.array_equal(ref["reward"], r[idx])


@pytest.mark.parametrize("rows", [16, 32])
def test_disc_forward_both_tile_heights_on_every_case(rows):
    """oly_disc_forward picks 16-row tiles (v_mfma_f32_16x16x4_f32, 256 tiles for a batch of 4096) for small batches
    and 32-row tiles (v_mfma_f32_32x32x2_f32) for large ones; OLY_K12_ROWS forces one kernel onto every K12 test of
    this file (the knob is read once per process, hence the child): both are bit-exact against the same oracle."""
    import subprocess
    import sys
    env = dict(os.environ, OLY_K12_ROWS=str(rows))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                          "disc_forward and not both_tile_heights", "-p", "no:cacheprovider"], capture_output=True,
                         text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


def test_discriminator_reward_fused_matches_layer_by_layer(eng, golden):
    """gail.DiscriminatorReward: the fused launch against the PyTorch-GEMM path it replaces (summation order
    of the Linear layers is the only difference), with the running statistics updated once per call by both,
    and a re-pack after an optimiser step."""
    from olympic_hip.gail import DiscriminatorReward, VariationalDiscriminator
    g = golden("vail_disc.npz")
    net = VariationalDiscriminator().load_reference_arrays(g).cuda()
    fused = DiscriminatorReward(eng, net, state_mask=np.arange(32))
    plain = DiscriminatorReward(eng, net, state_mask=np.arange(32))
    assert fused.fused
    x, eps = dev(g["x"]), dev(g["eps"])
    for it in range(2):                                    # the second call standardises with two batches of statistics
        r = host(fused(x, eps))
        d_ref, _, _ = plain.logits_unfused(x, eps)
        r_ref = host(eng.disc_reward(d_ref))
        assert np.array_equal(host(fused.stand.colstats), host(plain.stand.colstats))
        pr = 1.0 / (1.0 + np.exp(-host(d_ref).astype(np.float64)))
        # the two paths differ in the summation order of the Linear layers: |delta d| <= 5e-6 + 3e-5 |d| goes
        # through dr/dd = sigmoid(d) <= 1; then the float32 steps of the formula (see test_disc_golden)
        tol = (5e-6 + 3e-5 * np.abs(host(d_ref))) + 4 * 2.0 ** -24 / (1 - pr + 1e-8) + 4e-7 * np.abs(r_ref) + 1e-7
        assert (np.abs(r - r_ref) <= tol).all(), np.abs(r - r_ref).max()
        if it == 0:
            np.testing.assert_allclose(r, g["reward"], rtol=5e-3, atol=5e-3)
    with torch.no_grad():
        net.decoder.bias.add_(0.5)                         # what an optimiser step does: in-place, version bump
    r2 = host(fused(x, eps))
    plain.logits_unfused(x, eps)
    assert np.abs(r2 - r).max() > 1e-3
    d3, _, _ = plain.logits_unfused(x, eps)                # both standardisers have seen four batches now
    np.testing.assert_allclose(host(fused.logits(x, eps)[0]), host(d3), rtol=2e-5, atol=1e-4)   # |d| up to 200: cancellation


# --------------------------------------------------------------------------------- K4
def test_traj_golden(eng, golden):
    g = golden("trajectory.npz")
    table = g["table"]
    eng.traj_upload(table)
    resets = g["resets"]
    ct, cs, origin, sample = eng.traj_reset(dev(resets[:, 1], torch.int32), dev(resets[:, 0], torch.int32))
    assert np.array_equal(host(sample), g["reset_samples"])
    sub, tno = g["rnd_reset"]
    ct, cs, origin, sample = eng.traj_reset(dev([tno], torch.int32), dev([sub], torch.int32))
    walk = [host(sample)[0].copy()]
    L = table.shape[2]
    for _ in range(L + 2):
        at_end = eng.traj_next(ct, cs, origin, sample)
        if host(at_end)[0]:
            assert host(cs)[0] == L
            break
        walk.append(host(sample)[0].copy())
    assert np.array_equal(np.array(walk), g["walk"])


def test_traj_vs_oracle_many_envs(eng, golden, oracle):
    g = golden("trajectory.npz")
    table = g["table"]
    K, J, L = table.shape
    eng.traj_upload(table)
    N = 3000
    rng = np.random.default_rng(4)
    tn, st = rng.integers(0, J, N).astype(np.int32), rng.integers(0, L, N).astype(np.int32)
    ct, cs, origin, sample = eng.traj_reset(dev(tn), dev(st))
    e_ct, e_cs, e_or, e_sa = oracle.traj_reset(table, tn, st)
    assert np.array_equal(host(sample), e_sa) and np.array_equal(host(origin), e_or)
    active = (rng.uniform(size=N) < 0.7).astype(np.uint8)
    for _ in range(4):
        cur = rng.normal(size=(N, 17))
        eng.traj_euler(17, 0.01, dev(cur), sample)
        e_sa = oracle.traj_euler(17, 0.01, cur, e_sa)
        assert np.array_equal(host(sample), e_sa)
        at_end = eng.traj_next(ct, cs, origin, sample, active=dev(active))
        e_cs, e_sa, e_end = oracle.traj_next(table, e_ct, e_cs, e_or, e_sa, active=active)
        assert np.array_equal(host(cs), e_cs) and np.array_equal(host(at_end), e_end)
        assert np.array_equal(host(sample), e_sa)


# --------------------------------------------------------------------------------- K3
def test_contacts_golden_and_oracle(eng, golden, oracle):
    g = golden("contacts.npz")
    eng.contact_configure(g["geom_bodyid"], int(g["floor_body"]), int(g["rfoot_body"]), int(g["lfoot_body"]))
    pz = np.ascontiguousarray(g["pos"][:, :, 2])
    o = eng.contact_reduce(dev(g["ncon"]), dev(g["geom1"]), dev(g["geom2"]), dev(g["force6"]), dev(pz))
    o = {k: host(v) for k, v in o.items()}
    for k in ("n_r", "n_l", "idx_r", "idx_l"):
        assert np.array_equal(o[k], g[k]), k                          # bit-exact integers
    assert np.array_equal(o["bad"].astype(bool), g["bad"])
    e = oracle.contact_reduce(g["geom_bodyid"], 0, 7, 10, g["ncon"], g["geom1"], g["geom2"], g["force6"], pz)
    for k in ("grf_r", "grf_l", "min_z"):
        assert np.array_equal(o[k], e[k]), k                          # same in-order chain: bit-exact
    np.testing.assert_allclose(o["grf_r"], g["grf_r"], rtol=1e-14)


@pytest.mark.parametrize("N,C", [(1, 16), (5, 7), (4097, 16), (300, 40)])
def test_contacts_shapes(eng, oracle, N, C):
    rng = np.random.default_rng(N + C)
    gb = np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32)
    eng.contact_configure(gb, 0, 7, 10)
    ncon = np.minimum(rng.poisson(C / 3, N), C).astype(np.int32)
    g1 = rng.choice([0, 0, 0, 8, 3], (N, C)).astype(np.int32)
    g2 = rng.choice([8, 12, 8, 12, 0, 5], (N, C)).astype(np.int32)
    f6 = rng.normal(0, 100, (N, C, 6))
    pz = rng.uniform(-0.05, 0.05, (N, C))
    o = eng.contact_reduce(dev(ncon), dev(g1), dev(g2), dev(f6), dev(pz))
    e = oracle.contact_reduce(gb, 0, 7, 10, ncon, g1, g2, f6, pz)
    for k in e:
        assert np.array_equal(host(o[k]), e[k]), k


@pytest.mark.parametrize("N,C", [(1, 16), (130, 16), (4096, 16), (777, 5), (300, 40)])
def test_contact_reduce_csr_equals_the_padded_form(eng, N, C):
    """oly_contact_reduce_csr (compact 64-byte records behind per-env offsets: what the RL host batcher ships in
    compact mode) against oly_contact_reduce on the same contacts in padded slots: counts, force sums, minimum
    height and the bad flag bit-identical, including environments with more contacts than the C slots."""
    import ctypes
    rng = np.random.default_rng(N + C)
    ncon = rng.integers(0, C + 3, N).astype(np.int32)                # some beyond C: bad, truncated to C
    ncon[rng.uniform(size=N) < 0.1] = 0
    g1 = rng.choice([0, 0, 0, 3], (N, C)).astype(np.int32)
    g2 = rng.choice([7, 8, 10, 2, 99, -1], (N, C)).astype(np.int32)
    f6 = rng.normal(0, 100, (N, C, 6))
    pz = rng.normal(0, 0.05, (N, C))
    eng.contact_configure(np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32), 0, 7, 10)
    ref = eng.contact_reduce(dev(ncon), dev(g1), dev(g2), dev(f6), dev(pz), want_idx=False)
    used = np.clip(ncon, 0, C)
    coff = (np.cumsum(used) - used).astype(np.int32)
    rec = np.zeros(int(used.sum()) + 1, np.dtype([("g1", "<i4"), ("g2", "<i4"), ("f", "<f8", 6), ("z", "<f8")]))
    assert rec.dtype.itemsize == ctypes.sizeof(_abi.ContactRecord) == 64
    for n in range(N):
        k = used[n]
        sl = slice(coff[n], coff[n] + k)
        rec["g1"][sl], rec["g2"][sl], rec["f"][sl], rec["z"][sl] = g1[n, :k], g2[n, :k], f6[n, :k], pz[n, :k]
    got = eng.contact_reduce_csr(dev(ncon), dev(coff), dev(rec.view(np.uint8).reshape(-1)), C)
    for k in ("n_r", "n_l", "grf_r", "grf_l", "min_z", "bad"):
        assert torch.equal(got[k], ref[k]), k
    assert int(ref["bad"].sum()) > 0 or N == 1
    # offsets are device data: one that points past the records (or before them) is never dereferenced, the
    # environment comes back bad; the other environments are unaffected
    if N > 2:
        coff2 = coff.copy()
        victims = [i for i in (1, N - 1) if ncon[i] > 0][:2]
        for i, off in zip(victims, (len(rec), -5)):                   # first record index == n_records; negative
            coff2[i] = off
        got2 = eng.contact_reduce_csr(dev(ncon), dev(coff2), dev(rec.view(np.uint8).reshape(-1)), C)
        ok = np.ones(N, bool)
        ok[victims] = False
        for k in ("n_r", "n_l", "grf_r", "grf_l", "min_z", "bad"):
            assert torch.equal(got2[k][torch.as_tensor(ok)], ref[k][torch.as_tensor(ok)]), k
        assert all(int(got2["bad"][i]) == 1 for i in victims)


def test_contacts_more_than_the_staged_slots(eng):
    """ncon > C (or < 0): the surplus contacts were never staged, so the environment is flagged as
    a bad collision instead of being reduced as if it had C contacts; the staged slots
    are still counted.  (The oracle refuses such lists with OLY_ERANGE.)"""
    gb = np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32)
    eng.contact_configure(gb, 0, 7, 10)
    N, C = 6, 4
    ncon = np.array([4, 5, 9, -1, 0, 2], np.int32)
    g1 = np.zeros((N, C), np.int32)                       # floor first
    g2 = np.full((N, C), 8, np.int32)                     # right foot: every staged slot is a foot contact
    f6 = np.ones((N, C, 6))
    o = eng.contact_reduce(dev(ncon), dev(g1), dev(g2), dev(f6), dev(np.zeros((N, C))))
    assert host(o["n_r"]).tolist() == [4, 4, 4, 0, 0, 2]
    assert host(o["bad"]).tolist() == [0, 1, 1, 1, 0, 0]


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("OLY_FUZZ", "12"))))
def test_contacts_random(eng, oracle, seed):
    """Random slot counts and densities (none .. every slot a foot contact), valid geom ids and
    ncon <= C (the oracle rejects malformed contact lists with OLY_ERANGE; the kernel, which
    cannot raise, ignores such slots): all outputs bit-exact, force sums in contact order."""
    rng = np.random.default_rng(500 + seed)
    N, C = int(rng.integers(1, 3000)), int(rng.integers(1, 50))
    gb = np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32)
    eng.contact_configure(gb, 0, 7, 10)
    ncon = rng.integers(0, C + 1, N).astype(np.int32)
    p_floor = float(rng.choice([0.0, 0.3, 0.9, 1.0]))
    g1 = np.where(rng.uniform(size=(N, C)) < p_floor, 0, rng.integers(0, 13, (N, C))).astype(np.int32)
    g2 = rng.choice([7, 8, 11, 12, 3, 0, 5, 9], (N, C)).astype(np.int32)
    f6 = rng.normal(0, 100, (N, C, 6))
    pz = rng.uniform(-0.05, 0.05, (N, C))
    o = eng.contact_reduce(dev(ncon), dev(g1), dev(g2), dev(f6), dev(pz))
    e = oracle.contact_reduce(gb, 0, 7, 10, ncon, g1, g2, f6, pz)
    for k in e:
        assert np.array_equal(host(o[k]), e[k]), (k, N, C)


# --------------------------------------------------------------------------------- K2
@pytest.fixture(params=[1, 16])
def k2_lanes(request, monkeypatch):
    """oly_a3_step has two kernels, one lane per environment and sixteen (chosen by N; OLY_K2_LANES, read on every
    call, forces one): every K2 test runs under both."""
    monkeypatch.setenv("OLY_K2_LANES", str(request.param))
    return request.param


@pytest.mark.parametrize("N", [1, 15, 16, 17, 1000, 4099, 70001])
@pytest.mark.parametrize("obs_f64", [False, True])
def test_a3_step_lane_layouts_give_identical_bytes(eng, golden, monkeypatch, N, obs_f64):
    """The 16-lane kernel evaluates the same expressions on the same inputs as the lane-per-environment kernel:
    observations, rewards, flags and the task state agree byte for byte (ragged last group, both output types)."""
    g = golden("a3_task.npz")
    eng.a3_configure(specs.A3Spec(mass=41.5), g["clock_lut"])
    rng = np.random.default_rng(N)
    seq_len = rng.choice([1, 2, 5, 20], N).astype(np.int32)
    st_h = dict(phase=rng.integers(0, 88, N).astype(np.int32), t1=rng.integers(-1, 21, N).astype(np.int32),
                t2=rng.integers(-1, 21, N).astype(np.int32), reached_frames=rng.integers(0, 36, N).astype(np.int32),
                target_reached=rng.integers(0, 2, N).astype(np.uint8), mode=rng.integers(0, 4, N).astype(np.int32),
                seq_len=seq_len, sequence=rng.normal(0, 0.3, (N, 20, 4)), goal=np.zeros((N, 8)))
    inp = dict(qpos=rng.normal(0, 1, (N, 25)), qvel=rng.normal(0, 1, (N, 24)), act_len=rng.uniform(-1, 1, (N, 12)),
               act_vel=rng.normal(0, 2, (N, 12)), lf_pos=rng.normal(0, 0.2, (N, 3)), rf_pos=rng.normal(0, 0.3, (N, 3)),
               lf_vel=rng.normal(0, 0.2, (N, 3)), rf_vel=rng.normal(0, 0.2, (N, 3)),
               root_pos=np.array([0, 0, 0.7]) + rng.normal(0, 0.2, (N, 3)), root_quat=rng.normal(0, 1, (N, 4)),
               head_pos=np.array([0, 0, 1.2]) + rng.normal(0, 0.1, (N, 3)), grf_l=rng.uniform(0, 400, N),
               grf_r=rng.uniform(0, 400, N), min_z=rng.uniform(-0.01, 0.03, N), n_r=rng.integers(0, 3, N).astype(np.int32),
               n_l=rng.integers(0, 3, N).astype(np.int32), bad=(rng.uniform(size=N) < 0.1).astype(np.uint8))
    inp["qpos"][: max(N // 7, 1), 3:7] = [0.5, 0.5, -0.5, 0.5]          # the gimbal branch of quat2euler
    d_in = {k: dev(v) for k, v in inp.items()}
    res = {}
    for lanes in (1, 16):
        monkeypatch.setenv("OLY_K2_LANES", str(lanes))
        st_d = {k: dev(v) for k, v in st_h.items()}
        outs = []
        for _ in range(3):
            o = eng.a3_step(d_in, st_d, obs_f64=obs_f64)
            outs.append({k: host(v).copy() for k, v in o.items()})
        res[lanes] = (outs, {k: host(v) for k, v in st_d.items()})
    for a, b in zip(res[1][0], res[16][0]):
        for k in a:
            assert a[k].tobytes() == b[k].tobytes(), (k, N)
    for k in res[1][1]:
        assert res[1][1][k].tobytes() == res[16][1][k].tobytes(), (k, N)


def test_a3_golden_sequence(eng, golden, oracle, k2_lanes):
    g = golden("a3_task.npz")
    spec = specs.A3Spec(mass=float(g["mass"]))
    eng.a3_configure(spec, g["clock_lut"])
    eng.contact_configure(g["geom_bodyid"], int(g["floor_body"]), int(g["rfoot_body"]), int(g["lfoot_body"]))
    E, K = g["phase"].shape
    st_h, _ = a3_fixture_arrays(g, 0)
    st_o = {k: v.copy() for k, v in st_h.items()}
    st_d = {k: dev(v) for k, v in st_h.items()}
    for k in range(K):
        _, inp = a3_fixture_arrays(g, k)
        cr = eng.contact_reduce(dev(g["ncon"][:, k]), dev(g["geom1"][:, k]), dev(g["geom2"][:, k]),
                                dev(g["force6"][:, k]), dev(g["cpos_z"][:, k]), want_idx=False)
        d_in = {n: dev(v) for n, v in inp.items()}
        d_in.update({n: cr[n] for n in ("grf_l", "grf_r", "min_z", "n_r", "n_l", "bad")})
        o = eng.a3_step(d_in, st_d, obs_f64=True)
        # integer task state vs the reference: bit-exact
        assert np.array_equal(host(st_d["phase"]), g["phase"][:, k]), k
        assert np.array_equal(host(st_d["t1"]), g["t1"][:, k]) and np.array_equal(host(st_d["t2"]), g["t2"][:, k])
        assert np.array_equal(host(st_d["target_reached"]).astype(bool), g["target_reached"][:, k])
        assert np.array_equal(host(st_d["reached_frames"]), g["reached_frames"][:, k])
        assert np.array_equal(host(o["done"]).astype(bool), g["done"][:, k])
        # floats vs the reference: fp64 math through device libm, 1e-11 relative
        np.testing.assert_allclose(host(st_d["goal"]), g["goal"][:, k], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(host(o["obs"]), g["obs"][:, k], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(host(o["rew6"]), g["rew6"][:, k], rtol=2e-6, atol=1e-7)    # f32 outputs
        np.testing.assert_allclose(host(o["reward"]), g["reward"][:, k], rtol=2e-6, atol=1e-7)
        # and vs the oracle on the same inputs
        cro = oracle.contact_reduce(g["geom_bodyid"], 0, 7, 10, g["ncon"][:, k], g["geom1"][:, k],
                                    g["geom2"][:, k], g["force6"][:, k], g["cpos_z"][:, k])
        inp.update({n: cro[n] for n in ("grf_l", "grf_r", "min_z", "n_r", "n_l", "bad")})
        eo = oracle.a3_step(spec, g["clock_lut"], inp, st_o)
        assert np.array_equal(host(o["done"]), eo["done"])
        np.testing.assert_allclose(host(o["obs"]), eo["obs"], rtol=1e-13, atol=1e-14)


def test_a3_orientation_analytic_cases(eng, golden, k2_lanes):
    """transforms3d is absent (parity unpinned): get_obs' quat -> euler -> quat with the yaw dropped
    and update_goal_steps' frame change are pinned on closed-form cases through oly_a3_step
    (helpers.a3_analytic_cases: identity, pure yaw, +-90 deg about each axis, the gimbal branch,
    round trips).  The oracle is held to the same cases in test_oracle_golden.py."""
    g = golden("a3_task.npz")
    eng.a3_configure(specs.A3Spec(mass=41.5), g["clock_lut"])
    cases = a3_analytic_cases()
    st_d = {k: dev(v) for k, v in cases["state"].items()}
    o = eng.a3_step({k: dev(v) for k, v in cases["inputs"].items()}, st_d, obs_f64=True)
    check_a3_analytic(host(o["obs"]), host(st_d["goal"]), cases)


@pytest.mark.parametrize("N", [1500, 4096])
def test_a3_many_envs_vs_oracle(eng, golden, oracle, N, k2_lanes):
    g = golden("a3_task.npz")
    spec = specs.A3Spec(mass=41.5)
    eng.a3_configure(spec, g["clock_lut"])
    rng = np.random.default_rng(12)
    E = g["phase"].shape[0]
    pick = rng.integers(0, E, N)
    st_h, _ = a3_fixture_arrays(g, 0)
    st_h = {k: np.ascontiguousarray(v[pick]) for k, v in st_h.items()}
    st_h["phase"] = rng.integers(0, 88, N).astype(np.int32)
    seq = st_h["sequence"]
    tgt = seq[np.arange(N), np.minimum(st_h["t1"], st_h["seq_len"] - 1), :3]
    inp = dict(qpos=np.concatenate([rng.normal(0, 1, (N, 3)), rng.normal(0, 1, (N, 4)), rng.uniform(-1, 1, (N, 18))], 1),
               qvel=rng.normal(0, 1, (N, 24)), act_len=rng.uniform(-1, 1, (N, 12)), act_vel=rng.normal(0, 2, (N, 12)),
               lf_pos=tgt + rng.normal(0, 0.15, (N, 3)), rf_pos=tgt + rng.normal(0, 0.3, (N, 3)),
               lf_vel=rng.normal(0, 0.2, (N, 3)), rf_vel=rng.normal(0, 0.2, (N, 3)),
               root_pos=tgt + np.array([0, 0, 0.7]) + rng.normal(0, 0.2, (N, 3)),
               root_quat=rng.normal(0, 1, (N, 4)), head_pos=tgt + np.array([0, 0, 1.2]) + rng.normal(0, 0.1, (N, 3)),
               grf_l=rng.uniform(0, 400, N), grf_r=rng.uniform(0, 400, N), min_z=rng.uniform(-0.01, 0.03, N),
               n_r=rng.integers(0, 3, N).astype(np.int32), n_l=rng.integers(0, 3, N).astype(np.int32),
               bad=(rng.uniform(size=N) < 0.2).astype(np.uint8))
    st_d = {k: dev(v) for k, v in st_h.items()}
    st_o = {k: v.copy() for k, v in st_h.items()}
    for _ in range(3):
        o = eng.a3_step({k: dev(v) for k, v in inp.items()}, st_d, obs_f64=True)
        eo = oracle.a3_step(spec, g["clock_lut"], inp, st_o)
        for k in ("phase", "t1", "t2", "reached_frames", "target_reached"):
            assert np.array_equal(host(st_d[k]), st_o[k]), k
        assert np.array_equal(host(o["done"]), eo["done"])
        np.testing.assert_allclose(host(st_d["goal"]), st_o["goal"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(host(o["obs"]), eo["obs"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(host(o["reward"]), eo["reward"], rtol=2e-6, atol=1e-7)
        inp["lf_pos"] = inp["lf_pos"] + rng.normal(0, 0.02, (N, 3))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("OLY_FUZZ", "12"))))
def test_a3_state_machine_random(eng, golden, oracle, seed, k2_lanes):
    """WalkingTask's integer state machine at its edges: frame counters around the 30-frame delay,
    target indices at the end of short sequences, all walk modes, feet inside / outside the
    target radius; integer state and done bit-exact over several steps."""
    g = golden("a3_task.npz")
    spec = specs.A3Spec(mass=41.5)
    eng.a3_configure(spec, g["clock_lut"])
    rng = np.random.default_rng(900 + seed)
    N = int(rng.integers(1, 600))
    seq_len = rng.choice([1, 2, 5, 20], N).astype(np.int32)
    t1 = (rng.integers(0, 20, N) % seq_len).astype(np.int32)
    t2 = np.minimum(t1 + 1, seq_len - 1).astype(np.int32)
    mode = np.where(seq_len == 1, _abi.MODE_STANDING,
                    rng.choice([_abi.MODE_FORWARD, _abi.MODE_BACKWARD, _abi.MODE_LATERAL], N)).astype(np.int32)
    seq = np.zeros((N, 20, 4))
    seq[:, :, 0] = 0.3 * np.arange(20) + rng.normal(0, 0.02, (N, 20))
    seq[:, :, 1] = 0.15 * (1 - 2 * (np.arange(20) % 2))
    seq[:, :, 3] = rng.uniform(-0.3, 0.3, (N, 1))
    st_h = dict(phase=rng.integers(0, 88, N).astype(np.int32), t1=t1, t2=t2,
                reached_frames=rng.integers(0, 36, N).astype(np.int32), target_reached=rng.integers(0, 2, N).astype(np.uint8),
                mode=mode, seq_len=seq_len, sequence=seq, goal=np.zeros((N, 8)))
    tgt = seq[np.arange(N), t1, :3]
    near = rng.uniform(size=(N, 1)) < 0.6
    inp = dict(qpos=np.concatenate([rng.normal(0, 1, (N, 3)), rng.normal(0, 1, (N, 4)), rng.uniform(-1, 1, (N, 18))], 1),
               qvel=rng.normal(0, 1, (N, 24)), act_len=rng.uniform(-1, 1, (N, 12)), act_vel=rng.normal(0, 2, (N, 12)),
               lf_pos=tgt + np.where(near, rng.normal(0, 0.08, (N, 3)), rng.normal(0, 0.6, (N, 3))),
               rf_pos=tgt + rng.normal(0, 0.4, (N, 3)),
               lf_vel=rng.normal(0, 0.2, (N, 3)), rf_vel=rng.normal(0, 0.2, (N, 3)),
               root_pos=tgt + np.array([0, 0, 0.62]) + rng.normal(0, 0.05, (N, 3)),
               root_quat=rng.normal(0, 1, (N, 4)), head_pos=tgt + np.array([0, 0, 1.2]) + rng.normal(0, 0.1, (N, 3)),
               grf_l=rng.uniform(0, 400, N), grf_r=rng.uniform(0, 400, N), min_z=rng.uniform(-0.01, 0.03, N),
               n_r=rng.integers(0, 3, N).astype(np.int32), n_l=rng.integers(0, 3, N).astype(np.int32),
               bad=(rng.uniform(size=N) < 0.1).astype(np.uint8))
    st_d = {k: dev(v) for k, v in st_h.items()}
    st_o = {k: v.copy() for k, v in st_h.items()}
    for _ in range(5):
        o = eng.a3_step({k: dev(v) for k, v in inp.items()}, st_d, obs_f64=True)
        eo = oracle.a3_step(spec, g["clock_lut"], inp, st_o)
        for k in ("phase", "t1", "t2", "reached_frames", "target_reached"):
            assert np.array_equal(host(st_d[k]), st_o[k]), (k, seed)
        assert np.array_equal(host(o["done"]), eo["done"])
        np.testing.assert_allclose(host(st_d["goal"]), st_o["goal"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(host(o["obs"]), eo["obs"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(host(o["rew6"]), eo["rew6"], rtol=2e-6, atol=1e-7)


def test_a3_pd(eng, golden, oracle):
    g = golden("a3_task.npz")
    spec = specs.A3Spec()
    eng.a3_configure(spec, g["clock_lut"])
    a32 = g["pd_action"].astype(np.float32)
    tgt = eng.a3_pd_target(dev(a32))
    assert np.array_equal(host(tgt), oracle.a3_pd_target(spec, a32))
    for s in range(g["pd_q"].shape[1]):
        tau = eng.a3_pd_torque(dev(spec.kp), dev(spec.kd), dev(g["pd_target"]), dev(g["pd_q"][:, s]),
                               dev(g["pd_qd"][:, s]))
        assert np.array_equal(host(tau), g["pd_tau"][:, s])           # bit-exact vs the reference


# --------------------------------------------------------------------------------- edges
def test_foot_force_columns_generic_kernel(eng, oracle):
    """use_foot_forces=True appends mean_grf/1000 (loco_env_base.py:749-758): n_grf = 6 columns
    run through the table-driven kernel."""
    spec = specs.unitree_h1("walk")
    spec.n_grf = 6
    assert spec.n_obs == 38
    rng = np.random.default_rng(21)
    T, N = 3, 200
    qpos, qvel, act = h1_synthetic_block(spec, T, N, seed=5, fall_frac="wide")
    grf = rng.normal(0, 400, (T, N, 6))
    prev = rng.normal(1.25, 0.3, N)
    eng.il_configure(spec)
    o = eng.il_step(dev(qpos), dev(qvel), dev(act), dev(prev), grf_mean=dev(grf), obs_f64=True)
    torch.cuda.synchronize()
    ref = oracle.il_step(spec, qpos, qvel, act, prev, grf_mean=grf, obs_f64=True)
    assert np.array_equal(host(o["obs"]), ref["obs"])
    assert np.array_equal(host(o["obs"])[..., 32:], grf / 1000.0)
    assert np.array_equal(host(o["absorbing"]), ref["absorbing"])
    assert np.array_equal(host(o["ctrl"]), ref["ctrl"])
    from olympic_hip._ffi import OlyError
    with pytest.raises(OlyError):
        eng.il_step(dev(qpos), dev(qvel), dev(act), dev(prev))           # grf_mean missing


def test_empty_inputs(eng):
    spec = specs.unitree_h1("walk")
    eng.il_configure(spec)
    z = lambda *s, dt=torch.float64: torch.zeros(s, dtype=dt, device="cuda")
    o = eng.il_step(z(0, 5, 17), z(0, 5, 17), z(0, 5, 11, dt=torch.float32), z(5), prev_out=z(5))
    assert o["obs"].shape == (0, 5, 32)
    o = eng.il_step(z(1, 0, 17), z(1, 0, 17), None, z(0))
    assert o["reward"].numel() == 0
    ret, adv = eng.return_scan(_abi.SCAN_GAE, 0.99, 0.97, z(0, 4, dt=torch.float32), z(0, 4, dt=torch.float32),
                               z(0, 4, dt=torch.float32), z(0, 4, dt=torch.uint8))
    assert ret.numel() == 0
    st = eng.adv_stats(z(0, dt=torch.float32))
    assert host(st).tolist() == [0.0, 0.0, 0.0]
    assert eng.disc_reward(z(0, dt=torch.float32)).numel() == 0
    cs = eng.col_stats(z(0, 32, dt=torch.float32))
    assert not host(cs).any()


def test_error_codes_before_configure_and_bad_models():
    from olympic_hip._ffi import OlyError
    from olympic_hip.engine import Engine
    e = Engine(0)
    z = lambda *s, dt=torch.float64: torch.zeros(s, dtype=dt, device="cuda")
    with pytest.raises(OlyError, match="il_step before il_configure"):
        e.il_step(z(1, 4, 17), z(1, 4, 17), None, z(4))
    with pytest.raises(OlyError, match="before traj_upload"):
        e.traj_reset(z(4, dt=torch.int32), z(4, dt=torch.int32))
    with pytest.raises(OlyError, match="before contact_configure"):
        e.contact_reduce(z(4, dt=torch.int32), z(4, 16, dt=torch.int32), z(4, 16, dt=torch.int32), z(4, 16, 6), z(4, 16))
    with pytest.raises(OlyError, match="before a3_configure"):
        e.a3_step({}, {"phase": z(4, dt=torch.int32)})
    # the C layer itself reports OLY_ENOTCONF / OLY_ERANGE / OLY_EINVAL with a message
    import ctypes as C
    from olympic_hip import _ffi
    L = _ffi.lib()
    rc = L.oly_il_step(e.ctx.handle, 1, 4, None, None, None, None, None, None, None, None, None, None, None, 0, None)
    assert rc == _abi.OLY_ENOTCONF and b"before oly_il_configure" in L.oly_last_error(e.ctx.handle)
    bad = specs.unitree_h1("walk")
    bad.act_to_ctrl = bad.act_to_ctrl.copy()
    bad.act_to_ctrl[0] = 99
    rc = L.oly_il_configure(e.ctx.handle, C.byref(bad.to_c()))
    assert rc == _abi.OLY_ERANGE and b"act_to_ctrl" in L.oly_last_error(e.ctx.handle)
    bad2 = specs.unitree_h1("walk")
    bad2.qpos_adr = bad2.qpos_adr.copy()
    bad2.qpos_adr[5] = 17
    assert L.oly_il_configure(e.ctx.handle, C.byref(bad2.to_c())) == _abi.OLY_ERANGE
    e.il_configure(specs.unitree_h1("walk"))
    rc = L.oly_il_step(e.ctx.handle, 1, 4, None, None, None, None, None, None, None, None, None, None, None, 0, None)
    assert rc == _abi.OLY_EINVAL
    assert L.oly_return_scan(e.ctx.handle, 7, 1, 1, C.c_double(0.9), C.c_double(0.9), None, None, None, None, None,
                             None, None) == _abi.OLY_EINVAL


def test_unaligned_buffers_take_the_generic_path(eng, oracle):
    """Views that are not 16-B aligned must still be correct (slow path)."""
    spec = specs.unitree_h1("walk")
    eng.il_configure(spec)
    T, N = 2, 300
    qpos, qvel, act = h1_synthetic_block(spec, T, N, seed=8, fall_frac="wide")
    prev = np.zeros(N)
    big = torch.zeros(T * N * 17 + 1, dtype=torch.float64, device="cuda")
    qp = big[1:].view(T, N, 17)                       # 8-B aligned only
    qp.copy_(torch.as_tensor(qpos))
    assert qp.data_ptr() % 16 == 8 and qp.is_contiguous()
    o = eng.il_step(qp, dev(qvel), dev(act), dev(prev))
    ref = oracle.il_step(spec, qpos, qvel, act, prev)
    _cmp_il({k: (None if v is None else host(v)) for k, v in o.items()}, ref, False)


def _random_il_spec(rng, h1_shape=False):
    """A made-up table-driven robot: random gather permutation, thresholds, ranges."""
    from olympic_hip.specs import ILRobotSpec
    if h1_shape:
        nq = nv = n_pos = n_vel = 17
        n_act = nu = 11
    else:
        nq, nv = int(rng.integers(8, 40)), int(rng.integers(8, 40))
        n_pos, n_vel = int(rng.integers(3, nq + 1)), int(rng.integers(1, nv + 1))
        nu = int(rng.integers(1, 20))
        n_act = int(rng.integers(1, nu + 1))
    n_fall = int(rng.integers(0, 11)) if not h1_shape else int(rng.integers(0, 9))
    keys = [f"q_j{i}" for i in range(n_pos)] + [f"dq_j{i}" for i in range(n_vel)]
    n_obs = n_pos + n_vel - 2
    fall_keys = [keys[2 + int(i)] for i in rng.integers(0, n_obs, n_fall)]
    lo = rng.uniform(-1.0, -0.1, n_fall)
    hi = rng.uniform(0.1, 1.0, n_fall)
    sp = ILRobotSpec(
        name="rand", obs_keys=keys, joint_names=[f"j{i}" for i in range(nq)], nq=nq, nv=nv, n_pos=n_pos,
        n_vel=n_vel, qpos_adr=rng.permutation(nq)[:n_pos].astype(np.int32),
        qvel_adr=rng.permutation(nv)[:n_vel].astype(np.int32), joint_lo=-np.ones(n_pos), joint_hi=np.ones(n_pos),
        action_names=[f"a{i}" for i in range(n_act)], nu=nu,
        act_to_ctrl=rng.permutation(nu)[:n_act].astype(np.int32), ctrl_lo=rng.uniform(-2, -0.5, n_act),
        ctrl_hi=rng.uniform(0.5, 2, n_act), fall_tests=list(zip(fall_keys, lo.tolist(), hi.tolist())),
        fall_names=[f"c{i}" for i in range(n_fall)], target_velocity=float(rng.uniform(0.5, 3)))
    sp.__class__ = type("RandSpec", (ILRobotSpec,), {"reward_idx": property(lambda self: self._ridx)})
    sp._ridx = int(rng.integers(0, n_obs))
    if not h1_shape and rng.uniform() < 0.4:               # foot-force columns (mean_grf / 1000)
        sp.n_grf = int(rng.choice([3, 6, 12]))
    return sp


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("OLY_FUZZ", "12"))))
def test_random_table_driven_robots(eng, oracle, seed):
    """The IL kernel is table-driven (Atlas/Talos are data): random robots vs the oracle,
    alternating between H1-shaped tables (fast tile kernel) and arbitrary shapes (runtime-shape
    persistent kernel + ragged tail); every third case is large enough for several tiles per
    workgroup.  OLY_FUZZ=n runs n seeds."""
    rng = np.random.default_rng(100 + seed)
    sp = _random_il_spec(rng, h1_shape=(seed % 2 == 0))
    T = int(rng.integers(1, 5))
    N = int(rng.integers(1, 700)) if seed % 3 else int(rng.integers(5000, 40000))
    qpos = rng.uniform(-1.2, 1.2, (T, N, sp.nq))
    qvel = rng.normal(0, 1.5, (T, N, sp.nv))
    act = rng.uniform(-1.5, 1.5, (T, N, sp.n_act)).astype(np.float32)
    prev = rng.normal(1, 1, N)
    kw = dict(grf_mean=rng.normal(0, 400, (T, N, sp.n_grf))) if sp.n_grf else {}
    for f64 in (False, True):
        o = _run_il(eng, sp, qpos, qvel, act, prev, obs_f64=f64, ctrl_f64=f64,
                    **({"grf_mean": dev(kw["grf_mean"])} if kw else {}))
        ref = oracle.il_step(sp, qpos, qvel, act, prev, obs_f64=f64, ctrl_f64=f64, **kw)
        _cmp_il(o, ref, f64)
    assert (np.asarray(ref["fall_code"]) > 0).any() or len(sp.fall_tests) == 0


@pytest.mark.parametrize("robot", ["atlas", "talos"])
def test_atlas_talos_on_gpu(eng, golden, oracle, robot):
    g = golden(f"{robot}_tables.npz")
    for tag, kw in (("default", {}), ("all_joints", dict(disable_arms=False, disable_back_joint=False))):
        sp = getattr(specs, robot)("walk", **kw)
        obs = g[f"{tag}.obs"]
        full = np.concatenate([np.zeros((len(obs), 2)), obs], axis=1)
        qpos, qvel = h1_rows_from_full(sp, full)
        rng = np.random.default_rng(3)
        act = rng.uniform(-1.2, 1.2, (1, len(obs), sp.n_act)).astype(np.float32)
        prev = rng.normal(1.25, 0.4, len(obs))
        o = _run_il(eng, sp, qpos[None], qvel[None], act, prev, obs_f64=True)
        ref = oracle.il_step(sp, qpos[None], qvel[None], act, prev, obs_f64=True)
        _cmp_il(o, ref, True)
        assert np.array_equal(o["fall_code"][0], g[f"{tag}.code"])          # vs the reference class
        assert np.array_equal(o["absorbing"][0].astype(bool), g[f"{tag}.fallen"])


def test_make_atlas_env_steps(eng):
    from olympic_hip.envs import LocoEnvBase
    env = LocoEnvBase.make("Atlas.walk.real", seed=1)
    obs = env.reset()
    assert obs.shape == (env.spec.n_obs,) == (30,)
    o2, r, ab, _ = env.step(np.zeros(env.spec.n_act))
    assert o2.shape == (30,) and 0.0 < r <= 1.0 and ab is False


# ------------------------------------------------------------------------------ K9 / normalisers
def _ppo_case(rng, B, A, mode):
    mu = rng.normal(0, 0.3, (B, A)).astype(np.float32)
    old_mu = (mu + rng.normal(0, 0.05, (B, A))).astype(np.float32)
    std = {0: np.array([0.22], np.float32), 1: rng.uniform(0.1, 0.4, A).astype(np.float32),
           2: rng.uniform(0.1, 0.4, (B, A)).astype(np.float32)}[mode]
    old_std = std if mode != 2 else (std * rng.uniform(0.9, 1.1, (B, A))).astype(np.float32)
    action = (old_mu + rng.normal(0, 0.25, (B, A))).astype(np.float32)
    adv = rng.normal(0, 1, B).astype(np.float32)
    adv[::17] = 0.0                                          # tie of the two surrogate branches
    ret, value = rng.normal(0, 1, B).astype(np.float32), rng.normal(0, 1, B).astype(np.float32)
    return mu, std, old_mu, old_std, action, adv, ret, value


@pytest.mark.parametrize("B,A,mode", [(64, 12, 0), (1, 3, 1), (5000, 12, 2), (777, 64, 1), (300000, 11, 0),
                                      (4097, 16, 1), (100, 4, 0), (1000, 8, 2), (300001, 12, 0)])
def test_ppo_loss_vs_oracle(eng, oracle, B, A, mode):
    """Tolerance: device expf/logf vs libm are within 1-2 ulp per element; the sums are fp64 on
    both sides -> 2e-5 relative on the loss terms, 1e-4 relative (1e-9 absolute) on gradients."""
    rng = np.random.default_rng(B + A + mode)
    c = _ppo_case(rng, B, A, mode)
    r = eng.ppo_loss(*(dev(x) for x in c), 0.2, 0.5, want_grad=True, want_grad_std=True)
    scal, gmu, gsd, gv = oracle.ppo_loss(*c, 0.2, 0.5)
    np.testing.assert_allclose(host(r["scal"]), scal, rtol=2e-5, atol=1e-7)
    assert scal[4] > 0 or B == 1                               # the case exercises clipping
    np.testing.assert_allclose(host(r["grad_value"]), gv, rtol=1e-6, atol=0)
    np.testing.assert_allclose(host(r["grad_mu"]), gmu, rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(host(r["grad_std"]), gsd, rtol=1e-4, atol=1e-9)


def test_ppo_loss_golden_and_autograd(eng, golden):
    """(a) the reference's update_policy numbers; (b) the fused Function's parameter gradients
    equal torch autograd of the op-by-op form (ppo.update_policy) within fp32 rounding."""
    from helpers import ppo_update_arrays
    from olympic_hip.ppo import PPO, MLPCritic, MLPGaussianActor
    g = golden("ppo_update.npz")
    a = ppo_update_arrays(g)
    r = eng.ppo_loss(dev(a["mu"]), dev(np.array([a["std"]])), dev(a["old_mu"]), dev(np.array([a["std"]])),
                     dev(a["action"]), dev(a["adv"]), dev(a["ret"]), dev(a["value"]), a["clip"], 0.5)
    s = host(r["scal"])
    for i, n in enumerate(("actor_loss", "entropy_penalty", "critic_loss", "approx_kl_div", "clip_fraction")):
        np.testing.assert_allclose(s[i], float(g[n]), rtol=3e-5, atol=3e-7, err_msg=n)
    loss, gd, gm = eng.mirror_loss(dev(a["mu"]), dev(a["mir"]), dev(a["act_src"]), dev(a["act_sign"]))
    np.testing.assert_allclose(float(loss), float(g["mirror_loss"]), rtol=3e-5)

    def load(mod, tag):
        mod.load_state_dict({k[len(tag) + 1:]: torch.tensor(g[k]) for k in g.files if k.startswith(tag + ".")})
        return mod.cuda()
    std = torch.tensor(float(g["fixed_std"]))
    ppo = PPO.__new__(PPO)
    ppo.clip, ppo.vf_coeff = a["clip"], 0.5
    t = lambda k: dev(g[k], torch.float32)
    from olympic_hip.wrappers import SymmetricEnv

    class Dummy:
        base_obs_len = 41
    sym = SymmetricEnv(Dummy, mirrored_obs=g["mirrored_obs"].tolist(), mirrored_act=g["mirrored_acts"].tolist(),
                       clock_inds=[31, 32])
    grads = []
    for fused in (False, True):
        ppo.policy = load(MLPGaussianActor(41, 12, fixed_std=std), "pi")
        ppo.old_policy = load(MLPGaussianActor(41, 12, fixed_std=std), "old")
        ppo.critic = load(MLPCritic(41), "vf")
        if fused:
            out = ppo.update_policy_fused(eng, t("obs"), t("act"), t("ret"), t("adv"), sym.mirror_clock_observation,
                                          dev(a["act_src"]), dev(a["act_sign"]))
        else:
            out = ppo.update_policy(t("obs"), t("act"), t("ret"), t("adv"), 1, sym.mirror_clock_observation,
                                    sym.mirror_action)
        (out[0] + 0.4 * out[4] + 0.01 * out[1] + out[2]).sum().backward()   # actor and critic share no parameters
        grads.append([p.grad.clone() for p in list(ppo.policy.parameters()) + list(ppo.critic.parameters())])
        for i, n in enumerate(("actor_loss", "entropy_penalty", "critic_loss", "approx_kl_div", "mirror_loss")):
            np.testing.assert_allclose(float(out[i]), float(g[n]), rtol=3e-5, atol=3e-7, err_msg=n)
    for gu, gf in zip(*grads):
        scale = float(gu.abs().max()) + 1e-12
        assert float((gu - gf).abs().max()) <= 2e-5 * scale


def test_ppo_loss_learned_std_autograd(eng):
    """learn_std actors: d/d std through the fused Function (per-dim parameter and full [B,A])."""
    from olympic_hip.ppo import FusedPPOLoss
    rng = np.random.default_rng(3)
    B, A = 257, 6
    mu, _, old_mu, _, action, adv, ret, value = (dev(x) for x in _ppo_case(rng, B, A, 0))
    for shape in ((A,), (B, A)):
        log_std = torch.full(shape, -1.2, device="cuda").requires_grad_()
        mu_p = mu.clone().requires_grad_()
        res = []
        for fused in (False, True):
            log_std.grad = mu_p.grad = None
            std = log_std.exp()
            if fused:
                a_l, ent, c_l, _, _ = FusedPPOLoss.apply(eng, mu_p, std, old_mu, std.detach() * 1.05, action, adv, ret,
                                                         value, 0.2, 0.5)
            else:
                pdf = torch.distributions.Normal(mu_p, std.expand(B, A))
                old = torch.distributions.Normal(old_mu, (std.detach() * 1.05).expand(B, A))
                lp, olp = pdf.log_prob(action).sum(-1), old.log_prob(action).sum(-1)
                ratio = (lp - olp).exp()
                a_l = -torch.min(ratio * adv, ratio.clamp(0.8, 1.2) * adv).mean()
                ent = -pdf.entropy().mean()
            (a_l + 0.3 * ent).backward()
            res.append((float(a_l), float(ent), log_std.grad.clone(), mu_p.grad.clone()))
        (a0, e0, gs0, gm0), (a1, e1, gs1, gm1) = res
        assert a0 == pytest.approx(a1, rel=2e-5) and e0 == pytest.approx(e1, rel=2e-6)
        assert float((gs0 - gs1).abs().max()) <= 3e-5 * float(gs0.abs().max())
        assert float((gm0 - gm1).abs().max()) <= 3e-5 * float(gm0.abs().max())


def test_signed_perm_and_mirror_loss(eng, oracle, golden):
    from olympic_hip.wrappers import _signed_perm
    g = golden("symmetry.npz")
    for tab, x, want in ((g["mirrored_obs"], g["obs"], g["obs_mirror"]), (g["mirrored_acts"], g["act"], g["act_mirror"])):
        src, sgn = _signed_perm(tab.tolist())
        out = eng.signed_perm(dev(x, torch.float32), dev(src, torch.int32), dev(sgn))
        assert np.array_equal(host(out), want.astype(np.float32))          # +-1 times one input: exact
    rng = np.random.default_rng(5)
    B, A = 4096, 12
    src, sgn = _signed_perm(g["mirrored_acts"].tolist())
    det, mir = rng.normal(0, 1, (B, A)).astype(np.float32), rng.normal(0, 1, (B, A)).astype(np.float32)
    loss, gd, gm = eng.mirror_loss(dev(det), dev(mir), dev(src, torch.int32), dev(sgn))
    el, egd, egm = oracle.mirror_loss(det, mir, src, sgn)
    assert float(loss) == pytest.approx(el, rel=1e-12)                      # same fp32 squares, fp64 sum
    assert np.array_equal(host(gd), egd) and np.array_equal(host(gm), egm)  # mul/sub only: exact


def test_obs_filter_and_device_running_mean_std(eng, oracle, golden):
    from olympic_hip.normalize import RunningMeanStd
    g = golden("normalize.npz")
    clip, eps = float(g["clipob"]), float(g["epsilon"])
    out = eng.obs_filter(dev(g["frozen_in"]), dev(g["mean"][-1]), dev(g["var"][-1]), eps, clip)
    assert np.array_equal(host(out), g["frozen_out"].astype(np.float32))    # fp64 sub/div/sqrt: exact
    rms = RunningMeanStd(eng, shape=(6,))
    off = np.concatenate([[0], np.cumsum(g["lens"])])
    for i in range(len(g["lens"])):
        xb = dev(g["x"][off[i]:off[i + 1]])
        rms.update(xb)
        np.testing.assert_allclose(host(rms.mean), g["mean"][i], rtol=2e-6, atol=2e-7)   # reference moments fp32
        np.testing.assert_allclose(host(rms.var), g["var"][i], rtol=2e-5)
        assert rms.count == pytest.approx(float(g["count"][i]))
        o = eng.obs_filter(xb, rms.mean, rms.var, eps, clip)
        assert np.array_equal(host(o), oracle.obs_filter(g["x"][off[i]:off[i + 1]], host(rms.mean), host(rms.var), eps, clip))
        np.testing.assert_allclose(host(o), g["out"][off[i]:off[i + 1]], rtol=2e-5, atol=2e-6)
    big = np.random.default_rng(0).normal(0, 3, (100000, 32)).astype(np.float32)
    m, v = big.mean(0, dtype=np.float64), big.var(0, dtype=np.float64)
    o = eng.obs_filter(dev(big), dev(m), dev(v), 1e-8, 10.0)
    assert np.array_equal(host(o), oracle.obs_filter(big, m, v, 1e-8, 10.0))


# ------------------------------------------------------------------------------ K3: IL ground forces
@pytest.mark.parametrize("W,N,C", [(1, 1, 16), (10, 4097, 16), (10, 33, 40), (3, 1000, 5)])
def test_il_ground_forces_vs_oracle(eng, oracle, W, N, C):
    """First matching contact per sensor pair (either geom order), force[:3], in-order window mean:
    selections are index work (bit-exact), the mean is W in-order adds and one divide (bit-exact)."""
    from olympic_hip._ffi import OlyError
    rng = np.random.default_rng(W + N + C)
    ngeom = 30
    gg = np.full(ngeom, -1, np.int32)
    gg[0], gg[[7, 8]], gg[[20]], gg[[21]], gg[[22]] = 0, 1, 2, 3, 4
    pairs = [(0, 1), (0, 2), (0, 3), (0, 4)][:1 + (N % 4)]
    ncon = rng.integers(0, C + 3, (W, N)).astype(np.int32)            # RAW counts, also > C (only C slots are staged)
    if N > 2:
        ncon[0, 1] = -1                                               # a negative count: overflow, reduced as 0 contacts
    g1 = np.where(rng.uniform(size=(W, N, C)) < 0.5, 0, rng.integers(-1, ngeom + 1, (W, N, C))).astype(np.int32)
    g2 = rng.choice([0, 7, 8, 20, 21, 22, 3, 29, -1, ngeom], size=(W, N, C)).astype(np.int32)
    swap = rng.uniform(size=(W, N, C)) < 0.3
    g1, g2 = np.where(swap, g2, g1).astype(np.int32), np.where(swap, g1, g2).astype(np.int32)
    f6 = rng.normal(0, 200, (W, N, C, 6))
    eng.grf_configure(gg, pairs)
    o = eng.il_ground_forces(dev(ncon), dev(g1), dev(g2), dev(f6), want_steps=True, check=False)
    e_step, e_mean, e_over = oracle.il_ground_forces(gg, pairs, ncon, g1, g2, f6, want_overflow=True)
    assert np.array_equal(host(o["steps"]), e_step)
    assert np.array_equal(host(o["mean"]), e_mean)
    assert (e_step != 0).any() or N == 1
    # ncon > C: exact while every sensor pair has its first contact among the C staged slots (the reference scans
    # all data.ncon contacts, UnitreeH1.py:113-123), flagged otherwise; bit-exact flags; check=True reads them back and raises
    assert np.array_equal(host(o["overflow"]), e_over)
    if N > 2:
        assert e_over[1] == 1 and 0 < e_over.sum() < N, "the case must hold flagged and unflagged ncon > C environments"
        assert ((ncon > C).any(0) & (e_over == 0)).any()
        with pytest.raises(OlyError, match="staged slots"):
            eng.il_ground_forces(dev(ncon), dev(g1), dev(g2), dev(f6), check=True)
        keep = e_over == 0                                            # without the flagged environments: no error
        o2 = eng.il_ground_forces(dev(ncon[:, keep]), dev(g1[:, keep]), dev(g2[:, keep]), dev(f6[:, keep]), check=True)
        assert np.array_equal(host(o2["mean"]), e_mean[keep])
    # the dense form (rows already reduced per substep, as the packed host batcher stages them)
    assert np.array_equal(host(eng.il_grf_window(dev(e_step))), e_mean)


def test_h1_env_with_foot_forces(eng, oracle):
    """UnitreeH1(use_foot_forces=True): obs = [joint obs, mean_grf / 1000] with the window mean of
    the control step's substep contacts; zeros right after reset."""
    from olympic_hip._ffi import OlyError
    from olympic_hip.envs import ReplayPhysics, VecLocoEnv
    sp = specs.unitree_h1("walk").with_foot_forces("UnitreeH1")
    assert sp.n_grf == 6 and sp.n_obs == 38 and sp.geom_group[0] == 0 and sp.geom_group[22] == 1 and sp.geom_group[12] == 2
    T, N, W, C = 5, 256, 10, 16
    qpos, qvel, act = h1_synthetic_block(sp, T, N, seed=3, fall_frac="wide")
    rng = np.random.default_rng(4)
    con = dict(ncon=rng.integers(0, 6, (T, W, N)).astype(np.int32),
               geom1=np.zeros((T, W, N, C), np.int32),
               geom2=rng.choice([12, 22, 5, 30], size=(T, W, N, C)).astype(np.int32),
               force6=rng.normal(0, 300, (T, W, N, C, 6)))
    phys = ReplayPhysics(sp, dev(qpos), dev(qvel), contacts={k: dev(v) for k, v in con.items()})
    env = VecLocoEnv(sp, N, engine=eng, physics=phys, random_start=False)
    first = env.reset()
    assert first.shape == (N, 38) and not host(first)[:, 32:].any()           # mean_grf.reset()
    means = np.stack([oracle.il_ground_forces(sp.geom_group, sp.grf_pairs, con["ncon"][t], con["geom1"][t],
                                              con["geom2"][t], con["force6"][t])[1] for t in range(T)])
    ref = oracle.il_step(sp, qpos, qvel, act, host(env._prev), grf_mean=means)
    for t in range(T):
        o, r, a, _ = env.step(dev(act[t]))
        assert np.array_equal(host(o), ref["obs"][t])
        assert np.array_equal(host(o)[:, 32:], (means[t] / 1000.0).astype(np.float32))
        assert np.array_equal(host(a), ref["absorbing"][t].astype(bool))
    env.raise_if_contact_overflow()                                            # nothing was flagged
    # more contacts than staged slots, none of the staged ones between a sensor pair: step() does not block on it,
    # the sticky flag is read at the next host-synchronous point
    con2 = {k: v.copy() for k, v in con.items()}
    con2["ncon"][0, 3, 7] = C + 4
    con2["geom2"][0, 3, 7, :] = 5
    phys2 = ReplayPhysics(sp, dev(qpos), dev(qvel), contacts={k: dev(v) for k, v in con2.items()})
    env2 = VecLocoEnv(sp, N, engine=eng, physics=phys2, random_start=False)
    env2.reset()
    env2.step(dev(act[0]))
    env2.step(dev(act[1]))
    with pytest.raises(OlyError, match=r"environments \[7\]"):
        env2.reset()
    env2.reset()                                                               # the flags were consumed
    with pytest.raises(NotImplementedError):
        specs.atlas("walk").with_foot_forces("Atlas")


def test_error_paths_of_the_newer_entry_points():
    """Misuse is reported before any launch: OLY_ENOTCONF / OLY_EINVAL with a message (C layer),
    OlyError for shape / dtype / device mistakes (host layer)."""
    from olympic_hip._ffi import OlyError
    from olympic_hip.engine import Engine
    e = Engine(0)
    z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device="cuda")
    with pytest.raises(OlyError, match="before grf_configure"):
        e.il_ground_forces(z(2, 4, dt=torch.int32), z(2, 4, 16, dt=torch.int32), z(2, 4, 16, dt=torch.int32),
                           z(2, 4, 16, 6, dt=torch.float64))
    with pytest.raises(OlyError, match="oly_grf_configure: bad argument"):
        e.grf_configure(np.zeros(4, np.int32), [(0, 1)] * 5)                       # > OLY_MAX_GRF_PAIRS
    e.grf_configure(np.array([0, 1, 2], np.int32), [(0, 1)])
    with pytest.raises(OlyError, match="force6"):
        e.il_ground_forces(z(2, 4, dt=torch.int32), z(2, 4, 16, dt=torch.int32), z(2, 4, 16, dt=torch.int32),
                           z(2, 4, 16, 6))                                          # f32 instead of f64
    with pytest.raises(OlyError, match="oly_ppo_loss: bad argument"):
        e.ppo_loss(z(8, 65), z(1), z(8, 65), z(1), z(8, 65), z(8), z(8), z(8), 0.2)  # A > OLY_MAX_ACT
    with pytest.raises(OlyError, match="old_mu"):
        e.ppo_loss(z(8, 12), z(1), z(7, 12), z(1), z(8, 12), z(8), z(8), z(8), 0.2)
    x = z(4, 6)
    with pytest.raises(OlyError, match="oly_signed_perm: bad argument"):
        e.signed_perm(x, z(6, dt=torch.int32), z(6), out=x)                         # in place is not allowed
    with pytest.raises(OlyError, match="mean"):
        e.obs_filter(x, z(6), z(6, dt=torch.float64))                               # f32 statistics
    import ctypes as C
    from olympic_hip import _ffi
    L = _ffi.lib()
    assert L.oly_mirror_loss(e.ctx.handle, 0, 12, None, None, None, None, None, None, None, None) == _abi.OLY_EINVAL
    assert b"oly_mirror_loss" in L.oly_last_error(e.ctx.handle)
    assert L.oly_il_ground_forces(e.ctx.handle, 0, 4, 16, None, None, None, None, None, None, None, None) == _abi.OLY_EINVAL


@pytest.mark.parametrize("N", [1, 63, 4096, 100001])
def test_rollout_cuts_vs_oracle(eng, oracle, N):
    rng = np.random.default_rng(N)
    tl = rng.integers(0, 12, N).astype(np.int32)
    for last in (0, 1):
        done = (rng.uniform(size=N) < 0.1).astype(np.uint8)
        tl_d, fl_d, nc_d = dev(tl), torch.zeros(N, dtype=torch.uint8, device="cuda"), torch.full((1,), 77, dtype=torch.int32, device="cuda")
        eng.rollout_cuts(dev(done), tl_d, fl_d, nc_d, 10, last)
        fl, tl2, nc = oracle.rollout_cuts(done, tl, 10, last)
        assert np.array_equal(host(fl_d), fl) and np.array_equal(host(tl_d), tl2) and int(nc_d) == nc
        tl = tl2


# ------------------------------------------------------------------------------ memory safety
class _Guard:
    """Every tensor the engine allocates (and every in/out array handed to it) is carved out of a
    larger byte buffer whose 4 KiB margins are filled with 0xA5; check() fails if any kernel
    wrote outside its output.  (GPU AddressSanitizer is not available on this pool.)"""
    PAD = 4096

    def __init__(self, eng):
        self.eng, self.items, self.orig = eng, [], eng._new
        eng._new = self.new

    def new(self, shape, dtype, fill=None):
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        raw = torch.full((n + 2 * self.PAD,), 0xA5, dtype=torch.uint8, device="cuda")
        view = raw[self.PAD:self.PAD + n].view(dtype).view(shape)
        if fill is not None:
            view.copy_(torch.as_tensor(fill).to(dtype).reshape(shape))
        self.items.append((raw, n))
        return view

    def check(self, what):
        torch.cuda.synchronize()
        for raw, n in self.items:
            assert bool((raw[:self.PAD] == 0xA5).all()) and bool((raw[self.PAD + n:] == 0xA5).all()), what
        self.items.clear()

    def close(self):
        self.eng._new = self.orig


@pytest.mark.parametrize("N", [1, 63, 65, 129, 1000, 4097])
def test_no_kernel_writes_outside_its_outputs(golden, N):
    from olympic_hip.engine import Engine
    eng = Engine(0)
    gd = _Guard(eng)
    try:
        rng = np.random.default_rng(N)
        # K1 / K5: every static and runtime-shape path, T = 1 and 3, f32 and f64 outputs
        for sp in (specs.unitree_h1("walk"), specs.atlas("walk"), specs.unitree_h1("walk", disable_arms=False),
                   specs.unitree_h1("walk").with_foot_forces("UnitreeH1")):
            eng.il_configure(sp)
            for T in (1, 3):
                q, v, a = h1_synthetic_block(sp, T, N, seed=1, fall_frac="wide")
                grf = dev(rng.normal(0, 100, (T, N, sp.n_grf))) if sp.n_grf else None
                for f64 in (False, True):
                    eng.il_step(dev(q), dev(v), dev(a), gd.new((N,), torch.float64, np.ones(N)), grf_mean=grf,
                                obs_f64=f64, ctrl_f64=f64)
                    gd.check(("il_step", sp.name, sp.n_obs, T, f64))
        # K6 / rollout cuts / K7
        for T in (1, 33, 100):
            for Nn in (N, 4 * ((N + 3) // 4)):
                r, v, vn = (dev(rng.normal(0, 1, (T, Nn)).astype(np.float32)) for _ in range(3))
                fl = dev((rng.uniform(size=(T, Nn)) < 0.05).astype(np.uint8) * 3)
                for mode in (_abi.SCAN_RETURN, _abi.SCAN_GAE):
                    ret, adv = eng.return_scan(mode, 0.99, 0.97, r, v, vn, fl)
                    gd.check(("return_scan", T, Nn, mode))
        st = eng.adv_stats(adv)
        eng.adv_normalize(gd.new(adv.shape, torch.float32, adv.cpu()), st, 1, 1e-5)
        eng.col_stats(dev(rng.normal(0, 1, (N, 32)).astype(np.float32)))
        eng.col_stats(dev(rng.normal(0, 1, (N, 41)).astype(np.float32)))
        eng.rollout_cuts(dev((rng.uniform(size=N) < 0.2).astype(np.uint8)), gd.new((N,), torch.int32, np.zeros(N)),
                         gd.new((N,), torch.uint8, np.zeros(N)), gd.new((1,), torch.int32, [0]), 10, False)
        gd.check("K7 / cuts")
        # K3 (RL + IL) and K2
        gb = np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32)
        eng.contact_configure(gb, 0, 7, 10)
        for C in (5, 16, 40):
            ncon = dev(rng.integers(0, C + 1, N).astype(np.int32))
            g1, g2 = dev(np.zeros((N, C), np.int32)), dev(rng.integers(0, 13, (N, C)).astype(np.int32))
            eng.contact_reduce(ncon, g1, g2, dev(rng.normal(0, 1, (N, C, 6))), dev(rng.normal(0, 1, (N, C))))
            gd.check(("contact_reduce", C))
            eng.grf_configure(np.array([0, -1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 2, 2], np.int32), [(0, 1), (0, 2)])
            W = 3
            eng.il_ground_forces(dev(rng.integers(0, C + 1, (W, N)).astype(np.int32)), dev(np.zeros((W, N, C), np.int32)),
                                 dev(rng.integers(0, 13, (W, N, C)).astype(np.int32)), dev(rng.normal(0, 1, (W, N, C, 6))),
                                 want_steps=True)
            gd.check(("il_ground_forces", C))
        g = golden("a3_task.npz")
        sp3 = specs.A3Spec(mass=41.5)
        eng.a3_configure(sp3, g["clock_lut"])
        st_h, inp = a3_fixture_arrays(g, 0)
        pick = rng.integers(0, g["phase"].shape[0], N)
        st_d = {k: gd.new(v[pick].shape, torch.as_tensor(v).dtype, v[pick]) for k, v in st_h.items()}
        inp_d = {k: dev(np.ascontiguousarray(v[pick])) for k, v in inp.items()}
        inp_d.update(grf_l=dev(rng.uniform(0, 300, N)), grf_r=dev(rng.uniform(0, 300, N)), min_z=dev(rng.uniform(0, 0.02, N)),
                     n_r=dev(np.ones(N, np.int32)), n_l=dev(np.ones(N, np.int32)), bad=dev(np.zeros(N, np.uint8)))
        for f64 in (False, True):
            eng.a3_step(inp_d, st_d, obs_f64=f64)
        eng.a3_pd_target(dev(rng.uniform(-1, 1, (N, 12)).astype(np.float32)))
        gd.check("a3")
        # K8 / K9
        x = dev(rng.normal(0, 1, (N, 32)).astype(np.float32))
        m64, s64 = dev(np.zeros(32)), dev(np.ones(32))
        eng.disc_standardize(x, None, m64, s64)
        eng.disc_standardize(x, dev(np.arange(0, 32, 3, dtype=np.int32)), dev(np.zeros(11)), dev(np.ones(11)))
        eng.obs_filter(x, m64, s64)
        eng.disc_reward(dev(rng.normal(0, 3, N).astype(np.float32)))
        z = dev(rng.normal(0, 1, (N, 7)).astype(np.float32))
        eng.disc_reparam(z, z, z)
        for A in (3, 12, 64):
            c = _ppo_case(rng, N, A, 2)
            eng.ppo_loss(*(dev(t) for t in c), 0.2, 0.5, want_grad=True, want_grad_std=True)
            src = dev(rng.permutation(A).astype(np.int32))
            sg = dev(rng.choice([-1.0, 1.0], A).astype(np.float32))
            eng.mirror_loss(dev(c[0]), dev(c[2]), src, sg)
            eng.signed_perm(dev(c[0]), src, sg)
            gd.check(("K9", A))
        # K4
        tg = golden("trajectory.npz")
        eng.traj_upload(tg["table"])
        n_traj, L = tg["table"].shape[1], tg["table"].shape[2]
        ct, cs, org, smp = eng.traj_reset(dev(rng.integers(0, n_traj, N).astype(np.int32)), dev(rng.integers(0, L, N).astype(np.int32)))
        for _ in range(3):
            eng.traj_next(ct, cs, org, smp)
        eng.traj_euler(17, 0.01, dev(rng.normal(0, 1, (N, 17))), smp)
        gd.check("traj")
    finally:
        gd.close()


@pytest.mark.parametrize("robot", ["UnitreeH1+ff", "Talos+ff", "UnitreeH1+arms"])
def test_other_static_fast_paths(eng, oracle, robot):
    """use_foot_forces=True H1 / Talos and H1 with arms take compile-time-shape instantiations of
    the tile kernel: obs (incl. the mean_grf / 1000 columns), flags, ctrl bit-exact over several
    tiles per workgroup."""
    if robot == "UnitreeH1+arms":
        sp = specs.unitree_h1("walk", disable_arms=False)
        assert (sp.nq, sp.n_act, sp.n_obs) == (25, 19, 48)
    else:
        name = robot.split("+")[0]
        sp = (specs.unitree_h1("walk") if name == "UnitreeH1" else specs.talos("walk")).with_foot_forces(name)
    T, N = 5, 3000
    rng = np.random.default_rng(8)
    q, v, a = h1_synthetic_block(sp, T, N, seed=21, fall_frac="wide")
    grf = rng.normal(0, 400, (T, N, sp.n_grf)) if sp.n_grf else None
    prev = rng.normal(1.25, 0.3, N)
    for f64 in (False, True):
        o = _run_il(eng, sp, q, v, a, prev, obs_f64=f64, ctrl_f64=f64, **({"grf_mean": dev(grf)} if sp.n_grf else {}))
        ref = oracle.il_step(sp, q, v, a, prev, grf_mean=grf, obs_f64=f64, ctrl_f64=f64)
        _cmp_il(o, ref, f64)
    if sp.n_grf:
        assert np.array_equal(o["obs"][..., -sp.n_grf:], grf / 1000.0)


@pytest.mark.parametrize("robot", ["unitree_h1", "atlas", "talos"])
@pytest.mark.parametrize("arms,back", [(True, True), (True, False), (False, True), (False, False)])
def test_every_robot_configuration(eng, oracle, robot, arms, back):
    """All (disable_arms, disable_back_joint) combinations of the three robots have their own
    compile-time-shape instantiation of the tile kernel: each against the oracle."""
    sp = getattr(specs, robot)("walk", disable_arms=arms, disable_back_joint=back)
    T, N = 3, 1500
    q, v, a = h1_synthetic_block(sp, T, N, seed=31, fall_frac="wide")
    prev = np.random.default_rng(1).normal(1.25, 0.3, N)
    for f64 in (False, True):
        o = _run_il(eng, sp, q, v, a, prev, obs_f64=f64, ctrl_f64=f64)
        _cmp_il(o, oracle.il_step(sp, q, v, a, prev, obs_f64=f64, ctrl_f64=f64), f64)
