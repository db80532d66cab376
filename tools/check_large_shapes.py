#!/usr/bin/env python3
"""Shapes whose element counts pass 2^31 (the MI355X holds them: 288 GB): K1 on [400, 524288] H1 rows
(3.6e9 qpos elements), K6 / K7 on [400, 6 Mi] (2.5e9 elements).  Environments are independent, so a run on
a copied column slice must reproduce the big run's columns bit for bit; an index that wrapped at 32 bits
would not.  Too large for the test suite (about 80 GB of device memory): run by hand."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import torch  # noqa: E402

from olympic_hip import _abi, specs  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402


def main():
    eng = Engine(0)
    dev = eng.device
    sp = specs.unitree_h1("walk")
    eng.il_configure(sp)
    T, N = 400, 524288
    g = torch.Generator(device=dev).manual_seed(1)
    qpos = torch.randn((T, N, sp.nq), dtype=torch.float64, device=dev, generator=g) * 0.2
    qvel = torch.randn((T, N, sp.nv), dtype=torch.float64, device=dev, generator=g)
    act = torch.rand((T, N, sp.n_act), device=dev, generator=g) * 2 - 1
    prev = torch.rand(N, dtype=torch.float64, device=dev, generator=g)
    big = eng.il_step(qpos, qvel, act, prev)
    torch.cuda.synchronize()
    assert qpos.numel() > 2 ** 31
    for lo in (0, N // 2 - 64, N - 256, N - 128):
        sl = slice(lo, lo + 128)
        small = eng.il_step(qpos[:, sl].contiguous(), qvel[:, sl].contiguous(), act[:, sl].contiguous(), prev[sl].contiguous())
        for k in ("obs", "reward", "absorbing", "fall_code", "ctrl"):
            assert torch.equal(big[k][:, sl], small[k]), (k, lo)
        assert torch.equal(big["prev"][sl], small["prev"]), lo
    print("K1  [400, 524288]: columns reproduce (", qpos.numel(), "qpos elements )")
    del qpos, qvel, act, big, small
    torch.cuda.empty_cache()

    T, N = 400, 6 * 1024 * 1024
    r = torch.rand((T, N), device=dev, generator=g)
    v = torch.randn((T, N), device=dev, generator=g)
    nv = torch.randn((T, N), device=dev, generator=g)
    fl = (torch.rand((T, N), device=dev, generator=g) < 0.004).to(torch.uint8) * _abi.FLAG_LAST
    assert r.numel() > 2 ** 31
    for mode in (_abi.SCAN_RETURN, _abi.SCAN_GAE):
        st = torch.zeros(3, dtype=torch.float64, device=dev)
        ret, adv = eng.return_scan(mode, 0.99, 0.97, r, v, nv, fl, stats3=st if mode == _abi.SCAN_RETURN else None)
        torch.cuda.synchronize()
        for lo in (0, N - 4096, N // 2):
            sl = slice(lo, lo + 4096)
            r2, a2 = eng.return_scan(mode, 0.99, 0.97, r[:, sl].contiguous(), v[:, sl].contiguous(),
                                     nv[:, sl].contiguous(), fl[:, sl].contiguous())
            assert torch.equal(ret[:, sl], r2) and torch.equal(adv[:, sl], a2), (mode, lo)
        if mode == _abi.SCAN_RETURN:
            s = st.cpu().numpy()
            ref = adv.double().sum().item(), (adv.double() ** 2).sum().item()
            assert s[0] == T * N and abs(s[1] - ref[0]) <= 1e-9 * abs(ref[0]) + 1e-3 and abs(s[2] - ref[1]) <= 1e-9 * ref[1]
            a = adv.clone()
            eng.adv_normalize(a, st, 1, 1e-5)
            m, sd = adv.double().mean().item(), adv.double().std().item()
            chk = ((adv[:, -4096:].double() - m) / (sd + 1e-5)).float()
            assert (a[:, -4096:] - chk).abs().max().item() < 1e-5
            del a
        del ret, adv
    print("K6 / K7  [400, 6291456]: columns reproduce, statistics agree (", r.numel(), "elements )")


if __name__ == "__main__":
    main()
