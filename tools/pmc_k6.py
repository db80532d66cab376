#!/usr/bin/env python3
"""Workload for the K6 counter passes: the return scan at [T=400, N=4096] (16-env workgroups), f32
rewards without statistics (SURVEY's 17 B/element) and f64 rewards with the fused statistics (the
config-5 form, 21 B/element), 40 launches each.  Run under
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 tools/pmc_k6.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir> -- python3 tools/pmc_k6.py
and summarise with tools/pmc_summary.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import torch  # noqa: E402

from olympic_hip import _abi  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402

eng = Engine(0)
dev = eng.device
g = torch.Generator(device="cuda").manual_seed(0)
T, N = 400, 4096
r, v, nv = (torch.empty((T, N), device=dev).normal_(0, 1, generator=g) for _ in range(3))
r64 = r.double()
fl = ((torch.rand((T, N), device=dev, generator=g) < 1 / 300).to(torch.uint8) * 3)
ret, adv = torch.empty_like(r), torch.empty_like(r)
st = torch.zeros(3, dtype=torch.float64, device=dev)
# spoil the caches between launches so that every launch fetches its inputs from memory
spoil = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for i in range(40):
    spoil.fill_(i & 1)
    eng.return_scan(_abi.SCAN_RETURN, 0.99, 0.97, r, v, nv, fl, ret, adv)
    spoil.fill_(i & 1)
    eng.return_scan(_abi.SCAN_RETURN, 0.99, 0.97, r64, v, nv, fl, ret, adv, stats3=st)
torch.cuda.synchronize()
print("done")
