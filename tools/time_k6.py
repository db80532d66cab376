"""K6 timing at a few [T,N] shapes (HIP events; launches <= ~10 us are bounded by the Python
launch path here - use rocprofv3 --kernel-trace on this script for true kernel durations)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import torch

from olympic_hip._ffi import HipTimer
from olympic_hip.engine import Engine
eng = Engine(0); dev = eng.device
g = torch.Generator(device="cuda").manual_seed(0)
for T, N in ((400, 4096), (400, 32768), (100, 4096)):
    r, v, nv = (torch.empty((T, N), device=dev).normal_(0, 1, generator=g) for _ in range(3))
    fl = ((torch.rand((T, N), device=dev, generator=g) < 1 / 300).to(torch.uint8) * 3)
    ret, adv = torch.empty_like(r), torch.empty_like(r)
    for mode in (0, 1):
        f = lambda: eng.return_scan(mode, 0.99, 0.97, r, v, nv, fl, ret, adv)
        for _ in range(5): f()
        t = HipTimer(); s = eng.ctx.stream(); t.start(s)
        for _ in range(50): f()
        t.stop(s)
        print(os.environ.get("OLY_K6_VARIANT", "1"), T, N, mode, round(t.elapsed_ms() / 50 * 1e3, 1), "us")
