import os, sys, json, time
sys.path.insert(0, "olympics-mujoco_amd")
import torch
from olympic_hip import _abi
from olympic_hip.engine import Engine
from olympic_hip._ffi import HipTimer
eng = Engine(0); dev = eng.device
out = {}
for (T, N) in ((400, 524288), (400, 262144), (400, 131072), (400, 65536), (400, 32768), (400, 4096)):
    torch.manual_seed(0)
    r = torch.rand((T, N), device=dev); v = torch.randn((T, N), device=dev); nv = torch.randn((T, N), device=dev)
    fl = ((torch.rand((T, N), device=dev) < 0.003).to(torch.uint8) * 3)
    ret = torch.empty_like(r); adv = torch.empty_like(r)
    st = torch.zeros(3, dtype=torch.float64, device=dev)
    r64 = r.double()
    for mode, name, b in ((_abi.SCAN_RETURN, "return", 17), (_abi.SCAN_GAE, "gae", 21), (_abi.SCAN_RETURN, "return_f64_stats", 21)):
        f = (lambda: eng.return_scan(mode, 0.99, 0.97, r64, v, nv, fl, ret=ret, adv=adv, stats3=st)) if name.endswith('stats') else (lambda: eng.return_scan(mode, 0.99, 0.97, r, v, nv, fl, ret=ret, adv=adv))
        for _ in range(3): f()
        torch.cuda.synchronize()
        t = HipTimer(); s = torch.cuda.current_stream().cuda_stream
        t.start(s)
        for _ in range(20): f()
        t.stop(s); torch.cuda.synchronize()
        ms = t.elapsed_ms() / 20
        out[f"{name}_[{T},{N}]"] = dict(ms=round(ms, 4), TBps=round(b * T * N / ms / 1e9, 3))
print(os.environ.get("OLY_K6_PIPE", "auto"), json.dumps(out))
