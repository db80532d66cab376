"""Run under torch.distributed.run with ONE rank and backend "nccl" (= RCCL): every collective the
multi-GPU path issues (barrier with device_ids, all-gather of the statistics triples, the fused gradient
all-reduce, the parameter broadcast, a MAX all-reduce) goes through RCCL on the real device, followed by
the device-side combine.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from olympic_hip import _abi  # noqa: E402
from olympic_hip import dist as odist  # noqa: E402
from olympic_hip.engine import Engine  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    dist.barrier(device_ids=[0])
    eng = Engine(0)
    torch.manual_seed(3)
    T, N = 50, 1024
    rew = torch.rand((T, N), device=dev)
    val = torch.randn((T, N), device=dev)
    nv = torch.randn((T, N), device=dev)
    fl = (torch.rand((T, N), device=dev) < 0.01).to(torch.uint8) * _abi.FLAG_LAST
    st = torch.zeros(3, dtype=torch.float64, device=dev)
    ret, adv = eng.return_scan(_abi.SCAN_RETURN, 0.99, 1.0, rew, val, nv, fl, stats3=st)
    parts = odist.gather_stats(st)                         # RCCL all_gather_into_tensor, f64 on the device
    assert parts.shape == (1, 3) and parts.is_cuda
    a = adv.clone()
    eng.adv_normalize(a, parts, ddof=1, eps=1e-5)
    ref = (adv.double() - adv.double().mean()) / (adv.double().std() + 1e-5)
    err = float((a.double() - ref).abs().max())
    lin = torch.nn.Linear(8, 4).to(dev)
    lin(torch.randn(16, 8, device=dev)).sum().backward()
    g0 = [p.grad.clone() for p in lin.parameters()]
    odist.allreduce_gradients(list(lin.parameters()))      # RCCL all_reduce of the flat gradient buffer
    same = all(torch.equal(g, p.grad) for g, p in zip(g0, lin.parameters()))
    odist.broadcast_parameters([lin])                      # RCCL broadcast
    w = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    print(json.dumps(dict(backend=dist.get_backend(), world=dist.get_world_size(), stats=parts.cpu().tolist(),
                          norm_err=err, grads_unchanged=bool(same), max=float(w.item()))))
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
