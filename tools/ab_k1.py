#!/usr/bin/env python3
"""Interleaved A/B timing of K1 builds in ONE process (cdna_hip_programming.md rule 24).

    python tools/ab_k1.py lib1.so[:ROWS[:WG]] lib2.so[:ROWS[:WG]] ... [--rounds 8] [--steps 20]

Each variant is a separately built libolympic_hip*.so (e.g. -DOLY_K1_NT=1); ROWS / WG set
OLY_K1_ROWS / OLY_K1_WG_PER_CU before that library is loaded (they are read once per library).
Prints median / min kernel time per variant over the rounds.
"""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "olympics-mujoco_amd"))
import torch  # noqa: E402

from olympic_hip import _abi, specs  # noqa: E402
from olympic_hip.synthetic import h1_synthetic_block  # noqa: E402


class Lib:
    def __init__(self, spec_str, il_spec):
        parts = spec_str.split(":")
        path = parts[0]
        if not os.path.isabs(path):
            path = os.path.join(ROOT, "olympics-mujoco_amd", "lib", path)
        os.environ.pop("OLY_K1_ROWS", None)
        os.environ.pop("OLY_K1_WG_PER_CU", None)
        if len(parts) > 1 and parts[1]:
            os.environ["OLY_K1_ROWS"] = parts[1]
        if len(parts) > 2 and parts[2]:
            os.environ["OLY_K1_WG_PER_CU"] = parts[2]
        self.name = spec_str
        # a private copy so that dlopen gives every variant its own statics
        import shutil
        import tempfile
        tmp = tempfile.NamedTemporaryFile(suffix=".so", delete=False).name
        shutil.copy(path, tmp)
        self.L = C.CDLL(tmp)
        for n, (res, args) in _abi.SIGNATURES.items():
            f = getattr(self.L, n)
            f.restype, f.argtypes = res, args
        self.h = C.c_void_p()
        assert self.L.oly_create(C.byref(self.h), 0) == 0
        assert self.L.oly_il_configure(self.h, C.byref(il_spec.to_c())) == 0
        self.ev = [C.c_void_p(), C.c_void_p()]
        for e in self.ev:
            self.L.oly_event_create(C.byref(e))

    def run(self, bufs, steps, stream):
        p = lambda t: C.c_void_p(t.data_ptr())
        T, N = bufs["qpos"].shape[:2]
        self.L.oly_event_record(self.ev[0], stream)
        for i in range(steps):
            rc = self.L.oly_il_step(self.h, T, N, p(bufs["qpos"]), p(bufs["qvel"]), p(bufs["act"]), None,
                                    p(bufs["prev"][i & 1]), p(bufs["prev"][(i + 1) & 1]), p(bufs["obs"]),
                                    p(bufs["reward"]), p(bufs["absorbing"]), None, p(bufs["ctrl"]), 0, stream)
            assert rc == 0, self.L.oly_last_error(self.h)
        self.L.oly_event_record(self.ev[1], stream)
        self.L.oly_event_sync(self.ev[1])
        ms = C.c_float()
        self.L.oly_event_elapsed_ms(self.ev[0], self.ev[1], C.byref(ms))
        return ms.value / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--T", type=int, default=400)
    ap.add_argument("--N", type=int, default=4096)
    a = ap.parse_args()
    spec = specs.unitree_h1("walk")
    dev = torch.device("cuda", 0)
    qpos, qvel, act = h1_synthetic_block(spec, a.T, a.N, seed=1234)
    T, N = a.T, a.N
    bufs = dict(qpos=torch.as_tensor(qpos).to(dev), qvel=torch.as_tensor(qvel).to(dev),
                act=torch.as_tensor(act).to(dev),
                prev=[torch.full((N,), 1.25, dtype=torch.float64, device=dev),
                      torch.empty(N, dtype=torch.float64, device=dev)],
                obs=torch.empty((T, N, 32), dtype=torch.float32, device=dev),
                reward=torch.empty((T, N), dtype=torch.float32, device=dev),
                absorbing=torch.empty((T, N), dtype=torch.uint8, device=dev),
                ctrl=torch.empty((T, N, 11), dtype=torch.float32, device=dev))
    libs = [Lib(s, spec) for s in a.libs]
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    res = {l.name: [] for l in libs}
    for l in libs:                       # warm every variant (and the clocks)
        l.run(bufs, 10, stream)
    for r in range(a.rounds):
        for l in libs:
            res[l.name].append(l.run(bufs, a.steps, stream))
    rows = T * N
    print(f"{'variant':40s} {'median_us':>10s} {'min_us':>9s} {'Genv/s':>8s} {'frac(493B)':>10s}")
    for n, v in res.items():
        med, mn = statistics.median(v), min(v)
        print(f"{n:40s} {med*1e3:10.1f} {mn*1e3:9.1f} {rows/med/1e6:8.2f} {493*rows/(med*1e-3)/8e12:10.3f}")


if __name__ == "__main__":
    main()
