#!/usr/bin/env python3
"""Second probe: WHICH tensor of one captured forward + backward differs from its eager twin, and is
the captured version reproducible (replay vs replay, capture vs capture)?  Pure torch."""
import json
import sys

import torch
import torch.nn as nn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
OUT = int(sys.argv[2]) if len(sys.argv) > 2 else 12
torch.manual_seed(0)
net = nn.Sequential(nn.Linear(41, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, OUT)).cuda()
x = torch.randn(B, 41, device="cuda")
y = torch.randn(B, OUT, device="cuda")


def fwd_bwd():
    for p in net.parameters():
        p.grad = None
    out = net(x)
    loss = (out - y).pow(2).mean()
    loss.backward()
    return [out.detach().clone()] + [p.grad.detach().clone() for p in net.parameters()]


names = ["out"] + [n for n, _ in net.named_parameters()]
eager = [fwd_bwd(), fwd_bwd()]
res = {"B": B, "OUT": OUT, "blas": str(torch.backends.cuda.preferred_blas_library()), "eager_vs_eager": {n: float((a - b).abs().max()) for n, a, b in zip(names, *eager)}}


def capture(stream):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        outs = fwd_bwd()
    return g, outs


for label in ("capture_a", "capture_b"):
    g, outs = capture(torch.cuda.Stream())
    g.replay()
    torch.cuda.synchronize()
    first = [o.clone() for o in outs]
    g.replay()
    torch.cuda.synchronize()
    res[label + "_vs_eager"] = {n: float((a - b).abs().max()) for n, a, b in zip(names, first, eager[0])}
    res[label + "_replay_vs_replay"] = {n: float((a - b).abs().max()) for n, a, b in zip(names, first, outs)}
    res[label + "_rel"] = {n: float((a - b).abs().max() / b.abs().max()) for n, a, b in zip(names, first, eager[0])}
# eager on a NON-default stream: is it the stream, not the capture?
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    other = fwd_bwd()
torch.cuda.synchronize()
res["eager_other_stream_vs_eager"] = {n: float((a - b).abs().max()) for n, a, b in zip(names, other, eager[0])}
bad = {k: {n: v for n, v in d.items() if v != 0.0} for k, d in res.items() if isinstance(d, dict)}
print(json.dumps({"B": B, "OUT": OUT, "blas": res["blas"], "nonzero": {k: v for k, v in bad.items() if v}}))
