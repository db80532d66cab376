#!/usr/bin/env python3
"""Third probe: one full update (forward, backward, grad clip, Adam capturable) captured vs eager, step by
step: after which step, and in which tensor (gradient / exp_avg / exp_avg_sq / parameter), do they part?"""
import json
import sys
from copy import deepcopy

import torch
import torch.nn as nn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
FOREACH = None if len(sys.argv) < 3 or sys.argv[2] == "-" else bool(int(sys.argv[2]))
TWO = len(sys.argv) > 3 and sys.argv[3] == "two"
torch.manual_seed(0)
net0 = nn.Sequential(nn.Linear(41, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 12)).cuda()
crit0 = nn.Sequential(nn.Linear(41, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 1)).cuda()


class T:
    def __init__(self, graph):
        self.net = deepcopy(net0)
        self.opt = torch.optim.Adam(self.net.parameters(), lr=1e-3, eps=1e-5, capturable=True, foreach=FOREACH)
        self.crit = deepcopy(crit0)
        self.opt2 = torch.optim.Adam(self.crit.parameters(), lr=1e-3, eps=1e-5, capturable=True, foreach=FOREACH)
        self.r = torch.zeros(B, 1, device="cuda")
        self.x = torch.zeros(B, 41, device="cuda")
        self.y = torch.zeros(B, 12, device="cuda")
        self.g = None
        if graph:
            keep = deepcopy(self.net.state_dict())
            keep2 = deepcopy(self.crit.state_dict())
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    self.body()
            torch.cuda.current_stream().wait_stream(s)
            self.g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g, stream=torch.cuda.Stream()):
                self.body()
            self.net.load_state_dict(keep)
            self.crit.load_state_dict(keep2)
            for o in (self.opt, self.opt2):
                for st in o.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()

    def body(self):
        loss = (self.net(self.x) - self.y).pow(2).mean()
        if TWO:
            loss = loss + 0.5 * (self.crit(self.x) - self.r).pow(2).mean()
        self.opt.zero_grad(set_to_none=True)
        self.opt2.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.net.parameters(), 0.05)
        self.opt.step()
        if TWO:
            torch.nn.utils.clip_grad_norm_(self.crit.parameters(), 0.05)
            self.opt2.step()

    def step(self, x, y):
        self.x.copy_(x)
        self.y.copy_(y)
        self.r.copy_(y[:, :1])
        if self.g is None:
            self.body()
        else:
            self.g.replay()

    def snap(self):
        out = {}
        named = list(self.net.named_parameters()) + ([("c" + n, p) for n, p in self.crit.named_parameters()] if TWO else [])
        for (n, p) in named:
            out["param." + n] = p.detach().clone()
            out["grad." + n] = p.grad.detach().clone()
            for k, v in (self.opt.state[p] if p in self.opt.state else self.opt2.state[p]).items():
                out[k + "." + n] = v.detach().clone().float()
        return out


a, b = T(False), T(True)
gen = torch.Generator(device="cuda").manual_seed(1)
rep = []
for it in range(int(sys.argv[4]) if len(sys.argv) > 4 else 4):
    x = torch.randn(B, 41, device="cuda", generator=gen)
    y = torch.randn(B, 12, device="cuda", generator=gen)
    a.step(x, y)
    b.step(x, y)
    torch.cuda.synchronize()
    sa, sb = a.snap(), b.snap()
    diff = {k: float((sa[k] - sb[k]).abs().max()) for k in sa}
    rep.append({"step": it, "differing": {k: v for k, v in diff.items() if v != 0.0}})
print(json.dumps({"B": B, "foreach": FOREACH, "two": TWO, "report": rep}))
