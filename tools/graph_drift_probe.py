#!/usr/bin/env python3
"""Root-cause probe for the graph-lifetime drift recorded in round 1 (DESIGN 4 / VERDICT r1 #4): a
PPO-update graph kept alive across iterations drifted from the eager update when eager work
interleaved between replays.  Pure torch, no kernel of this repository.

A "trainer" holds actor + critic MLPs, Adam(capturable) and static minibatch buffers; `graph`
trainers replay one captured update, the `eager` trainer runs the same ops op by op.  All trainers
start from the same weights and see the same minibatches; between update phases the script runs
rollout-like eager work (allocations, GEMMs, frees) and, like round 1's PPO.train, captures and
resets a forward-only graph on the SAME capture stream.  After every phase the parameters of each
graph trainer are compared bit for bit with the eager trainer's.

Variants (one trainer each, all in one process so they see the same allocator history):
  base        what round 1 did: grads created inside capture (zero_grad(set_to_none=True)), shared
              capture stream, p.grad = None after every phase
  keepgrad    same, but p.grad is NOT cleared after the phase
  staticgrad  grads preallocated outside capture, zero_grad(set_to_none=False) inside
  ownstream   own capture stream (not shared with the forward graph)
  prewarm     one GEMM on the capture stream before the first capture (BLAS workspace from the
              ordinary pool)
"""
import argparse
import json
import sys
from copy import deepcopy

import torch
import torch.nn as nn


PERM_BUF = None
SCRATCH = None
FOREACH_ADAM = None
CLIP = "foreach"


def clip(params, max_norm=0.05):
    params = list(params)
    if CLIP == "none":
        return
    if CLIP == "foreach":
        torch.nn.utils.clip_grad_norm_(params, max_norm)
        return
    if CLIP == "single":
        torch.nn.utils.clip_grad_norm_(params, max_norm, foreach=False)
        return
    total = torch.sqrt(sum((p.grad * p.grad).sum() for p in params))        # "manual": plain elementwise ops
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for p in params:
        p.grad.mul_(coef)


def mlp(i, o):
    return nn.Sequential(nn.Linear(i, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, o)).cuda()


class Trainer:
    def __init__(self, actor, critic, batch, mode, streams):
        self.actor, self.critic = deepcopy(actor), deepcopy(critic)
        self.mode = mode
        cap = mode != "eager"
        # the SAME optimiser arithmetic everywhere: Adam(capturable=True) computes its bias corrections with
        # device tensors, capturable=False with python floats - last-bit differences that Adam amplifies
        self.opts = [torch.optim.Adam(m.parameters(), lr=1e-3, eps=1e-5, capturable=True, foreach=FOREACH_ADAM)
                     for m in (self.actor, self.critic)]
        self.obs = torch.zeros(batch, 41, device="cuda")
        self.act = torch.zeros(batch, 12, device="cuda")
        self.ret = torch.zeros(batch, 1, device="cuda")
        self.graph = None
        if cap:
            self._capture(streams)

    def params(self):
        return list(self.actor.parameters()) + list(self.critic.parameters())

    def body(self):
        loss = (self.actor(self.obs) - self.act).pow(2).mean() + 0.5 * (self.critic(self.obs) - self.ret).pow(2).mean()
        for o in self.opts:
            o.zero_grad(set_to_none=self.mode != "staticgrad")
        loss.backward()
        clip(self.actor.parameters())
        self.opts[0].step()
        clip(self.critic.parameters())
        self.opts[1].step()
        return loss.detach()

    def _capture(self, streams):
        side, cap = streams if self.mode != "ownstream" else (torch.cuda.Stream(), torch.cuda.Stream())
        keep = [deepcopy(m.state_dict()) for m in (self.actor, self.critic)]
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self.body()
        torch.cuda.current_stream().wait_stream(side)
        if self.mode == "prewarm":
            with torch.cuda.stream(cap):
                torch.mm(torch.randn(64, 64, device="cuda"), torch.randn(64, 64, device="cuda"))
            torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=cap):
            self.loss = self.body()
        for m, sd in zip((self.actor, self.critic), keep):
            m.load_state_dict(sd)
        for o in self.opts:
            for st in o.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()

    def update(self, obs, act, ret):
        self.obs.copy_(obs)
        self.act.copy_(act)
        self.ret.copy_(ret)
        if self.graph is None:
            return self.body()
        self.graph.replay()
        return self.loss

    def end_phase(self):
        if self.mode in ("base", "ownstream", "prewarm"):
            for p in self.params():
                p.grad = None


def rollout_like(actor, critic, streams, fwd_graph, heavy):
    """Eager work between two update phases: forward passes, scratch allocations of many sizes, and (as
    round 1's sample_vec did) a forward-only graph captured on the shared capture stream, replayed, reset."""
    with torch.no_grad():
        x = torch.randn(4096, 41, device="cuda")
        if fwd_graph:
            side, cap = streams
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                actor(x), critic(x)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):
                mu, v = actor(x), critic(x)
            for _ in range(20):
                g.replay()
            g.reset()
            del g, mu, v
        for i in range(30 if heavy else 3):
            a = actor(x)
            junk = [torch.randn(1 << (10 + (i + j) % 12), device="cuda") for j in range(6)]
            b = critic(x)
            del a, b, junk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--phases", type=int, default=6)
    ap.add_argument("--updates", type=int, default=12)
    ap.add_argument("--skip-rollout-at", type=int, default=3, help="phase before which the eager work is skipped")
    ap.add_argument("--fwd-graph", type=int, default=1)
    ap.add_argument("--modes", default="eager,eager,base,keepgrad,staticgrad,ownstream,prewarm")
    ap.add_argument("--detail", action="store_true")
    ap.add_argument("--snapshot", action="store_true")
    ap.add_argument("--between", default="compare", help="what runs between phases when comparing only at the end: "
                    "compare | none | cat | reduce | alloc | sync")
    ap.add_argument("--no-rollout", action="store_true", help="no eager work at all between phases")
    ap.add_argument("--clip", default="foreach", choices=["foreach", "single", "manual", "none"])
    ap.add_argument("--adam", default="default", choices=["default", "foreach", "single"])
    ap.add_argument("--reserve-mb", type=int, default=0, help="hold this much device memory across the captures, release after")
    ap.add_argument("--blas", default="", help="torch.backends.cuda.preferred_blas_library(...)")
    args = ap.parse_args()
    if args.blas:
        torch.backends.cuda.preferred_blas_library(args.blas)
    global FOREACH_ADAM, CLIP
    CLIP = args.clip
    FOREACH_ADAM = {"default": None, "foreach": True, "single": False}[args.adam]
    torch.manual_seed(0)
    global PERM_BUF
    PERM_BUF = torch.zeros(400000, dtype=torch.int64, device="cuda")
    global SCRATCH
    SCRATCH = torch.zeros(2 << 20, device="cuda")
    actor, critic = mlp(41, 12), mlp(41, 1)
    streams = (torch.cuda.Stream(), torch.cuda.Stream())
    modes = args.modes.split(",")
    # a forward graph on the shared capture stream BEFORE the update graphs exist (round 1's order: the first
    # rollout ran before the first update phase)
    if not args.no_rollout:
        rollout_like(actor, critic, streams, args.fwd_graph, heavy=True)
    reserve = torch.empty(args.reserve_mb << 20, dtype=torch.uint8, device="cuda") if args.reserve_mb else None
    trainers = [Trainer(actor, critic, args.batch, m, streams) for m in modes]
    del reserve          # its segments stay cached: address ranges that existed BEFORE the captures
    if args.between == "O":          # prime the allocator's large-block cache AFTER the captures, BEFORE the first replay
        prime = [torch.randn(333000, device="cuda") * 2 for _ in range(4)]
        torch.cuda.synchronize()
        del prime
    report = []
    gen = torch.Generator(device="cuda").manual_seed(5)
    for ph in range(args.phases):
        if ph != args.skip_rollout_at and not args.no_rollout:
            rollout_like(trainers[0].actor, trainers[0].critic, streams, args.fwd_graph, heavy=ph % 2 == 0)
        for u in range(args.updates):
            obs = torch.randn(args.batch, 41, device="cuda", generator=gen)
            act = torch.randn(args.batch, 12, device="cuda", generator=gen)
            ret = torch.randn(args.batch, 1, device="cuda", generator=gen)
            ls = [t.update(obs, act, ret) for t in trainers]
            if args.snapshot:
                print("LOSS", ph, u, [float(x) for x in ls], file=sys.stderr)
        last = ph == args.phases - 1
        if args.between != "compare" and not last:
            _ = None
            if args.between == "item":
                _ = torch.ones(4, device="cuda").sum().item()
            elif args.between == "item_only":
                _ = torch.ones(1, device="cuda")
                torch.cuda.synchronize()
                _ = _.item()
            elif args.between == "d2h":
                _ = torch.ones(1024, device="cuda").cpu()
            elif args.between == "cmp_noitem":
                a_ = torch.cat([p.detach().reshape(-1) for p in trainers[0].params()])
                b_ = torch.cat([p.detach().reshape(-1) for p in trainers[1].params()])
                _ = (a_ - b_).abs().max()
            elif args.between == "J":      # what PPO.train did between epochs: randperm on the host, H2D into a fresh block
                _ = torch.randperm(400000).to("cuda")
            elif args.between == "K":      # the same with the destination allocated once, before any capture
                PERM_BUF.copy_(torch.randperm(400000))
            elif args.between == "N":      # sync + kernel writes into a block that existed BEFORE the capture
                torch.cuda.synchronize()
                SCRATCH.normal_()
            elif args.between == "Q":      # which graph-visible tensor does the eager write land on?
                t1_ = trainers[1]
                vis = {}
                for tag, o_ in zip(("actor", "critic"), t1_.opts):
                    for i_, p_ in enumerate(o_.param_groups[0]["params"]):
                        vis[f"{tag}.{i_}.param"] = p_
                        if p_.grad is not None:
                            vis[f"{tag}.{i_}.grad"] = p_.grad
                        for k_, v_ in o_.state[p_].items():
                            vis[f"{tag}.{i_}.{k_}"] = v_
                vis.update(obs=t1_.obs, act=t1_.act, ret=t1_.ret, loss=t1_.loss)
                torch.cuda.synchronize()
                before = {k_: v_.detach().clone() for k_, v_ in vis.items()}
                ptrs = {k_: (v_.data_ptr(), v_.numel() * v_.element_size()) for k_, v_ in vis.items()}
                torch.cuda.synchronize()
                _ = torch.randn(333000, device="cuda") * 2
                torch.cuda.synchronize()
                lo_, hi_ = _.data_ptr(), _.data_ptr() + _.numel() * 4
                changed = [k_ for k_, v_ in vis.items() if not torch.equal(v_, before[k_])]
                overlap = [k_ for k_, (a_, n_) in ptrs.items() if a_ < hi_ and lo_ < a_ + n_]
                print("VICTIMS", ph, json.dumps({"changed": changed, "overlap": overlap, "eager_block": [lo_, hi_]}), file=sys.stderr)
            elif args.between in ("O", "P"):      # sync + kernel writes into a large block; O: the block was hipMalloc'ed (and cached)
                torch.cuda.synchronize()          # right after the capture, before any replay; P: same without that priming
                _ = torch.randn(333000, device="cuda") * 2
            elif args.between == "E":
                torch.cuda.synchronize()
                _ = [torch.empty(1 << (8 + i), device="cuda") for i in range(14)]
            elif args.between == "F":
                torch.cuda.current_stream().synchronize()
                a_ = torch.cat([p.detach().reshape(-1) for p in trainers[0].params()])
                b_ = torch.cat([p.detach().reshape(-1) for p in trainers[1].params()])
                _ = float((a_ - b_).abs().max())
            elif args.between == "G":
                torch.cuda.synchronize()
                _ = torch.cat([p.detach().reshape(-1) for p in trainers[0].params()])
            elif args.between == "H":
                torch.cuda.synchronize()
                _ = torch.cat([p.detach().reshape(-1) for p in trainers[1].params()])
            elif args.between == "I":
                torch.cuda.synchronize()
                _ = torch.randn(333000, device="cuda") * 2
            elif args.between in ("A", "B", "C", "D"):
                if args.between == "A":
                    torch.cuda.synchronize()
                if args.between == "C":
                    a_ = torch.randn(333000, device="cuda")
                    b_ = torch.randn(333000, device="cuda")
                elif args.between == "D":
                    a_ = torch.cat([p.detach().reshape(-1) for p in trainers[0].params()])
                    b_ = torch.cat([p.detach().reshape(-1) for p in trainers[0].params()])
                else:
                    a_ = torch.cat([p.detach().reshape(-1) for p in trainers[0].params()])
                    b_ = torch.cat([p.detach().reshape(-1) for p in trainers[1].params()])
                _ = float((a_ - b_).abs().max())
                if args.snapshot:
                    segs = torch.cuda.memory_snapshot()
                    priv = [(sg["address"], sg["address"] + sg["total_size"], sg.get("segment_pool_id")) for sg in segs
                            if tuple(sg.get("segment_pool_id", (0, 0))) != (0, 0)]
                    where = {}
                    for name, tns in (("a_", a_), ("b_", b_)):
                        ptr_ = tns.data_ptr()
                        where[name] = [str(pid) for lo, hi, pid in priv if lo <= ptr_ < hi]
                    g_ = trainers[1].params()[0].grad
                    where["graph_grad0_in_private_pool"] = [str(pid) for lo, hi, pid in priv
                                                            if g_ is not None and lo <= g_.data_ptr() < hi]
                    where["n_private_segments"] = len(priv)
                    print("SNAPSHOT", ph, json.dumps(where), file=sys.stderr)
            elif args.between == "cat":
                _ = torch.cat([p.detach().reshape(-1) for p in trainers[0].params()])
            elif args.between == "reduce":
                _ = torch.randn(1 << 20, device="cuda").max()
            elif args.between == "reduce_small":
                _ = torch.randn(256, device="cuda").max()
            elif args.between == "alloc":
                _ = [torch.empty(1 << (8 + i), device="cuda") for i in range(14)]
            elif args.between == "sync":
                torch.cuda.synchronize()
            del _
            continue
        torch.cuda.synchronize()
        ref = torch.cat([p.detach().reshape(-1) for p in trainers[0].params()])
        row = {"phase": ph, "rollout": ph != args.skip_rollout_at}
        for t, m in list(zip(trainers, modes))[1:]:
            cur = torch.cat([p.detach().reshape(-1) for p in t.params()])
            row[m if m != "eager" else "eager2"] = float((cur - ref).abs().max())
            if args.detail and m != "eager":
                st = {}
                for o_g, o_e, tag in zip(t.opts, trainers[0].opts, ("actor", "critic")):
                    for i, (pg, pe) in enumerate(zip(o_g.param_groups[0]["params"], o_e.param_groups[0]["params"])):
                        for k in o_g.state[pg]:
                            d_ = float((o_g.state[pg][k].float() - o_e.state[pe][k].float()).abs().max())
                            if d_:
                                st[f"{tag}.{i}.{k}"] = d_
                        if pg.grad is not None and pe.grad is not None and not torch.equal(pg.grad, pe.grad):
                            st[f"{tag}.{i}.grad"] = float((pg.grad - pe.grad).abs().max())
                row[m + "_state"] = st
                names = [("actor." + n) for n, _ in t.actor.named_parameters()] + [("critic." + n) for n, _ in t.critic.named_parameters()]
                row[m + "_detail"] = {n: float((a - b).abs().max()) for n, a, b in zip(names, t.params(), trainers[0].params())
                                      if not torch.equal(a, b)}
        report.append(row)
        for t in trainers:
            t.end_phase()
    print(json.dumps({"batch": args.batch, "updates_per_phase": args.updates, "max_abs_param_diff_vs_eager": report}))


if __name__ == "__main__":
    main()
