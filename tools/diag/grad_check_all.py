#!/usr/bin/env python3
"""Every shipped morphology: one TD3 update (it = 0: critic + actor + targets) at default-like weights on the device path the
trainer runs (own kernels, deferred weight gradients, the critic's unused gradients skipped, table optimizer; then the same through
GraphedUpdates: two eager warm-ups + captured replays) against the SAME module in float64 on the CPU (whose float32 form is pinned
to the reference's Agent.update by tests/golden/td3_update*.npz).  Prints, per morphology, the worst per-tensor relative error of the
raw gradients (relative to the tensor's own norm) and of the parameter steps.   usage: grad_check_all.py [batch=128] [name-filter]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle.formula import apply_default_like_, scripted_batch
from sgrl_amd import graph as G, mjcf, td3
from sgrl_amd.td3 import Agent, GraphedUpdates, default_train_args

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
flt = sys.argv[2] if len(sys.argv) > 2 else ""
TRAV = ["pre", "inlcrs", "postlcrs"]


def make(device, dtype, use_hip):
    torch.manual_seed(0)
    a = Agent(default_train_args(), device=device, use_hip=use_hip)
    for mod in (a.actor, a.critic):
        apply_default_like_(mod, 6)
    if dtype == torch.float64:
        a.double()
    with torch.no_grad():
        for tgt, src in ((a.actor_target, a.actor), (a.critic_target, a.critic)):
            for tp, sp in zip(tgt.parameters(), src.parameters()):
                tp.copy_(sp)
    a.models2train()
    return a


def grads_at_clip(agent, run):
    got = {}
    real = td3.clip_and_step

    def spy(opt, max_norm):
        which = "critic" if opt is agent.critic_optimizer else "actor"
        got[which] = [None if p.grad is None else p.grad.detach().double().cpu().clone() for p in getattr(agent, which).parameters()]
        return real(opt, max_norm)
    td3.clip_and_step = spy
    try:
        run()
    finally:
        td3.clip_and_step = real
    return got


def rel_worst(got, ref, names):
    worst, where = 0.0, ""
    for g, r, n in zip(got, ref, names):
        if r is None or g is None:
            assert (r is None) == (g is None), n
            continue
        nr = float(r.norm())
        if nr < 1e-30:
            continue
        e = float((g - r).norm()) / nr
        if e > worst and nr > 1e-14:
            worst, where = e, n
    return worst, where


names = sorted(n for n in mjcf.list_assets() if any(f in n for f in flt.split(",")))
ref_agent = make("cpu", torch.float64, False)
state0 = {k: v.clone() for k, v in ref_agent.state_dict().items()}
for name in names:
    m = mjcf.load_asset(name)
    L = m.num_limbs
    rows = scripted_batch(L, B, 11 + L)
    noise = torch.zeros(B, 3 * L).normal_(0, 0.2, generator=torch.Generator().manual_seed(5))
    t0 = time.time()
    # float64 CPU reference of the same module
    ref_agent.load_state_dict(state0)
    ref_agent.actor_optimizer = torch.optim.Adam(ref_agent.actor.parameters(), lr=ref_agent.args.lr)
    ref_agent.critic_optimizer = torch.optim.Adam(ref_agent.critic.parameters(), lr=ref_agent.args.lr)
    gd64 = G.getGraphDict(m.parents, TRAV, [], device=torch.device("cpu"))
    ref_agent.change_morphology({k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in gd64.items()})
    b64 = {k: torch.from_numpy(v).double() for k, v in rows.items()}
    before = [p.detach().clone() for nm in ("actor", "critic") for p in getattr(ref_agent, nm).parameters()]
    gref = grads_at_clip(ref_agent, lambda: ref_agent.update(b64, 0, noise=noise.double()))
    step_ref = [float((p.detach() - q).norm()) for p, q in zip([p for nm in ("actor", "critic") for p in getattr(ref_agent, nm).parameters()], before)]
    t_ref = time.time() - t0
    out = [name, "L=%d" % L]
    for mode in ("plain", "trainer", "graphed"):
        agent = make("cuda:0", torch.float32, True)
        gd = G.getGraphDict(m.parents, TRAV, [], device=torch.device("cuda:0"))
        agent.change_morphology(gd)
        bd = {k: torch.from_numpy(v).cuda() for k, v in rows.items()}
        nz = noise.cuda()
        before_d = [p.detach().double().cpu().clone() for nm in ("actor", "critic") for p in getattr(agent, nm).parameters()]
        if mode == "plain":
            g = grads_at_clip(agent, lambda: agent.update(bd, 0, noise=nz))
        elif mode == "trainer":
            for opt in (agent.actor_optimizer, agent.critic_optimizer):
                for grp in opt.param_groups:
                    grp["capturable"] = True
            g = grads_at_clip(agent, lambda: agent.update(bd, 0, noise=nz, lazy_stats=True, skip_unused_critic_grads=True))
        else:
            # the graph path end to end: ONE update at it = 0 cannot be replayed without warm-ups, so the comparison here is on the
            # STEP of a replayed update against an eager trainer-path update from the same state: run 2 warm-ups + 2 replays on one
            # agent, the same four updates eagerly on a second, compare parameters
            gu = GraphedUpdates(agent, B)
            agent2 = make("cuda:0", torch.float32, True)
            agent2.change_morphology(gd)
            for opt in (agent2.actor_optimizer, agent2.critic_optimizer):
                for grp in opt.param_groups:
                    grp["capturable"] = True
            torch.manual_seed(77)
            gu.warm(0, gd, L, bd, iters=2)
            for it in (2, 3, 4, 5):
                gu.update(0, gd, L, bd, it)
            torch.manual_seed(77)
            nzs = []
            for it in range(6):
                nz2 = torch.zeros(B, 3 * L, device="cuda").normal_(0, 0.2)      # GraphedUpdates._load draws the same way
                agent2.update(bd, it, noise=nz2, lazy_stats=True, skip_unused_critic_grads=True)
            pa = [p.detach().double().cpu() for nm in ("actor", "critic", "actor_target", "critic_target") for p in getattr(agent, nm).parameters()]
            pb = [p.detach().double().cpu() for nm in ("actor", "critic", "actor_target", "critic_target") for p in getattr(agent2, nm).parameters()]
            worst = max(float((x - y).abs().max()) for x, y in zip(pa, pb))
            moved = max(float((x - y).abs().max()) for x, y in zip(pa[:len(before_d)], before_d))
            out.append("graphed-vs-eager after 6 updates: max |dp| %.2e (moved %.2e)" % (worst, moved))
            continue
        pn = [n for nm in ("actor", "critic") for n, _ in getattr(agent, nm).named_parameters()]
        wa, na = rel_worst(g["actor"], gref["actor"], [n for n, _ in agent.actor.named_parameters()])
        wc, nc = rel_worst(g["critic"], gref["critic"], [n for n, _ in agent.critic.named_parameters()])
        after_d = [p.detach().double().cpu() for nm in ("actor", "critic") for p in getattr(agent, nm).parameters()]
        step_d = [float((p - q).norm()) for p, q in zip(after_d, before_d)]
        ws = max(abs(a - b) / max(b, 1e-12) for a, b in zip(step_d, step_ref) if b > 1e-9)
        out.append("%s: grad actor %.1e (%s) critic %.1e (%s) step %.1e" % (mode, wa, na.split(".")[-2] if na else "", wc, nc.split(".")[-2] if nc else "", ws))
    print(" | ".join(out), "| cpu ref %.0f s" % t_ref, flush=True)
