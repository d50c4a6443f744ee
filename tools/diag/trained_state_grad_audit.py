#!/usr/bin/env python3
"""Are the update's gradients still right once the networks have TRAINED (saturated actions, grown weights, real replay rows)?
Trains hopper++ (3 morphologies x 64 environments, shipped path) for some rounds, then, per morphology, runs ONE update (it = 0) from
the trained state on a real replay batch three ways -- own kernels on the GPU (plain Agent.update), PyTorch's vendor kernels on the GPU
(train_ops.ENABLED = False, twin critics off), float64 on the CPU -- and prints the per-tensor relative errors of both float32 paths
against float64 (raw gradients where the update clips), plus losses and target statistics.  usage: trained_state_grad_audit.py [rounds=25] [seed=3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgrl_amd import graph as G, mjcf, td3, train_ops, set_policy
from sgrl_amd.td3 import Agent, default_train_args
from sgrl_amd.train_loop import DeviceTrainer

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 25
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
names = sorted(n for n in mjcf.list_assets() if "hopper" in n)
TRAV = ["pre", "inlcrs", "postlcrs"]
tr = DeviceTrainer(names, 64, args=default_train_args(), seed=seed, device="cuda:0", max_buffer_size=100000, graph_updates=True, lag_flag=False)
for _ in range(400):
    if tr.collect_step(random_actions=True):
        tr.begin_round()
for r in range(rounds):
    s = tr.train_round()
print("trained %d rounds: return %.1f length %.1f" % (rounds, s["performance/train_return"], s["performance/train_length"]), flush=True)
sd = {k: v.detach().clone() for k, v in tr.agent.state_dict().items()}


def fresh(device, dtype, use_hip):
    a = Agent(default_train_args(), device=device, use_hip=use_hip)
    a.load_state_dict({k: v.to(device) for k, v in sd.items()})
    if dtype == torch.float64:
        a.double()
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
        out = run()
    finally:
        td3.clip_and_step = real
    return got, out


def compare(tag, got, ref, names_):
    rows = []
    for g, r, n in zip(got, ref, names_):
        if r is None or g is None:
            continue
        nr = float(r.norm())
        if nr < 1e-20:
            continue
        rows.append((float((g - r).norm()) / nr, nr, n))
    rows.sort(reverse=True)
    tot = np.sqrt(sum(float(r.norm()) ** 2 for r in ref if r is not None))
    totg = np.sqrt(sum(float(g.norm()) ** 2 for g in got if g is not None))
    print("   %-22s total |g| %.4e (f64 %.4e, rel %.1e); worst tensors: %s" % (
        tag, totg, tot, abs(totg - tot) / tot, "; ".join("%s %.1e (|g| %.1e)" % (n.replace("transformer_encoder.", "").replace("self_attn.", "sa."), e, nr) for e, nr, n in rows[:4])), flush=True)


for k, name in enumerate(names):
    m = mjcf.load_asset(name)
    L = m.num_limbs
    batch = tr.buffers[k].sample(256, generator=tr.gen)
    noise = torch.zeros(256, 3 * L, device="cuda").normal_(0, 0.2)
    b64 = {kk: v.detach().double().cpu() for kk, v in batch.items()}
    ref = fresh("cpu", torch.float64, False)
    gd_cpu = G.getGraphDict(m.parents, TRAV, [], device=torch.device("cpu"))
    ref.change_morphology({kk: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for kk, v in gd_cpu.items()})
    gref, lref = grads_at_clip(ref, lambda: ref.update(b64, 0, noise=noise.double().cpu()))
    print("%s (L=%d): f64 critic_loss %.5f actor_loss %.5f |obs|max %.1f |action batch|max %.2f" % (
        name, L, float(lref["loss/critic_loss"]), float(lref["loss/actor_loss"]), float(batch["obs"].abs().max()), float(batch["action"].abs().max())), flush=True)
    gd = G.getGraphDict(m.parents, TRAV, [], device=torch.device("cuda:0"))
    for tag, enabled in (("own kernels", True), ("vendor kernels", False)):
        train_ops.ENABLED = enabled
        a = fresh("cuda:0", torch.float32, True)
        a.change_morphology(gd)
        g, l = grads_at_clip(a, lambda: a.update(batch, 0, noise=noise))
        print("   %s: critic_loss %.5f actor_loss %.5f" % (tag, float(l["loss/critic_loss"]), float(l["loss/actor_loss"])))
        compare(tag + " critic", g["critic"], gref["critic"], [n for n, _ in a.critic.named_parameters()])
        compare(tag + " actor", g["actor"], gref["actor"], [n for n, _ in a.actor.named_parameters()])
    train_ops.ENABLED = True
