#!/usr/bin/env python3
"""Diagnostic: a few training rounds of config 5 (cwhh, 24 envs per morphology) printing, per round, whether anything went non-finite
and how far the actor / critic parameters moved.  Usage: train_alive_check.py [rounds=4] [family=cwhh]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgrl_amd import mjcf
from sgrl_amd.td3 import default_train_args
from sgrl_amd.train_loop import DeviceTrainer
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
family = sys.argv[2] if len(sys.argv) > 2 else "cwhh"
held = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
        "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
names = sorted(n for n in mjcf.list_assets() if (family == "cwhh" and n not in held) or n.split("_")[1] == family)
targs = default_train_args()
tr = DeviceTrainer(names, 24, args=targs, seed=3, device="cuda:0", max_buffer_size=400000, graph_updates=os.environ.get("GRAPH", "1") == "1")
steps = 0
while steps < 400:
    steps += 1
    if tr.collect_step(random_actions=True):
        tr.begin_round()
def snap(mod):
    return torch.cat([p.detach().flatten().double() for p in mod.parameters()])
a0, c0 = snap(tr.agent.actor), snap(tr.agent.critic)
for r in range(rounds):
    s = tr.train_round()
    a1, c1 = snap(tr.agent.actor), snap(tr.agent.critic)
    at, ct = snap(tr.agent.actor_target), snap(tr.agent.critic_target)
    print("round %d: return %.2f len %.1f iters %d | actor moved %.3e (finite %s) critic moved %.3e (finite %s) | targets finite %s %s | losses %s" % (
        r + 1, s["performance/train_return"], s["performance/train_length"], s["per_morph_iter"], float((a1 - a0).abs().sum()),
        bool(torch.isfinite(a1).all()), float((c1 - c0).abs().sum()), bool(torch.isfinite(c1).all()), bool(torch.isfinite(at).all()),
        bool(torch.isfinite(ct).all()), {k: (float(v) if torch.is_tensor(v) else v) for k, v in s.items() if k.startswith("loss")}), flush=True)
    ag = sum(float(p.grad.abs().sum()) for p in tr.agent.actor.parameters() if p.grad is not None)
    cg = sum(float(p.grad.abs().sum()) for p in tr.agent.critic.parameters() if p.grad is not None)
    na = sum(1 for p in tr.agent.actor.parameters() if p.grad is None)
    print("          actor |grad| %.3e (%d params without grad)  critic |grad| %.3e  last losses %s" % (ag, na, cg,
          {k: {kk: float(vv) for kk, vv in v.items() if torch.is_tensor(vv)} for k, v in list(tr.last_losses.items())[:2]}), flush=True)
    from sgrl_amd import td3 as _td3
    tot = sorted(float(e["scratch"][0]) for e in _td3._tables.values() if "scratch" in e)
    print("          squared gradient norms in the optimizer tables (min / median / max of %d): %.3e %.3e %.3e" % (
        len(tot), tot[0] if tot else -1, tot[len(tot) // 2] if tot else -1, tot[-1] if tot else -1), flush=True)
    a0, c0 = a1, c1
# the target critics on REAL replay rows through both paths (the synthetic rows of twin_target_check.py agree to 1e-6)
from sgrl_amd import set_policy
tr.agent.models2train()
for k in range(0, len(names), max(1, len(names) // 6)):
    b = tr.buffers[k].sample(tr.batch_size, generator=tr.gen)
    tr.agent.change_morphology(tr.graph_dicts[k])
    with torch.no_grad():
        na = tr.agent.actor_target(b["next_obs"])
        old = set_policy.TWIN_TARGETS
        set_policy.TWIN_TARGETS = True
        q1, q2 = tr.agent.critic_target(b["next_obs"], na)
        set_policy.TWIN_TARGETS = False
        r1, r2 = tr.agent.critic_target(b["next_obs"], na)
        set_policy.TWIN_TARGETS = old
    print("%-38s real rows: max |twin - rollout kernels| %.3e / %.3e  scale %.3e  |next_obs| max %.3e" % (
        names[k], float((q1 - r1).abs().max()), float((q2 - r2).abs().max()), float(r1.abs().max()), float(b["next_obs"].abs().max())), flush=True)
