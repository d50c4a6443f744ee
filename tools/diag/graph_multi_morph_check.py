#!/usr/bin/env python3
"""Diagnostic: hipGraph-replayed TD3 updates over SEVERAL morphologies in turn (what DeviceTrainer does): how far actor and critic
move per update, graphed against eager, from the same start."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgrl_amd import graph as G, mjcf
from sgrl_amd.rollout import TRAV
from sgrl_amd.td3 import Agent, GraphedUpdates, default_train_args
names = sys.argv[1:] or ["3d_walker_7_full", "3d_hopper_3_shin", "3d_cheetah_10_tail_leftbleg"]
if names == ["cwhh"]:
    held = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
            "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
    names = sorted(n for n in mjcf.list_assets() if n not in held)
ROUNDS = int(os.environ.get("ROUNDS", "3"))
dev = torch.device("cuda:0")
targs = default_train_args()
B = targs.agent_batch_size
def batch_for(L, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    def ob():
        o = torch.randn((B, L, 41), device=dev, generator=g) * 0.5
        o[:, :, 3:5] = 0; o[:, :, 5] = -9.81; o[:, :, 8] = 0
        return o.reshape(B, 41 * L).contiguous()
    return {"obs": ob(), "next_obs": ob(), "action": torch.rand(B, 3 * L, device=dev, generator=g) * 2 - 1,
            "reward": torch.randn(B, 1, device=dev, generator=g), "done": torch.zeros(B, 1, device=dev)}
def snap(m):
    return torch.cat([p.detach().flatten().double() for p in m.parameters()])
def run(graphed):
    torch.manual_seed(0)
    agent = Agent(targs, device=dev)
    agent.models2train()
    ms = [mjcf.load_asset(n) for n in names]
    gds = [G.getGraphDict(m.parents, TRAV, [], device=dev) for m in ms]
    gu = GraphedUpdates(agent, B) if graphed else None
    it = 0
    if gu is not None:
        for k, (m, gd) in enumerate(zip(ms, gds)):
            agent.change_morphology(gd)
            gu.warm(k, gd, m.num_limbs, batch_for(m.num_limbs, 100 + k), iters=3, first_it=it)
            it += 3
    else:
        for k, (m, gd) in enumerate(zip(ms, gds)):
            agent.change_morphology(gd)
            for j in range(3):
                agent.update(batch_for(m.num_limbs, 100 + k), it, lazy_stats=True, skip_unused_critic_grads=True); it += 1
    moved = []
    for rnd in range(ROUNDS):
        if gu is not None and rnd > 0 and os.environ.get("FORCE_RECAPTURE"):      # what a changed workspace stamp does in the trainer
            from sgrl_amd import td3 as _td3
            for key, sl in gu.slots.items():
                for flag in list(sl["graphs"]):
                    del sl["graphs"][flag]
                    _td3.release_tables((id(gu), key, flag))
        if rnd > 0 and os.environ.get("ROLLOUT_BETWEEN"):                          # the collection between two rounds: no-grad actor forwards
            agent.models2eval()
            with torch.no_grad():
                for k2, (m2, gd2) in enumerate(zip(ms, gds)):
                    agent.change_morphology(gd2)
                    agent.actor(batch_for(m2.num_limbs, 7)["obs"][:24])
            agent.models2train()
        for k, (m, gd) in enumerate(zip(ms, gds)):
            for j in range(4):
                a0, c0 = snap(agent.actor), snap(agent.critic)
                b = batch_for(m.num_limbs, 1000 * rnd + 10 * k + j)
                if gu is not None:
                    out = gu.update(k, gd, m.num_limbs, b, it)
                else:
                    agent.change_morphology(gd)
                    out = agent.update(b, it, lazy_stats=True, skip_unused_critic_grads=True)
                it += 1
                moved.append((rnd, names[k][3:14], it - 1, float((snap(agent.actor) - a0).abs().sum()), float((snap(agent.critic) - c0).abs().sum()),
                              float(out["loss/critic_loss"])))
    return moved
for graphed in ((True,) if os.environ.get("ONLY_GRAPHED") else (False, True)):
    print("== graphed" if graphed else "== eager")
    for r in run(graphed):
        print("  round %d %-12s it %3d actor moved %10.3e critic moved %10.3e critic loss %.4e" % r)
