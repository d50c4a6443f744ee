#!/usr/bin/env python3
"""Diagnostic: which torch operators launch the small elementwise kernels of one TD3 update (eager, torch.profiler)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from oracle.formula import synth_obs
from sgrl_amd import graph as G, mjcf
from sgrl_amd.rollout import TRAV
from sgrl_amd.td3 import Agent, default_train_args
dev = torch.device("cuda:0")
torch.manual_seed(0)
agent = Agent(default_train_args(), device=dev)
m = mjcf.load_asset("3d_walker_7_full")
gd = G.getGraphDict(m.parents, TRAV, [], device=dev)
agent.change_morphology(gd)
agent.models2train()
B, L = 100, m.num_limbs
batch = {"obs": torch.from_numpy(synth_obs(L, B, 1).astype(np.float32)).to(dev), "next_obs": torch.from_numpy(synth_obs(L, B, 2).astype(np.float32)).to(dev),
         "action": (torch.rand(B, 3 * L, device=dev) * 2 - 1), "reward": torch.randn(B, 1, device=dev), "done": torch.zeros(B, 1, device=dev)}
for it in range(4):
    agent.update(batch, it, lazy_stats=True, skip_unused_critic_grads=True) if "skip_unused_critic_grads" in agent.update.__code__.co_varnames else agent.update(batch, it, lazy_stats=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for it in range(4, 6):
        agent.update(batch, it, lazy_stats=True, skip_unused_critic_grads=True) if "skip_unused_critic_grads" in agent.update.__code__.co_varnames else agent.update(batch, it, lazy_stats=True)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    if e.count and not e.key.startswith("sgrl") and (e.device_time_total > 0 or "fill" in e.key or "copy" in e.key or "add" in e.key or "zero" in e.key or "cat" in e.key):
        rows.append((e.count / 2.0, e.key, e.device_time_total / 2.0))
rows.sort(reverse=True)
for c, k, t in rows[:40]:
    print("%7.1f per update  %-60s %8.1f us device" % (c, k[:60], t))
