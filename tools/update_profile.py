#!/usr/bin/env python3
"""One morphology's TD3 update in isolation (synthetic batch of args.agent_batch_size = 256 rows, the reference's
configs/default.py:61; SGRL_UPDATE_BATCH=<n> overrides): eager timing and, under rocprofv3, the per-kernel
decomposition of the update's GPU time.  Usage: update_profile.py [morphology] [iters]; SGRL_GRAPH_UPDATES=1 replays hipGraphs."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgrl_amd import graph as G, mjcf
from sgrl_amd.rollout import TRAV
from sgrl_amd.td3 import Agent, GraphedUpdates, default_train_args
name = sys.argv[1] if len(sys.argv) > 1 else "3d_walker_7_full"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
targs = default_train_args()
agent = Agent(targs, device=dev)
m = mjcf.load_asset(name)
gd = G.getGraphDict(m.parents, TRAV, [], device=dev)
agent.change_morphology(gd)
agent.models2train()
B, L = int(os.environ.get("SGRL_UPDATE_BATCH", targs.agent_batch_size)), m.num_limbs
def synth_obs(seed):     # plausible magnitudes: positions / velocities O(1), the constant columns of the 41-float limb row
    g = torch.Generator(device=dev).manual_seed(seed)
    o = torch.randn((B, L, 41), device=dev, generator=g) * 0.5
    o[:, :, 3:5] = 0; o[:, :, 5] = -9.81; o[:, :, 8] = 0
    return o.reshape(B, 41 * L).contiguous()
batch = {"obs": synth_obs(1), "next_obs": synth_obs(2),
         "action": (torch.rand(B, 3 * L, device=dev) * 2 - 1), "reward": torch.randn(B, 1, device=dev), "done": torch.zeros(B, 1, device=dev)}
graphed = GraphedUpdates(agent, B) if os.environ.get("SGRL_GRAPH_UPDATES", "0") == "1" else None
if graphed is not None:
    graphed.warm(0, gd, L, batch, iters=3)
    for it in range(4):
        graphed.update(0, gd, L, batch, it)
else:
    for it in range(4):
        agent.update(batch, it, lazy_stats=True)
torch.cuda.synchronize()
t0 = time.time()
for it in range(iters):
    if graphed is not None:
        graphed.update(0, gd, L, batch, it)
    else:
        agent.update(batch, it, lazy_stats=True)
torch.cuda.synchronize()
ms = round((time.time() - t0) / iters * 1e3, 3)
chk = float(sum(p.detach().double().abs().sum() for p in list(agent.actor.parameters()) + list(agent.critic.parameters())))
print(json.dumps({"morphology": name, "limbs": L, "batch": B, "graphed": graphed is not None,
                  "split_graphs": bool(graphed is not None and graphed.split), "iters": iters, "ms_per_update": ms,
                  "param_abs_sum_after": chk}))
