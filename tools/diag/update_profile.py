#!/usr/bin/env python3
"""How launch-bound is one TD3 update?  Counts kernels and GPU-busy time of 20 updates on walker_7 (B = agent_batch_size = 256)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
from sgrl_amd import graph as G, mjcf
from sgrl_amd.td3 import Agent, default_train_args
from oracle.formula import synth_obs
import numpy as np
args = default_train_args()
agent = Agent(args, device="cuda:0")
m = mjcf.load_asset("3d_walker_7_full")
gd = G.getGraphDict(m.parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cuda:0"))
agent.change_morphology(gd)
B, L = args.agent_batch_size, 7
batch = {"obs": torch.from_numpy(synth_obs(L, B, 1).astype(np.float32)).cuda(), "next_obs": torch.from_numpy(synth_obs(L, B, 2).astype(np.float32)).cuda(),
         "action": torch.rand(B, 3 * L, device="cuda") * 2 - 1, "reward": torch.randn(B, 1, device="cuda"), "done": torch.zeros(B, 1, device="cuda")}
agent.models2train()
for it in range(4):
    agent.update(batch, it)
torch.cuda.synchronize()
t0 = time.time()
for it in range(20):
    agent.update(batch, it)
torch.cuda.synchronize()
print("ms per update (eager): %.2f" % ((time.time() - t0) / 20 * 1e3))
