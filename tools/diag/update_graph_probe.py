#!/usr/bin/env python3
"""Can one TD3 update (PyTorch autograd + the ctypes HIP target path) be captured into a hipGraph and replayed?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from sgrl_amd import graph as G, mjcf
from sgrl_amd.td3 import Agent, default_train_args
from oracle.formula import synth_obs
args = default_train_args()
agent = Agent(args, device="cuda:0")
for opt in (agent.actor_optimizer, agent.critic_optimizer):
    for g in opt.param_groups:
        g["capturable"] = True
m = mjcf.load_asset("3d_walker_7_full")
gd = G.getGraphDict(m.parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cuda:0"))
agent.change_morphology(gd)
B, L = 100, 7
batch = {"obs": torch.from_numpy(synth_obs(L, B, 1).astype(np.float32)).cuda(), "next_obs": torch.from_numpy(synth_obs(L, B, 2).astype(np.float32)).cuda(),
         "action": torch.rand(B, 3 * L, device="cuda") * 2 - 1, "reward": torch.randn(B, 1, device="cuda"), "done": torch.zeros(B, 1, device="cuda")}
agent.models2train()
noise = torch.zeros(B, 3 * L, device="cuda")

def step(it):
    noise.normal_(0, args.policy_noise)
    return agent.update(batch, it, noise=noise, lazy_stats=True)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for it in range(4):
        step(it)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graphs = {}
for flag in (0, 1):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step(flag)
    graphs[flag] = (g, out)
torch.cuda.synchronize()
print("captured")
before = [p.detach().clone() for p in agent.critic.parameters()]
t0 = time.time()
for it in range(40):
    graphs[it % 2][0].replay()
torch.cuda.synchronize()
print("ms per update (graph replay): %.2f" % ((time.time() - t0) / 40 * 1e3))
moved = max(float((p - q).abs().max()) for p, q in zip(agent.critic.parameters(), before))
print("critic moved by", moved, "critic_loss", float(graphs[0][1]["loss/critic_loss"]))
