import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from oracle.formula import synth_obs
from sgrl_amd import graph as G, mjcf
from sgrl_amd.rollout import TRAV
from sgrl_amd.td3 import Agent, GraphedUpdates, default_train_args
dev = torch.device("cuda:0")
names = sys.argv[1:] or ["3d_walker_7_full"]
agent = Agent(default_train_args(), device=dev)
agent.models2train()
gr = GraphedUpdates(agent, 100)
def cost(fn, n=10):
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(n): fn(i)
    c = (time.time() - t0) / n * 1e3
    torch.cuda.synchronize(); w = (time.time() - t0) / n * 1e3
    return c, w
data = []
for k, name in enumerate(names):       # warm EVERY morphology first (workspaces reach their final size), then capture
    m = mjcf.load_asset(name); L = m.num_limbs
    gd = G.getGraphDict(m.parents, TRAV, [], device=dev)
    batch = {"obs": torch.from_numpy(synth_obs(L, 100, 1).astype(np.float32)).to(dev), "next_obs": torch.from_numpy(synth_obs(L, 100, 2).astype(np.float32)).to(dev),
             "action": (torch.rand(100, 3 * L, device=dev) * 2 - 1), "reward": torch.randn(100, 1, device=dev), "done": torch.zeros(100, 1, device=dev)}
    gr.warm(k, gd, L, batch, iters=2)
    data.append((gd, L, batch))
for k, name in enumerate(names):
    gd, L, batch = data[k]
    for it in range(2): gr.update(k, gd, L, batch, it)
    sl = gr.slots[k]
    c, w = cost(lambda i: sl["graphs"][i % 2].replay())
    print("%-34s after capturing %2d morphologies: replay cpu %.1f ms wall %.1f ms" % (name, k + 1, c, w), flush=True)
for k in (0, 1, len(names) - 1):
    sl = gr.slots[k]
    c, w = cost(lambda i: sl["graphs"][i % 2].replay())
    print("morphology %d again: replay cpu %.1f ms wall %.1f ms" % (k, c, w))
    for flag in (0, 1):
        c, w = cost(lambda i: sl["graphs"][flag].replay())
        print("    flag %d only: wall %.1f ms" % (flag, w))
sl = gr.slots[0]
torch.cuda.synchronize()
for i in range(4): sl["graphs"][1].replay()
torch.cuda.synchronize()
