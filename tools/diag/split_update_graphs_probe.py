import os, sys, time, json
os.environ["SGRL_SPLIT_UPDATE_GRAPHS"] = "1"
sys.path.insert(0, "/root/repo")
import torch
from sgrl_amd import graph as G, mjcf
from sgrl_amd.rollout import TRAV
from sgrl_amd.td3 import Agent, GraphedUpdates, default_train_args
dev = torch.device("cuda:0"); torch.manual_seed(0)
agent = Agent(default_train_args(), device=dev)
m = mjcf.load_asset("3d_walker_7_full"); gd = G.getGraphDict(m.parents, TRAV, [], device=dev)
agent.change_morphology(gd); agent.models2train()
B, L = 100, m.num_limbs
o = lambda: (torch.randn((B, 41 * L), device=dev) * 0.5).contiguous()
batch = {"obs": o(), "next_obs": o(), "action": torch.rand(B, 3 * L, device=dev) * 2 - 1, "reward": torch.randn(B, 1, device=dev), "done": torch.zeros(B, 1, device=dev)}
gu = GraphedUpdates(agent, B)
gu.warm(0, gd, L, batch, iters=3)
for it in range(4): gu.update(0, gd, L, batch, it)
torch.cuda.synchronize()
sl = gu.slots[0]
def t(fn, n=50):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
for flag in (0, 1):
    g = sl["graphs"][flag]
    print("flag", flag, "g1 %.3f g2 %.3f g3 %.3f ms" % (t(g[0].replay), t(g[1].replay), t(g[2].replay)))
    cur = torch.cuda.current_stream()
    def both():
        gu._side.wait_stream(cur); g[0].replay()
        with torch.cuda.stream(gu._side): g[1].replay()
        cur.wait_stream(gu._side)
    def seq():
        g[0].replay(); g[1].replay()
    print("   g1+g2 sequential %.3f, side by side %.3f ms" % (t(seq), t(both)))
it = [4]
def upd():
    gu.update(0, gd, L, batch, it[0]); it[0] += 1
print("full update() alternating flags: %.3f ms" % t(upd, 100))
def raw():
    for flag in (0, 1):
        g = sl["graphs"][flag]
        gu._side.wait_stream(cur); g[0].replay()
        with torch.cuda.stream(gu._side): g[1].replay()
        cur.wait_stream(gu._side); g[2].replay()
print("raw replays of both flags / 2: %.3f ms" % (t(raw, 50) / 2))
def raw_load():
    for flag in (0, 1):
        gu._load(sl, batch)
        g = sl["graphs"][flag]
        gu._side.wait_stream(cur); g[0].replay()
        with torch.cuda.stream(gu._side): g[1].replay()
        cur.wait_stream(gu._side); g[2].replay()
print("the same with the batch load in front: %.3f ms" % (t(raw_load, 50) / 2))
