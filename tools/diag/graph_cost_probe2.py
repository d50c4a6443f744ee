import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from oracle.formula import synth_obs
from sgrl_amd import graph as G, mjcf
from sgrl_amd.rollout import TRAV
from sgrl_amd.td3 import Agent, GraphedUpdates, default_train_args
dev = torch.device("cuda:0")
agent = Agent(default_train_args(), device=dev)
agent.models2train()
gr = GraphedUpdates(agent, 100)
def cost(fn, n=10):
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(n): fn(i)
    c = (time.time() - t0) / n * 1e3
    torch.cuda.synchronize(); w = (time.time() - t0) / n * 1e3
    return "cpu %.1f ms wall %.1f ms" % (c, w)
name = "3d_walker_7_full"
m = mjcf.load_asset(name); L = m.num_limbs
gd = G.getGraphDict(m.parents, TRAV, [], device=dev)
batch = {"obs": torch.from_numpy(synth_obs(L, 100, 1).astype(np.float32)).to(dev), "next_obs": torch.from_numpy(synth_obs(L, 100, 2).astype(np.float32)).to(dev),
         "action": (torch.rand(100, 3 * L, device=dev) * 2 - 1), "reward": torch.randn(100, 1, device=dev), "done": torch.zeros(100, 1, device=dev)}
gr.warm(0, gd, L, batch, iters=2)
for it in range(2): gr.update(0, gd, L, batch, it)
sl = gr.slots[0]
rep = lambda i: sl["graphs"][i % 2].replay()
print("A baseline                  ", cost(rep), flush=True)
streams = [torch.cuda.Stream() for _ in range(32)]
print("B +32 torch streams         ", cost(rep), flush=True)
big = [torch.zeros(200000, 600, device=dev) for _ in range(20)]
print("C +9.6 GB of tensors        ", cost(rep), flush=True)
from sgrl_amd.vec_env import BatchedModularVecEnv
env = BatchedModularVecEnv(["3d_walker_7_full", "3d_hopper_3_shin"], 64, seed=1, device="cuda:0")
env.reset_device(); torch.cuda.synchronize()
print("D +engine, 2 morphologies   ", cost(rep), flush=True)
HELD_OUT = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_arm", "3d_humanoid_8_right_knee",
            "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
names = sorted(n for n in mjcf.list_assets() if n not in HELD_OUT)
env2 = BatchedModularVecEnv(names, 64, seed=1, device="cuda:0")
env2.reset_device(); torch.cuda.synchronize()
print("E +engine, %d morphologies  " % len(names), cost(rep), "launch groups", env2.launch_groups if hasattr(env2, "launch_groups") else "?", flush=True)
a = torch.zeros((env2.num_envs, env2.action_max_len), device=dev)
for _ in range(5): env2.step_device(a)
torch.cuda.synchronize()
print("F after engine steps        ", cost(rep), flush=True)
