import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from oracle.formula import synth_obs
from sgrl_amd import mjcf
from sgrl_amd.td3 import default_train_args, GraphedUpdates
from sgrl_amd.train_loop import DeviceTrainer
names = ["3d_walker_7_full", "3d_hopper_3_shin", "3d_humanoid_9_full"] if len(sys.argv) < 3 else sys.argv[2:]
mode = sys.argv[1]
tr = DeviceTrainer(names, 64, args=default_train_args(), seed=1, device="cuda:0", max_buffer_size=50000, graph_updates=True, tune_gemms=False)
if mode != "nowarmup":
    tr.warmup(40)
dev = torch.device("cuda:0")
gr = tr.graphed
if mode == "freshagent":
    from sgrl_amd.td3 import Agent
    ag = Agent(default_train_args(), device=dev)
    gr = GraphedUpdates(ag, 100)
    ag.models2train()
tr.agent.models2train()
data = []
for k, name in enumerate(names):
    L = tr.ro.env.num_limbs[k]; gd = tr.graph_dicts[k]
    if mode == "sampled":
        batch = tr.buffers[k].sample(100, generator=tr.gen)
    else:
        batch = {"obs": torch.from_numpy(synth_obs(L, 100, 1).astype(np.float32)).to(dev), "next_obs": torch.from_numpy(synth_obs(L, 100, 2).astype(np.float32)).to(dev),
                 "action": (torch.rand(100, 3 * L, device=dev) * 2 - 1), "reward": torch.randn(100, 1, device=dev), "done": torch.zeros(100, 1, device=dev)}
    gr.warm(k, gd, L, batch, iters=2)
    data.append((gd, L, batch))
for k, name in enumerate(names):
    gd, L, batch = data[k]
    for it in range(4): gr.update(k, gd, L, batch, it)
    sl = gr.slots[k]
    for flag in (0, 1):
        torch.cuda.synchronize(); t0 = time.time()
        for i in range(6): sl["graphs"][flag].replay()
        torch.cuda.synchronize()
        print("%s %-20s flag %d: wall %.1f ms" % (mode, name, flag, (time.time() - t0) / 6 * 1e3), flush=True)
