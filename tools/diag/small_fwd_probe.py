"""GPU time of one SET actor forward at the TD3 update's batch (100 environments of one morphology) and at one environment,
replayed from a hipGraph; run with SGRL_SET_SMALL_NODES=0 for the 128 x 128 tile path at every size."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from oracle.formula import synth_obs
from sgrl_amd import mjcf
from sgrl_amd.td3 import default_train_args, Agent
from sgrl_amd import graph as G
from sgrl_amd.rollout import TRAV
dev = torch.device("cuda:0")
ag = Agent(default_train_args(), device=dev)
CASES = (("3d_hopper_3_shin", 100), ("3d_walker_7_full", 100), ("3d_cheetah_14_full", 100), ("3d_walker_7_full", 1), ("3d_cheetah_14_full", 20))
if len(sys.argv) > 2:
    CASES = ((sys.argv[1], int(sys.argv[2])),)
for name, B in CASES:
    m = mjcf.load_asset(name); gd = G.getGraphDict(m.parents, TRAV, [], device=dev); L = len(m.parents)
    obs = torch.from_numpy(synth_obs(L, B, 1).astype(np.float32)).to(dev)
    ag.change_morphology(gd)
    def f():
        with torch.no_grad():
            return ag.actor_target(obs)
    y = f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            y2 = f()
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(50): g.replay()
    torch.cuda.synchronize()
    print("%-20s B %3d nodes %5d: %.1f us / forward" % (name, B, B * L, (time.time() - t0) / 50 * 1e6), flush=True)
