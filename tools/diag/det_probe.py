import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sgrl_amd.set_policy import make_policy
from sgrl_amd.set_hip import HipSetActor
from sgrl_amd import graph as G, mjcf
names = sorted(n for n in mjcf.list_assets() if "walker" in n)
pol = make_policy(device="cuda:0").eval()
gds = [G.getGraphDict(mjcf.load_asset(n).parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cuda:0")) for n in names]
act = HipSetActor(pol)
act.configure(gds, [1024] * 8)
torch.manual_seed(0)
obs = torch.randn((8192, 287), device="cuda") * 0.5
ref = act.forward_batch(obs).clone()
for stage in list(range(6)) + [-1]:
    act.debug_stop_after(stage)
    bad = {}
    for it in range(6):
        out = act.forward_batch(obs).clone(); torch.cuda.synchronize()
        bufs = {w: act.peek(w, p) for w, p in ((0, 384), (1, 256), (2, 96), (3, 1), (4, 768), (7, 96), (8, 384), (9, 128))}
        if it == 0: base = bufs; continue
        for w in bufs:
            d = np.abs(bufs[w] - base[w]).max()
            if d > 0: bad[w] = max(bad.get(w, 0), d)
    print("stage", stage, "nondeterministic buffers:", bad)
