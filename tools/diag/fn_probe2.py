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
act.debug_stop_after(0)
np.set_printoptions(linewidth=250, precision=5)
for it in range(12):
    act.forward_batch(obs); torch.cuda.synchronize()
    zc = act.peek(2, 96).reshape(-1, 3, 32)
    dbg = act.peek(5, 24).reshape(-1, 4, 6)
    zq = zc.reshape(-1, 3, 4, 8).astype(np.float64)     # [n, s, kq, 8]
    x, y, z = zq[:, 0], zq[:, 1], zq[:, 2]
    ref = np.stack([(x * x).sum(-1), (y * y).sum(-1), (z * z).sum(-1), (x * y).sum(-1), (x * z).sum(-1), (y * z).sum(-1)], -1)   # [n, kq, 6]
    err = np.abs(dbg - ref) > 1e-4 * (np.abs(ref) + 1)
    rows = np.nonzero(err.any((1, 2)))[0]
    print("iter", it, "bad rows", rows[:12])
    for r in rows[:2]:
        print("  row", r, "bad (kq,k):", np.argwhere(err[r]).tolist())
        print("  dbg", dbg[r].reshape(-1)); print("  ref", ref[r].reshape(-1))
