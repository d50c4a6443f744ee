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
for it in range(40):
    act.forward_batch(obs); torch.cuda.synchronize()
    zc = act.peek(2, 96).astype(np.float64).reshape(-1, 3, 32)
    fn = act.peek(3, 1)[:, 0]
    gm = np.einsum("nsa,nsb->nab", zc, zc)
    ref = np.sqrt((gm ** 2).sum((1, 2))) + 1.0
    bad = np.nonzero(np.abs(fn - ref) > 1e-4 * ref)[0]
    print("iter", it, "bad rows", len(bad), "first", bad[:20], "last", bad[-5:], "rows mod 128:", sorted(set((bad % 128).tolist()))[:40])
    if len(bad):
        print("  fn", fn[bad[:6]], "ref", ref[bad[:6]])
    for r in bad[:4]:
        z = zc[r]
        def f(zz):
            gm = zz.T @ zz
            return np.sqrt((gm ** 2).sum()) + 1
        cands = {}
        for q in range(4):
            for h in range(2):
                for sx in range(3):
                    zz = z.copy(); zz[sx, 8 * q + 4 * h: 8 * q + 4 * h + 4] = 0
                    cands["zero q%d h%d s%d" % (q, h, sx)] = f(zz)
        zz = z.copy(); zz[:, 30:] = 0; cands["zero gdir"] = f(zz)
        for q in range(4):
            zz = z.copy(); zz[:, 8 * q:8 * q + 8] = 0; cands["zero quarter %d" % q] = f(zz)
        best = sorted(cands.items(), key=lambda kv: abs(kv[1] - fn[r]))[:3]
        print("   row", r, "fn", fn[r], "ref", ref[r], "closest:", best)
