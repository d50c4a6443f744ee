#!/usr/bin/env python3
"""Diagnostic: the no-grad target critics of every shipped morphology through both paths -- one twin pass of the training kernels
(set_policy.TWIN_TARGETS) against two passes of the rollout kernels -- at the update batch (256 rows), random-init weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgrl_amd import graph as G, mjcf, set_policy
from sgrl_amd.rollout import TRAV
from sgrl_amd.set_policy import make_critic
dev = torch.device("cuda:0")
torch.manual_seed(0)
crit = make_critic(device="cuda:0").eval()
B = int(os.environ.get("B", "256"))
worst = 0.0
for name in sorted(mjcf.list_assets()):
    m = mjcf.load_asset(name)
    L = m.num_limbs
    crit.change_morphology(G.getGraphDict(m.parents, TRAV, [], device=dev))
    g = torch.Generator(device=dev).manual_seed(L)
    obs = torch.randn((B, L, 41), device=dev, generator=g) * 0.5
    obs[:, :, 3:5] = 0; obs[:, :, 5] = -9.81; obs[:, :, 8] = 0
    obs = obs.reshape(B, 41 * L).contiguous()
    act = torch.rand(B, 3 * L, device=dev, generator=g) * 2 - 1
    with torch.no_grad():
        set_policy.TWIN_TARGETS = True
        q1, q2 = crit(obs, act)
        set_policy.TWIN_TARGETS = False
        r1, r2 = crit(obs, act)
    d = max(float((q1 - r1).abs().max()), float((q2 - r2).abs().max()))
    bad = not (torch.isfinite(q1).all() and torch.isfinite(q2).all())
    worst = max(worst, d if np.isfinite(d) else 1e30)
    print("%-40s L %2d  max |twin - rollout kernels| %.3e  scale %.3e %s" % (name, L, d, float(r1.abs().max()), "NON-FINITE" if bad else ""), flush=True)
print("worst", worst)
