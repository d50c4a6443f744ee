#!/usr/bin/env python3
"""Every shipped morphology, batch sizes on both sides of the small-batch threshold (2 048 nodes): the no-grad HIP forward of the SET
actor and of the twin critics (what the rollout and the TD3 target networks run) against the same modules in float64 on PyTorch.
Default-like and formula weights.  usage: forward_check_all.py [name-filter]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle.formula import apply_default_like_, apply_formula_, synth_obs
from sgrl_amd import graph as G, mjcf
from sgrl_amd.set_policy import make_critic, make_policy

flt = sys.argv[1] if len(sys.argv) > 1 else ""
TRAV = ["pre", "inlcrs", "postlcrs"]
dev = torch.device("cuda:0")
names = sorted(n for n in mjcf.list_assets() if any(f in n for f in flt.split(",")))
for wname, init in (("default-like", lambda m: apply_default_like_(m, 6)), ("formula", apply_formula_)):
    pol, cri = make_policy(device=dev).eval(), make_critic(device=dev).eval()
    init(pol); init(cri)
    pol64, cri64 = make_policy(device=dev, use_hip=False).double().eval(), make_critic(device=dev, use_hip=False).double().eval()
    pol64.load_state_dict({k: v.double() for k, v in pol.state_dict().items()})
    cri64.load_state_dict({k: v.double() for k, v in cri.state_dict().items()})
    for name in names:
        m = mjcf.load_asset(name)
        L = m.num_limbs
        gd = G.getGraphDict(m.parents, TRAV, [], device=dev)
        gd64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in gd.items()}
        line = ["%-12s %-40s L=%2d" % (wname, name, L)]
        for B in (24, 256):
            obs = torch.from_numpy(synth_obs(L, B, 3 + L)).to(dev)
            act_in = torch.rand(B, 3 * L, device=dev, dtype=torch.float64) * 2 - 1
            with torch.no_grad():
                pol.change_morphology(gd); cri.change_morphology(gd)
                pol64.change_morphology(gd64); cri64.change_morphology(gd64)
                a = pol(obs.float()).double()
                q1, q2 = cri(obs.float(), act_in.float())
                a64 = pol64(obs)
                p1, p2 = cri64(obs, act_in)
            ea = float((a - a64).abs().max())
            eq = max(float((q1.double() - p1).abs().max()), float((q2.double() - p2).abs().max()))
            line.append("B=%3d (%4d nodes): action err %.1e (|a| %.2f) Q err %.1e (|Q| %.2e)" % (B, B * L, ea, float(a64.abs().max()), eq, float(p1.abs().max())))
        print(" | ".join(line), flush=True)
