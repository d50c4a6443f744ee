#!/usr/bin/env python3
"""Quick SET-forward timing on the GPU box: walker mix, n envs per morphology."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgrl_amd.set_policy import make_policy
from sgrl_amd.set_hip import HipSetActor
from sgrl_amd import graph as G, mjcf
names = sorted(n for n in mjcf.list_assets() if "walker" in n)
per = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pol = make_policy(device="cuda:0").eval()
gds = [G.getGraphDict(mjcf.load_asset(n).parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cuda:0")) for n in names]
act = HipSetActor(pol)
act.configure(gds, [per] * len(names))
obs = torch.randn((per * len(names), 287), device="cuda") * 0.5
out = act.forward_batch(obs)
torch.cuda.synchronize()
act.scale_redos()
ms = act.time_forward(obs, out, 5)
nodes = act.num_nodes
print("tile repeats (row-scale estimates that fell short) in the timed forwards:", act.scale_redos())
print("envs %d nodes %d: %.3f ms/forward = %.2f us/env-step, %.1f TFLOP/s (10.07 MFLOP/node)" % (
    per * len(names), nodes, ms, ms * 1e3 / (per * len(names)), nodes * 10.07e6 / (ms * 1e-3) / 1e12))
