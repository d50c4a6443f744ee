#!/usr/bin/env python3
"""Is the batched SET forward a FUNCTION of its inputs?  The same observations and weights through Rollout.policy_forward N times,
every result compared bit for bit with the first -- on the config-5 batch (23 morphologies x 24 environments: the chain kernels with
their side stream) and interleaved with engine steps / replay packing like a collection loop, while other copies of this script load
the GPU (timing noise: a missing dependency between the forward's two streams would show as a mismatch that comes and goes).
usage: forward_determinism.py [iters=400] [per_morph=24] [tag]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgrl_amd import mjcf
from sgrl_amd.rollout import Rollout
from sgrl_amd.set_policy import make_policy
from oracle.formula import apply_default_like_

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
per = int(sys.argv[2]) if len(sys.argv) > 2 else 24
tag = sys.argv[3] if len(sys.argv) > 3 else ""
HELD = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
        "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
names = sorted(n for n in mjcf.list_assets() if n not in HELD)
pol = make_policy(device="cuda:0").eval()
apply_default_like_(pol, 6)
ro = Rollout(names, per, policy=pol, seed=3, device="cuda:0", hold_weights=True)
ro.reset()
for _ in range(60):
    ro.step(ro.random_actions())
obs = ro.env.obs.clone()
ref = ro.policy_forward(obs).clone()
torch.cuda.synchronize()
bad = 0
worst = 0.0
t0 = time.time()
for i in range(iters):
    if i % 3 == 0:                       # a collection loop's neighbours: an engine step on the main stream, some torch work
        ro.step(ro.random_actions())
        junk = torch.randn(256, 1024, device="cuda") @ torch.randn(1024, 512, device="cuda")
    a = ro.policy_forward(obs)
    if not torch.equal(a, ref):
        bad += 1
        worst = max(worst, float((a - ref).abs().max()))
torch.cuda.synchronize()
print("%s forward_determinism: %d forwards of %d nodes, %d differ from the first (largest difference %.3e), %.1f s" % (
    tag, iters, ro.actor.num_nodes, bad, worst, time.time() - t0), flush=True)
