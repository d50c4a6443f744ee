#!/usr/bin/env python3
"""In which product form / batch size does a Rollout that holds its actor's packed weights act on the CURRENT parameters after they
change?  For each (envs per morphology) the HIP forward against the PyTorch path before and after an in-place change of every
parameter + weights_changed().  usage: [SGRL_SET_GEMM=f32] stale_pack_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgrl_amd.rollout import Rollout
from sgrl_amd.set_policy import make_policy
from oracle.formula import apply_default_like_

names = ["3d_walker_7_full", "3d_hopper_3_shin", "3d_cheetah_14_full"]


def torch_actions(ro, pol, obs):
    out = torch.zeros_like(ro.policy_actions)
    env = ro.env
    with torch.no_grad():
        for k, sl in enumerate(env.morph_slices):
            L = env.num_limbs[k]
            x = obs[sl, :41 * L].reshape(sl.stop - sl.start, L, 41)
            out[sl, :3 * L] = (pol.max_action * torch.tanh(pol.actor(x, ro.graph_dicts[k], False))).reshape(sl.stop - sl.start, 3 * L)
    return out


for per, hold in ((4, True), (200, True), (200, False)):
    pol = make_policy(device="cuda:0").eval()
    apply_default_like_(pol, 6)
    ro = Rollout(names, per, policy=pol, seed=1, device="cuda:0", hold_weights=hold)
    ro.reset()
    obs = ro.env.obs.clone()
    d0 = float((ro.policy_forward(obs) - torch_actions(ro, pol, obs)).abs().max())
    g = torch.Generator(device="cuda").manual_seed(5)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.02 * torch.randn(p.shape, device="cuda", generator=g))
    ro.weights_changed()
    a1 = ro.policy_forward(obs).clone()
    t1 = torch_actions(ro, pol, obs)
    print("SGRL_SET_GEMM=%s nodes %5d hold %s: |HIP - torch| before the change %.2e, after change + weights_changed() %.2e (|a| mean %.3f)" % (
        os.environ.get("SGRL_SET_GEMM", "f16x3"), ro.actor.num_nodes, hold, d0, float((a1 - t1).abs().max()), float(t1.abs().mean())), flush=True)
