#!/usr/bin/env python3
"""Two trainers from the same seed take the same round-1 data (the rollout is deterministic) and the same sampled batches; only the
arithmetic of their updates differs: (A) hipGraph replay + table optimizer + deferred gradients, (B) plain eager Agent.update with
torch's Adam.  After the first round of updates: how far apart are the two actors -- parameters, and deterministic actions on the
next round's first observations -- and how do both play round 2?  Run under SGRL_SET_GEMM=f32 and without for the rollout forms.
usage: round1_divergence.py [seed=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgrl_amd import mjcf
from sgrl_amd.td3 import default_train_args
from sgrl_amd.train_loop import DeviceTrainer

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
HELD = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
        "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
names = sorted(n for n in mjcf.list_assets() if n not in HELD)
res = {}
for arm, graphed in (("graphed", True), ("eager", False)):
    tr = DeviceTrainer(names, 24, args=default_train_args(), seed=seed, device="cuda:0", max_buffer_size=100000, graph_updates=graphed, lag_flag=False)
    for _ in range(400):
        if tr.collect_step(random_actions=True):
            tr.begin_round()
    s1 = tr.train_round()
    params = torch.cat([p.detach().reshape(-1) for p in tr.agent.actor.parameters()]).double().cpu()
    cparams = torch.cat([p.detach().reshape(-1) for p in tr.agent.critic.parameters()]).double().cpu()
    obs = tr.ro.env.obs.clone()                  # the observations round 2 starts from (same reset in both arms)
    act = tr.ro.policy_forward(obs).clone().cpu()
    fills = [b.max_sample_size for b in tr.buffers]
    s2 = tr.train_round()
    res[arm] = dict(params=params, cparams=cparams, obs=obs.cpu(), act=act, fills=fills, r1=s1["performance/train_return"], r2=s2["performance/train_return"],
                    it1=s1["per_morph_iter"], it2=s2["per_morph_iter"])
    print("%s: round 1 return %.2f (iters %d), round 2 return %.2f (iters %d), buffer fills %s" % (arm, res[arm]["r1"], res[arm]["it1"], res[arm]["r2"], res[arm]["it2"], fills[:6]), flush=True)
    del tr
    torch.cuda.empty_cache()
a, b = res["graphed"], res["eager"]
print("SGRL_SET_GEMM=%s seed %d" % (os.environ.get("SGRL_SET_GEMM", "(default f16x3)"), seed))
print("round-2 start observations identical: %s" % bool(torch.equal(a["obs"], b["obs"])))
print("actor parameters after round 1: max |d| %.3e, rms %.3e (|p| rms %.3e)" % (float((a["params"] - b["params"]).abs().max()), float((a["params"] - b["params"]).pow(2).mean().sqrt()), float(a["params"].pow(2).mean().sqrt())))
print("critic parameters after round 1: max |d| %.3e, rms %.3e" % (float((a["cparams"] - b["cparams"]).abs().max()), float((a["cparams"] - b["cparams"]).pow(2).mean().sqrt())))
print("deterministic actions on those observations: max |d| %.3e, mean |d| %.3e; share of saturated entries (|a| > 0.99): %.2f / %.2f" % (
    float((a["act"] - b["act"]).abs().max()), float((a["act"] - b["act"]).abs().mean()), float((a["act"].abs() > 0.99).float().mean()), float((b["act"].abs() > 0.99).float().mean())))
