#!/usr/bin/env python3
"""The no-grad HIP forwards the TD3 update runs for its targets (actor_target, twin critic_target: batch 256 -> the big-batch product
path from 2 048 nodes = 8 limbs on) on REAL replay rows of a config-5 run against the same modules in float64 on PyTorch -- per
morphology: largest |dQ| relative to max |Q|, largest action difference, and the TD target's.  usage: target_forward_audit.py [rounds=2] [seed=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sgrl_amd import graph as G, mjcf, set_policy
from sgrl_amd.set_policy import make_critic, make_policy
from sgrl_amd.td3 import default_train_args
from sgrl_amd.train_loop import DeviceTrainer

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
HELD = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
        "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
names = sorted(n for n in mjcf.list_assets() if n not in HELD)
tr = DeviceTrainer(names, 24, args=default_train_args(), seed=seed, device="cuda:0", max_buffer_size=100000, graph_updates=True, lag_flag=False)
for _ in range(400):
    if tr.collect_step(random_actions=True):
        tr.begin_round()
for r in range(rounds):
    s = tr.train_round()
print("trained %d rounds: return %.1f" % (rounds, s["performance/train_return"]), flush=True)
dev = torch.device("cuda:0")
ag = tr.agent
pol64 = make_policy(device=dev, use_hip=False).double().eval()
cri64 = make_critic(device=dev, use_hip=False).double().eval()
pol64.load_state_dict({k: v.double() for k, v in ag.actor_target.state_dict().items()})
cri64.load_state_dict({k: v.double() for k, v in ag.critic_target.state_dict().items()})
ag.models2eval()
for twin in (False, True):
    set_policy.TWIN_TARGETS = twin
    print("== target critics through %s" % ("twin_forward (training kernels)" if twin else "the rollout kernels (set_actor.hip)"))
    for k, name in enumerate(names):
        L = tr.ro.env.num_limbs[k]
        b = tr.buffers[k].sample(256, generator=tr.gen)
        gd = tr.graph_dicts[k]
        gd64 = {kk: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for kk, v in gd.items()}
        with torch.no_grad():
            ag.change_morphology(gd)
            pol64.change_morphology(gd64); cri64.change_morphology(gd64)
            a = ag.actor_target(b["next_obs"])
            a64 = pol64(b["next_obs"].double())
            q1, q2 = ag.critic_target(b["next_obs"], a)
            p1, p2 = cri64(b["next_obs"].double(), a.double())
        qs = float(p1.abs().max())
        print("%-36s L=%2d nodes %4d |obs|max %6.1f: action err %.1e  Q err %.1e (|Q|max %.2e, rel %.1e)" % (
            name, L, 256 * L, float(b["next_obs"].abs().max()), float((a.double() - a64).abs().max()),
            max(float((q1.double() - p1).abs().max()), float((q2.double() - p2).abs().max())), qs,
            max(float((q1.double() - p1).abs().max()), float((q2.double() - p2).abs().max())) / max(qs, 1e-30)), flush=True)
