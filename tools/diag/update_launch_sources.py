#!/usr/bin/env python3
"""Which source lines of one TD3 update launch how many kernels (torch.profiler, eager, walker_7, batch = agent_batch_size = 256): the launches of a
policy iteration (it % policy_freq == 0: critic + actor + target updates) grouped by the innermost frame inside sgrl_amd/ and by
kernel name.  Usage: update_launch_sources.py [morphology]"""
import os, sys, collections, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from sgrl_amd import graph as G, mjcf
from sgrl_amd.rollout import TRAV
from sgrl_amd.td3 import Agent, default_train_args
name = sys.argv[1] if len(sys.argv) > 1 else "3d_walker_7_full"
dev = torch.device("cuda:0")
torch.manual_seed(0)
targs = default_train_args()
agent = Agent(targs, device=dev)
m = mjcf.load_asset(name)
gd = G.getGraphDict(m.parents, TRAV, [], device=dev)
agent.change_morphology(gd)
agent.models2train()
B, L = targs.agent_batch_size, m.num_limbs
g = torch.Generator(device=dev).manual_seed(1)
def obs():
    o = torch.randn((B, L, 41), device=dev, generator=g) * 0.5
    o[:, :, 3:5] = 0; o[:, :, 5] = -9.81; o[:, :, 8] = 0
    return o.reshape(B, 41 * L).contiguous()
batch = {"obs": obs(), "next_obs": obs(), "action": (torch.rand(B, 3 * L, device=dev) * 2 - 1),
         "reward": torch.randn(B, 1, device=dev), "done": torch.zeros(B, 1, device=dev)}
for it in range(4):
    agent.update(batch, it, lazy_stats=True)
torch.cuda.synchronize()
for which, it in (("policy iteration", 4), ("critic-only iteration", 5)):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        agent.update(batch, it, lazy_stats=True, skip_unused_critic_grads=True)
        torch.cuda.synchronize()
    by_src, by_op, by_shape = collections.Counter(), collections.Counter(), collections.Counter()
    n = 0
    for ev in prof.events():
        # CPU-side ops that launched kernels directly: count their kernels against the innermost sgrl_amd frame of their stack
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        # only leaf ops (children may carry the kernels too: take events without cpu children that have kernels)
        if any(c.kernels for c in ev.cpu_children):
            continue
        k = len(ev.kernels)
        n += k
        src = "?"
        for fr in ev.stack or []:
            if "sgrl_amd/" in fr and "site-packages" not in fr:
                src = re.sub(r".*sgrl_amd/", "", fr)
                break
        by_src[src] += k
        by_op[ev.name] += k
        by_shape[(ev.name, str(ev.input_shapes)[:110])] += k
    print("== %s: %d kernel launches attributed" % (which, n))
    for s, c in by_src.most_common(45):
        print("   %4d  %s" % (c, s))
    print("   -- by op and input shapes")
    for (o, sh), c in by_shape.most_common(70):
        print("   %4d  %-34s %s" % (c, o, sh))
    print("   -- by op")
    for s, c in by_op.most_common(25):
        print("   %4d  %s" % (c, s))
