#!/usr/bin/env python3
"""Probe: two half batches on two streams (each with its own true dependency chain  obs -> SET forward -> step -> obs), against
one batch of the same total size doing forward and step back to back: do the step kernel's one-wave workgroups fill the tails and
launch boundaries of the forward's tile kernels (and vice versa)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sgrl_amd.set_policy import make_policy
from sgrl_amd.rollout import Rollout
from sgrl_amd import mjcf
dev = torch.device("cuda:0")
names = sorted(n for n in mjcf.list_assets() if "walker" in n)
per = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pol = make_policy(device="cuda:0").eval()

def make(n, seed):
    ro = Rollout(names, n, policy=pol, seed=seed, device=dev, hold_weights=True)
    ro.reset()
    for _ in range(60):
        ro.step(ro.random_actions())
    return ro

def chain(ro):                      # one policy-driven time step of one group
    a = ro.policy_forward(ro.env.obs)
    ro.step(a)

def timed(fn, steps=40, reps=5):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / steps * 1e3)
    return float(np.median(out)), float(min(out))

one = make(per, 0)
for _ in range(5): chain(one)
ms1 = timed(lambda: chain(one))
n_env = one.env.num_envs
print("one batch of %d envs, forward -> step: %.3f ms per step (min %.3f) = %.0f env-steps/s" % (n_env, ms1[0], ms1[1], n_env / ms1[0] * 1e3))
del one
groups = [make(per // parts, 10 + i) for i in range(parts)]
streams = [torch.cuda.Stream() for _ in range(parts)]
def piped():
    for ro, s in zip(groups, streams):
        with torch.cuda.stream(s):
            chain(ro)
for _ in range(5): piped()
ms2 = timed(piped)
tot = sum(g.env.num_envs for g in groups)
print("%d groups of %d envs on %d streams: %.3f ms per step of all (min %.3f) = %.0f env-steps/s  (%.3fx)" % (
    parts, groups[0].env.num_envs, parts, ms2[0], ms2[1], tot / ms2[0] * 1e3, (tot / ms2[0]) / (n_env / ms1[0])))
# the groups one after the other on one stream: what the smaller batches cost by themselves
def serial():
    for ro in groups:
        chain(ro)
ms3 = timed(serial)
print("the same groups one after the other on one stream: %.3f ms (min %.3f)" % ms3)
