#!/usr/bin/env python3
"""Experiment: two half-batches on two streams -- physics of one overlapping the SET forward of the other."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgrl_amd import mjcf
from sgrl_amd.rollout import Rollout
from sgrl_amd.set_policy import make_policy
names = sorted(n for n in mjcf.list_assets() if "walker" in n)
pol = make_policy(device="cuda:0").eval()
per = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A = Rollout(names, per, policy=pol, seed=1, device="cuda:0", rank=0)
B = Rollout(names, per, policy=pol, seed=1, device="cuda:0", rank=1)
for r in (A, B):
    r.reset()
    for _ in range(150):
        r.step(r.random_actions())
torch.cuda.synchronize()
K = 20
# sequential reference
t0 = time.time()
for _ in range(K):
    for r in (A, B):
        obs, *_ = r.step(r.random_actions())
        r.policy_forward(obs)
torch.cuda.synchronize()
seq = (time.time() - t0) / K
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ev = [torch.cuda.Event(), torch.cuda.Event()]
torch.cuda.synchronize()
t0 = time.time()
for _ in range(K):
    # phase 1: A physics || B policy ; phase 2: B physics || A policy
    with torch.cuda.stream(s1):
        A.step(A.random_actions())
    with torch.cuda.stream(s2):
        B.policy_forward(B.env.obs)
    s1.synchronize(); s2.synchronize()
    with torch.cuda.stream(s1):
        B.step(B.random_actions())
    with torch.cuda.stream(s2):
        A.policy_forward(A.env.obs)
    s1.synchronize(); s2.synchronize()
torch.cuda.synchronize()
ovl = (time.time() - t0) / K
n = A.env.num_envs + B.env.num_envs
print("envs %d: sequential %.2f ms/step (%.0f env-steps/s) | overlapped %.2f ms/step (%.0f env-steps/s)" % (n, seq * 1e3, n / seq, ovl * 1e3, n / ovl))
