#!/usr/bin/env python3
"""Host time and stream time of RoundCollector.record() alone, tensor form against the one-launch form (8 188 environments)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgrl_amd import rollout
n = 8188
g = torch.Generator(device="cuda").manual_seed(0)
rew = torch.randn(n, device="cuda", generator=g)
done = (torch.rand(n, device="cuda", generator=g) < 0.01)
for fused in (False, True, False, True):
    rollout.FUSED_RECORD = fused
    c = rollout.RoundCollector(n, max_episode_steps=1000, device="cuda:0")
    for _ in range(20):
        c.record(rew, done, sync=False)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(500):
        c.record(rew, done, sync=False)
    t_host = (time.time() - t0) / 500
    torch.cuda.synchronize()
    t_all = (time.time() - t0) / 500
    print("fused %-5s: host %.1f us per call, with the stream drained %.1f us per call" % (fused, t_host * 1e6, t_all * 1e6), flush=True)
rollout.FUSED_RECORD = True
